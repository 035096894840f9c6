"""`VectorQuantize` for MI355X: the reference's nn.Module API (vector_quantize_pytorch.py:802-1403)
over the HIP hot path.

Same constructor keywords (vqp.py:803-849), same forward keywords (:1093-1107), same outputs
`(quantized, indices, commit_loss[, LossBreakdown])`, same state_dict keys.  Options that are not
on the north-star hot path (SURVEY.md §2.1 "delegated", §8f) raise NotImplementedError at
construction / call time -- there is no silent fallback and nothing here runs on the CPU.
"""
from __future__ import annotations

import os

from collections import namedtuple
from functools import cache
from typing import Callable, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import Tensor, nn

from . import _lib as L
from .codebook import Codebook

LossBreakdown = namedtuple('LossBreakdown', ['commitment', 'codebook_diversity', 'orthogonal_reg', 'inplace_optimize'])


@cache
def _is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def SCORE_CHUNK_BYTES() -> int:
    """largest score matrix (`dist`, vqp.py:741-743) the options that read whole score rows materialise at once; beyond it the
    rows are processed a slice of positions at a time (VQHIP_SCORE_CHUNK_MB, default 128)"""
    return int(float(os.environ.get("VQHIP_SCORE_CHUNK_MB", "128")) * (1 << 20))


class _QuantizeFn(torch.autograd.Function):
    """Forward = HIP assign/gather/EMA (+ vq_route_kernel for the routed value); backward = vq_route_kernel,
    the closed-form gradients of the reference's graph:
      commit loss   mean((q.detach() - x)^2)              -> 2 (x - q) / count        (vqp.py:1327)
      straight-through x + (q - x).detach()               -> g                        (vqp.py:282-283)
      rotation trick (all of u, qh, w, s detached)        -> s (g - 2 (g.w) w + 2 (g.qh) u)   (vqp.py:287-318)
    Only x and q are saved; u, qh, w, s are recomputed in the backward kernel."""

    @staticmethod
    def forward(ctx, x, vq, mask, kw, loss_scale=1.0, embed_param=None, fill=0):
        """loss_scale: constant folded into the squared-error reduction (1 / numel for the unmasked commit loss), so that the
        third output IS mean((q - x)^2) without further elementwise kernels.
        embed_param [1, C, D]: a codebook that receives gradients (learnable_codebook / the output of vq_bridge; vqp.py:710, 766) --
        searched as given; its gradient is that of the commitment loss mean((codes[idx] - x)^2) with `quantize` NOT detached
        (vqp.py:1214, 1327): g_loss * 2 * loss_scale * (count_c * codes_c - sum of the rows quantized to c), i.e. the EMA statistics
        of x (one vqhip_ema_accumulate in backward) -- the routed output itself carries no gradient to the codes (vqp.py:282-318).
        fill (with a mask): the padding rows of the output take x's rows (1) or zeros (2) and their indices -1 here, in place on the
        padding only (vqhip_mask_fill_rows) -- the reference's torch.where(mask, quantize, orig_input | zeros) / where(mask, indices, -1)
        (vqp.py:1386-1394), whose gradient the backward kernel reproduces (padding rows: the upstream gradient, or none)."""
        cb = vq._codebook
        mode = 0
        if vq.training and x.requires_grad and vq.route_gradients_to_input:
            mode = L.ROTATION if vq.rotation_trick else L.STRAIGHT_THROUGH
        # Routed output on ONE codebook: `quantized` is a function of x and the winning CODE, so the [N, D] tensor of gathered codes need
        # not exist -- the search returns indices only and the routing kernels (forward here, backward below) gather the code rows from a
        # C x D snapshot in the rows' dtype (taken before the search: the EMA fold rewrites `embed` in place).  1 GB less HBM traffic per
        # cfg-2 step in each direction.
        gather = (mode != 0 and x.ndim == 3 and cb.num_codebooks == 1 and cb._is_initted() and not cb.affine_param
                  and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16))
        src = cb.embed if embed_param is None else embed_param
        codes = src[0].detach().to(x.dtype, copy=True) if gather else None
        if embed_param is not None:
            kw = dict(kw, embed_override=embed_param.detach())
        r = cb.quantize(x, mask=mask, want_sqerr=vq.training and vq.has_commitment_loss, loss_scale=float(loss_scale),
                        **(dict(kw, want_q=False) if gather else kw))
        q, idx = r["q"], r["idx"]
        loss_sum = None
        ctx.loss_scale = float(loss_scale)
        if vq.training and vq.has_commitment_loss:
            loss_sum = r["loss"] if r.get("loss") is not None else L.reduce_partials(r["sqerr_partials"], r["nblk"], float(loss_scale))
        out = q
        if gather:
            idx = idx.contiguous()
            out = L.route_fwd_gather(x, codes, idx, mode)
        elif mode != 0:
            out = L.route_fwd(x, q, mode)
        ctx.fill = int(fill) if mask is not None else 0
        if ctx.fill:
            idx = idx.contiguous()
            L.mask_fill_rows(out, x, mask, idx, zeros=ctx.fill == 2)
        ctx.mode = mode
        ctx.gather = gather
        ctx.has_mask = mask is not None
        ctx.has_embed = embed_param is not None and embed_param.requires_grad
        extra = []
        if ctx.has_embed:     # (a snapshot: dead-code replacement may rewrite rows of the parameter in place before backward runs)
            extra = [idx, embed_param[0].detach().float().clone()]
        ctx.save_for_backward(x, *((codes, idx) if gather else (q,)), *([mask] if mask is not None else []), *extra)
        ctx.mark_non_differentiable(idx)
        if loss_sum is None:
            loss_sum = torch.zeros((), dtype=torch.float32, device=x.device)
        return out, idx, loss_sum

    @staticmethod
    def backward(ctx, g_out, g_idx, g_loss):
        tensors = ctx.saved_tensors
        x = tensors[0]
        g_embed = None
        if ctx.has_embed:
            idx_e, codes_e = tensors[-2:]
            tensors = tensors[:-2]
            if g_loss is not None:
                C = codes_e.shape[0]
                count, esum = L.ema_accumulate(x, idx_e.reshape(-1), C, row_mask=tensors[-1] if ctx.has_mask else None)
                g_embed = ((count[:, None] * codes_e - esum) * (2.0 * ctx.loss_scale * g_loss.to(torch.float32)))[None]
        mask = tensors[-1] if ctx.has_mask else None
        use_g = ctx.mode != 0 and g_out is not None
        if not use_g and g_loss is None:
            return None, None, None, None, None, g_embed, None
        if g_loss is not None and ctx.loss_scale != 1.0:
            g_loss = g_loss * ctx.loss_scale
        if ctx.gather:
            gx = L.route_bwd_gather(x, tensors[1], tensors[2], L.rows_contiguous(g_out) if use_g else None, g_loss, mask, ctx.mode if use_g else 0,
                                    masked_rows=ctx.fill)
        else:
            gx = L.route_bwd(x, tensors[1], L.rows_contiguous(g_out) if use_g else None, g_loss, mask, ctx.mode if use_g else 0,
                             masked_rows=ctx.fill)
        return gx, None, None, None, None, g_embed, None


def other_float_dtypes_as_fp32(forward):
    """The kernels take float32 and bfloat16 rows.  The reference's codebook computes in float32 whatever comes in (`x.float()`,
    vqp.py:690) and hands `quantize` back in the input's dtype (`.type(dtype)`, :1178): float16 / float64 inputs do exactly that here,
    around the whole forward."""
    import functools

    @functools.wraps(forward)
    def wrapped(self, *args, **kwargs):
        x = args[0] if args else kwargs.get("x")
        if torch.is_tensor(x) and x.is_floating_point() and x.dtype not in (torch.float32, torch.bfloat16):
            if args:
                args = (x.float(), *args[1:])
            else:
                kwargs = dict(kwargs, x=x.float())
            out = forward(self, *args, **kwargs)
            if not isinstance(out, tuple):
                return out
            # (quantize, indices, loss[, all_codes of the residual modules | LossBreakdown]): the code tensors go back to x's dtype
            back = lambda t: t.to(x.dtype) if torch.is_tensor(t) and t.is_floating_point() and t.ndim >= 3 else t
            return (out[0].to(x.dtype), *out[1:3], *(back(t) for t in out[3:]))
        return forward(self, *args, **kwargs)
    return wrapped


class _L2NormFn(torch.autograd.Function):
    """l2norm of the input of a cosine-similarity codebook (vqp.py:37-38 at :1159) as ONE kernel each way: forward vq_l2norm_kernel
    (bit for bit F.normalize, bf16 rounding included), backward vq_l2norm_bwd_kernel (autograd's F.normalize gradient, recomputed from
    x -- the normalised rows are not saved for it).  F.normalize itself: a reduction and three elementwise kernels forward, about ten
    backward, each over the N x D tensor."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return L.l2norm_rows(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return L.l2norm_rows_bwd(x, g)


def _l2norm_input(x):
    if L.l2norm_rows_supported(x) and os.environ.get("VQHIP_L2NORM_FN", "1") != "0":
        return _L2NormFn.apply(x) if (x.requires_grad and torch.is_grad_enabled()) else L.l2norm_rows(x)
    return F.normalize(x, p=2, dim=-1, eps=1e-6)


class _RowsOfChannelFirstFn(torch.autograd.Function):
    """[b, n, d] rows of a channel-first input (a transposed VIEW of [b, d, n]: channel_last = False, feature maps; vqp.py:1136-1147) as
    one tiled transposing copy each way (vq_transpose_kernel); the gradient goes back as a view of a contiguous [b, d, n] tensor, the
    layout the caller's tensor has."""

    @staticmethod
    def forward(ctx, xt):
        return L.transpose_rows(xt)

    @staticmethod
    def backward(ctx, g):
        return L.transpose_rows(g.contiguous().transpose(1, 2)).transpose(1, 2)


def _rows_of(x):
    """x.contiguous() for a [b, n, d] view whose last dim is strided"""
    if L.is_transposed_view(x) and os.environ.get("VQHIP_TRANSPOSE", "1") != "0":
        return _RowsOfChannelFirstFn.apply(x) if (x.requires_grad and torch.is_grad_enabled()) else L.transpose_rows(x)
    return x.contiguous()


class _CodesOfIndicesFn(torch.autograd.Function):
    """codes[idx] for a codebook that receives gradients (learnable_codebook / vq_bridge; vqp.py:710, 766).  The VALUE is the gather the
    search already wrote (`q`, rows of the codebook as it was searched); the gradient to the codebook is the per-code sum of the
    incoming rows -- the EMA statistics' counting sort + segmented sum (vqhip_ema_accumulate), instead of F.embedding's backward behind
    an extra gather and two elementwise passes in forward."""

    @staticmethod
    def forward(ctx, codes, idx, q):
        ctx.save_for_backward(idx)
        ctx.C, ctx.dt = codes.shape[0], codes.dtype
        return q.view_as(q)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        _, esum = L.ema_accumulate(L.rows_contiguous(g), idx.reshape(-1), ctx.C)
        return esum.to(ctx.dt), None, None


class _RouteFn(torch.autograd.Function):
    """routed value (straight-through / rotation trick) with gradient to x only -- the target is detached inside the
    reference's formula as well (vqp.py:292-316); HIP kernels vq_route_kernel fwd / bwd."""

    @staticmethod
    def forward(ctx, x, q, mode):
        ctx.mode = mode
        ctx.save_for_backward(x, q)
        return L.route_fwd(x, q, mode)

    @staticmethod
    def backward(ctx, g):
        x, q = ctx.saved_tensors
        return L.route_bwd(x, q, L.rows_contiguous(g), None, None, ctx.mode), None, None


class _CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(dist, codes, ignore_index=-1) (vqp.py:1242-1256) without the [N, C] score tensor.  Forward: the exact sweep
    with a streaming log-sum-exp epilogue (vqhip_scores_lse): loss = mean over the rows with codes >= 0 of lse - dist[code].
    Backward: (softmax - onehot) / count pushed through dist's closed-form gradients (those of _ScoresFn), the scores recomputed over
    row chunks of at most 128 MiB -- peak memory no longer grows with N x C.
    `embed_at_search` is the codebook the scores were computed from; `embed` is the live tensor (the gradient goes to it).  They
    differ for an EMA codebook, which is rewritten in place between forward and backward -- and the reference's cdist backward then
    reads the REWRITTEN buffer next to the saved distances (its saved tensor aliases `self.embed`, whose .data the EMA step copies
    into).  Replicated: the softmax weights come from the scores at search time, the GEMM operand is the live codebook."""

    CHUNK_BYTES = 1 << 27

    @staticmethod
    def forward(ctx, x, embed, embed_at_search, codes, cosine, count=None):
        """count (optional 0-dim tensor): the number of valid targets the mean runs over when this call covers only a part of
        them (one head of a multi-headed module: the reference takes ONE mean over all heads, vqp.py:1242-1256)"""
        e = embed_at_search.detach().float().contiguous()
        lse, ts, _ = L.scores_lse(x.detach(), L.pack_codebook(e), e, codes, cosine=cosine, skip_l2norm=True)
        valid = codes.reshape(lse.shape) >= 0
        count = valid.sum() if count is None else count
        ctx.cosine = cosine
        ctx.save_for_backward(x, embed, e, codes, count)
        return torch.where(valid, lse - ts, torch.zeros_like(lse)).sum() / count

    @staticmethod
    def backward(ctx, g):
        x, embed, e_search, codes, count = ctx.saved_tensors
        need_x, need_e = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        D, C = x.shape[-1], embed.shape[0]
        rows, tg = x.detach().reshape(-1, D), codes.reshape(-1)
        ef = embed.detach().float()
        packed = L.pack_codebook(e_search)
        gx = torch.empty(rows.shape, dtype=torch.float32, device=x.device) if need_x else None
        ge = torch.zeros(embed.shape, dtype=torch.float32, device=x.device) if need_e else None
        scale = (g / count).to(torch.float32)
        step = max(1, _CrossEntropyFn.CHUNK_BYTES // (4 * C))
        for i in range(0, rows.shape[0], step):
            xc, tc = rows[i:i + step], tg[i:i + step]
            dist, _, _ = L.scores(xc, packed, e_search, cosine=ctx.cosine, skip_l2norm=True)      # [rows, C], as in the forward
            p = dist.softmax(dim=-1)
            ok = tc >= 0
            p.scatter_add_(1, tc.clamp(min=0)[:, None], -ok.to(p.dtype)[:, None])                  # - onehot (no host sync)
            p = torch.where(ok[:, None], p, torch.zeros_like(p)) * scale                           # d loss / d dist
            xf = xc.float()
            if ctx.cosine:                                                                         # dist = x . e
                if need_x:
                    gx[i:i + step] = p @ ef
                if need_e:
                    ge += p.t() @ xf
            else:                                                                                  # dist = -|x - e|, zero where clamped
                d = -dist
                w = torch.where(d > 1.0001e-4, p / d, torch.zeros_like(p))
                if need_x:
                    gx[i:i + step] = w @ ef - w.sum(-1, keepdim=True) * xf
                if need_e:
                    ge -= w.sum(0)[:, None] * ef - w.t() @ xf
        return (gx.reshape(x.shape).to(x.dtype) if need_x else None, ge.to(embed.dtype) if need_e else None, None, None, None, None)


class _ScoresFn(torch.autograd.Function):
    """dist [b, n, C] = -cdist(x, embed) (Euclidean) or x . embed^T (cosine, x already unit-norm): forward on the HIP
    kernel (vqhip_scores, the reference's rounding sequence), backward = the closed-form gradients of vqp.py:58-62 /
    :741 as two GEMMs:  d(-d_ij)/dx_i = (c_j - x_i) / d_ij,  d(-d_ij)/dc_j = (x_i - c_j) / d_ij  (zero where the
    clamp(min=1e-8) is active).  Only the rare options that read the whole row use it."""

    @staticmethod
    def forward(ctx, x, embed, cosine, embed_at_search=None):
        """embed_at_search: a snapshot of the codebook to compute `dist` from.  The EMA fold rewrites `embed` in place right after the
        search, and like the reference's autograd graph (cdist saves the buffer, `.data.copy_` rewrites it) the backward below reads
        the LIVE embed next to the distances of the search; a forward that is re-run later (the position slices are recomputed in
        backward, torch.utils.checkpoint) must see the codebook the search saw."""
        e = (embed if embed_at_search is None else embed_at_search).detach().float().contiguous()
        dist, _, _ = L.scores(x.detach(), L.pack_codebook(e), e, cosine=cosine, skip_l2norm=True)
        ctx.cosine = cosine
        ctx.save_for_backward(x, embed, dist)
        return dist

    @staticmethod
    def backward(ctx, g):
        x, embed, dist = ctx.saved_tensors
        xf, ef = x.float().reshape(-1, x.shape[-1]), embed.float()
        g2 = g.reshape(-1, g.shape[-1]).float()
        if ctx.cosine:
            gx, ge = g2 @ ef, g2.t() @ xf
        else:
            d = -dist.reshape(g2.shape)
            w = torch.where(d > 1.0001e-4, g2 / d, torch.zeros_like(g2))        # sqrt(1e-8) = 1e-4: clamp active -> zero gradient
            gx = w @ ef - w.sum(-1, keepdim=True) * xf
            ge = w.sum(0)[:, None] * ef - w.t() @ xf
            ge = -ge
        gx = gx.reshape(x.shape).to(x.dtype) if ctx.needs_input_grad[0] else None
        ge = ge.to(embed.dtype) if ctx.needs_input_grad[1] else None
        return gx, ge, None, None


class VectorQuantize(nn.Module):
    def __init__(
        self,
        dim,
        codebook_size,
        codebook_dim=None,
        heads=1,
        separate_codebook_per_head=False,
        decay=0.8,
        eps=1e-5,
        freeze_codebook=False,
        kmeans_init=False,
        kmeans_iters=10,
        sync_kmeans=True,
        use_cosine_sim=False,
        layernorm_after_project_in=False,
        threshold_ema_dead_code=0,
        channel_last=True,
        accept_image_fmap=False,
        accept_3d_fmap=False,
        commitment_weight=1.,
        commitment_use_cross_entropy_loss=False,
        orthogonal_reg_weight=0.,
        orthogonal_reg_active_codes_only=False,
        orthogonal_reg_max_codes=None,
        codebook_diversity_loss_weight=0.,
        codebook_diversity_temperature=100.,
        stochastic_sample_codes=False,
        sample_codebook_temp=1.,
        straight_through=False,
        rotation_trick=None,
        directional_reparam=False,
        directional_reparam_variance=5e-3,
        sync_codebook=None,
        sync_affine_param=False,
        ema_update=None,
        vq_bridge: Optional[nn.Module] = None,
        manual_ema_update=False,
        learnable_codebook=None,
        in_place_codebook_optimizer: Optional[Callable] = None,
        manual_in_place_optimizer_update=False,
        affine_param=False,
        affine_param_batch_decay=0.99,
        affine_param_codebook_decay=0.9,
        sync_update_v=0.,
        return_zeros_for_masked_padding=True,
        route_gradients_to_input=True,
        shard_codebook=False,
    ):
        """Constructor keywords = the reference's (vqp.py:803-849) plus ONE extension: `shard_codebook=True` partitions the
        codebook over the ranks of the default process group (rank p owns codes [p C/P, (p+1) C/P), BASELINE config 4): every
        rank searches its shard with the HIP kernels and ONE RCCL all-reduce(MAX) of packed (score, index) keys picks the
        global winner with the reference's tie rule (parallel.ShardedVectorQuantize).  EMA codebooks with heads == 1 only."""
        super().__init__()

        # derived defaults, as vqp.py:854-856
        ema_update = (not directional_reparam and vq_bridge is None) if ema_update is None else ema_update
        learnable_codebook = (directional_reparam or vq_bridge is not None) if learnable_codebook is None else learnable_codebook
        rotation_trick = (not directional_reparam and dim > 1) if rotation_trick is None else rotation_trick

        if affine_param:
            assert not use_cosine_sim, 'affine param is only compatible with euclidean codebook'
        dense_options = (commitment_use_cross_entropy_loss or codebook_diversity_loss_weight > 0. or stochastic_sample_codes
                         or straight_through)
        # (round 5: several heads -- one shared codebook on [(b h), n, d] rows or one codebook per head, vqp.py:1044-1049 -- combine with
        #  the options that read whole score rows, with affine_param and with learnable codebooks like the reference's einsums do;
        #  forward(topk=) with heads > 1 fails in the reference itself and raises here)

        # the reference's own cross-flag checks (vqp.py:884, 898-913), for identical error behaviour
        assert not (use_cosine_sim and learnable_codebook), 'cosine sim distance codebook not compatible with learnable codebook yet'
        assert sum(map(int, (straight_through, rotation_trick, directional_reparam))) <= 1
        assert not (straight_through and learnable_codebook), 'gumbel straight through not allowed when learning the codebook'
        assert not (directional_reparam and threshold_ema_dead_code == 0), 'periodic dead code replacement should be enabled when differential reparam method is turned on'
        assert not (ema_update and learnable_codebook), 'learnable codebook not compatible with EMA update'
        assert not (vq_bridge is not None and not learnable_codebook), 'learnable_codebook must be set to True if vq_bridge is passed in'
        assert not (vq_bridge is not None and ema_update), 'ema_update must be False if vq_bridge is passed in'
        assert 0 <= sync_update_v <= 1.
        assert not (sync_update_v > 0. and not learnable_codebook), 'learnable codebook must be turned on'
        has_orth = orthogonal_reg_weight > 0.

        self.dim = dim
        self.heads = heads
        self.separate_codebook_per_head = separate_codebook_per_head
        codebook_dim = dim if codebook_dim is None else codebook_dim
        codebook_input_dim = codebook_dim * heads
        requires_projection = codebook_input_dim != dim

        if requires_projection:
            proj = [nn.Linear(dim, codebook_input_dim)]
            if layernorm_after_project_in:
                proj.append(nn.LayerNorm(codebook_input_dim))
            self.project_in = proj[0] if len(proj) == 1 else nn.Sequential(*proj)
            self.project_out = nn.Linear(codebook_input_dim, dim)
        else:
            self.project_in = nn.Identity()
            self.project_out = nn.Identity()
        self.has_projections = requires_projection

        self.eps = eps
        self.has_commitment_loss = commitment_weight > 0. and not directional_reparam
        self.commitment_weight = commitment_weight
        self.learnable_codebook = learnable_codebook
        self.has_codebook_orthogonal_loss = has_orth
        self.orthogonal_reg_weight = orthogonal_reg_weight
        self.orthogonal_reg_active_codes_only = orthogonal_reg_active_codes_only
        self.orthogonal_reg_max_codes = orthogonal_reg_max_codes
        self.directional_reparam = directional_reparam
        self.directional_reparam_variance = directional_reparam_variance
        self.sync_update_v = sync_update_v
        self.commitment_use_cross_entropy_loss = commitment_use_cross_entropy_loss
        self.has_codebook_diversity_loss = codebook_diversity_loss_weight > 0.
        self.codebook_diversity_loss_weight = codebook_diversity_loss_weight
        self.codebook_diversity_temperature = codebook_diversity_temperature
        self.stochastic_sample_codes = stochastic_sample_codes
        self.gumbel_straight_through = straight_through
        self.rotation_trick = rotation_trick
        self.route_gradients_to_input = route_gradients_to_input
        self.use_cosine_sim = use_cosine_sim
        self.codebook_size = codebook_size
        self.accept_image_fmap = accept_image_fmap
        self.accept_3d_fmap = accept_3d_fmap
        self.channel_last = channel_last
        self.return_zeros_for_masked_padding = return_zeros_for_masked_padding
        self.freeze_codebook = freeze_codebook

        if sync_codebook is None:
            sync_codebook = _is_distributed()

        self._codebook = Codebook(
            dim=codebook_dim,
            num_codebooks=heads if separate_codebook_per_head else 1,
            codebook_size=codebook_size,
            kmeans_init=kmeans_init,
            kmeans_iters=kmeans_iters,
            sync_kmeans=sync_kmeans,
            decay=decay,
            eps=eps,
            threshold_ema_dead_code=threshold_ema_dead_code,
            use_ddp=sync_codebook,
            sample_codebook_temp=sample_codebook_temp,
            ema_update=ema_update,
            manual_ema_update=manual_ema_update,
            use_cosine_sim=use_cosine_sim,
            learnable_codebook=has_orth or learnable_codebook,           # vqp.py:939
            vq_bridge=vq_bridge,
            affine_param=affine_param,
            sync_affine_param=sync_affine_param,
            affine_param_batch_decay=affine_param_batch_decay,
            affine_param_codebook_decay=affine_param_codebook_decay,
        )
        self._sharded = None
        if shard_codebook:
            bad = [n for n, v in dict(heads=heads > 1, kmeans_init=kmeans_init, learnable_codebook=learnable_codebook or has_orth,
                                      affine_param=affine_param, vq_bridge=vq_bridge is not None,
                                      threshold_ema_dead_code=threshold_ema_dead_code > 0, dense_options=dense_options,
                                      directional_reparam=directional_reparam, accept_fmap=accept_image_fmap or accept_3d_fmap,
                                      in_place_codebook_optimizer=in_place_codebook_optimizer is not None,
                                      ema_update_off=not ema_update).items() if v]
            if bad:
                raise NotImplementedError(f"shard_codebook=True supports plain EMA codebooks only (unsupported here: {', '.join(bad)})")
            from .parallel import ShardedVectorQuantize
            self._sharded = ShardedVectorQuantize(codebook_dim, codebook_size, use_cosine_sim=use_cosine_sim, decay=decay, eps=eps,
                                                  commitment_weight=commitment_weight, rotation_trick=rotation_trick,
                                                  route_gradients_to_input=route_gradients_to_input,
                                                  init_embed=self._codebook.embed,     # same init as the unsharded module
                                                  register_codebook=False)
            # this rank's shard, registered ONCE (here): state_dict keys `_codebook.*` as usual, holding the shard's rows.  A
            # checkpoint of the whole codebook loads too: _load_from_state_dict below keeps the rows this rank owns.
            self._codebook = self._sharded._codebook
        self.in_place_codebook_optimizer = in_place_codebook_optimizer(self._codebook.parameters()) \
            if in_place_codebook_optimizer is not None else None
        self.manual_in_place_optimizer_update = manual_in_place_optimizer_update
        self.register_buffer('zero', torch.tensor(0.), persistent=False)

    # ---- reference-compatible accessors (vqp.py:978-1022) -----------------------------------------
    @property
    def ema_update(self):
        return self._codebook.ema_update

    @property
    def codebook(self):
        cb = self._codebook.embed
        return cb if self.separate_codebook_per_head else cb[0]

    @codebook.setter
    def codebook(self, codes):
        if not self.separate_codebook_per_head:
            codes = codes[None]
        self._codebook.embed.copy_(codes)

    def _from_rows_layout(self, t):
        if not self.channel_last or self.accept_image_fmap or self.accept_3d_fmap:
            t = t.movedim(-1, 1)
        return t

    def get_codes_from_indices(self, indices):
        cb = self.codebook
        if cb.ndim == 2:
            codes = L.decode_sum(indices[..., None], cb.contiguous())
        else:   # separate codebook per head: indices [b, ..., h]
            parts = [L.decode_sum(indices[..., h:h + 1].contiguous(), cb[h].contiguous()) for h in range(cb.shape[0])]
            codes = torch.cat(parts, -1)
        return self._from_rows_layout(codes)

    def get_output_from_indices(self, indices):
        codes = self.get_codes_from_indices(indices)
        if isinstance(self.project_out, nn.Identity):
            return codes
        if not self.channel_last or self.accept_image_fmap or self.accept_3d_fmap:
            return self.project_out(codes.movedim(1, -1)).movedim(-1, 1)
        return self.project_out(codes)

    def update_in_place_optimizer(self):                                     # vqp.py:1024-1042
        if self.in_place_codebook_optimizer is None:
            return
        if self._codebook.use_ddp:
            for param in self._codebook.parameters():
                if param.grad is not None:
                    dist.all_reduce(param.grad)
                    param.grad /= dist.get_world_size()
        self.in_place_codebook_optimizer.step()
        self.in_place_codebook_optimizer.zero_grad()

    # ---- codebooks that receive gradients (learnable / orthogonal reg / vq_bridge / in-place optimizer) ----
    # The nearest-code search still runs on the HIP kernel (no gradient flows through an argmin); what changes is
    # that `quantize` is a differentiable gather of the (possibly bridged) codebook parameter and that the losses
    # are built from autograd ops.  Reference: vqp.py:710-717 (learnable embed / bridge), :1186-1237.
    def _forward_general(self, xs, rmask, freeze_codebook, kw, *, dense, topk, temp, need_dist=False, transform_fn=None):
        """everything that is not the fused hot path: codebooks with gradients and/or options that read the full score row"""
        cb = self._codebook
        embed_eff = cb.embed if cb.vq_bridge is None else cb.vq_bridge(cb.embed)      # [1, C, D]
        if not cb.learnable_codebook:
            embed_eff = embed_eff.detach()
        temp = cb.sample_codebook_temp if temp is None else temp
        topk_only = (topk is not None and not self.commitment_use_cross_entropy_loss and not self.has_codebook_diversity_loss
                     and not self.gumbel_straight_through and not (self.training and self.stochastic_sample_codes and temp > 0)
                     and not need_dist)

        def search(update_usage=True):
            if transform_fn is not None:
                # QINCo (vqp.py:729-738, 754-776): the callable turns the codebook [h, c, d] into one codebook per row,
                # [h, b, n, c, d]; the search runs on the HIP row-wise kernel (nothing to keep resident: it streams the N x C x D
                # codes once), `quantize` is a differentiable gather from the transformed codes, and an EMA codebook is updated
                # from the chosen indices as usual (vqp.py:783-784)
                if not cb._is_initted():
                    cb.init_embed_(xs.detach().reshape(1, -1, xs.shape[-1]).float(), None if rmask is None else rmask.reshape(1, -1))
                te = transform_fn(embed_eff.to(xs.dtype))                                  # [1, b, n, c, d]
                te = te.reshape(*xs.shape[:-1], te.shape[-2], te.shape[-1])
                if cb.use_cosine_sim:
                    te = F.normalize(te, p=2, dim=-1, eps=1e-6)                            # l2norm (vqp.py:37-38)
                ind = L.assign_rowwise(xs, te, cosine=cb.use_cosine_sim)
                q = te.gather(-2, ind[..., None, None].expand(*ind.shape, 1, te.shape[-1]))[..., 0, :]
                if self.training and update_usage and not freeze_codebook:
                    cb.update_indices(xs.detach(), ind, mask=rmask, ema_update_weight=kw.get("ema_update_weight"),
                                      accum_ema_update=kw.get("accum_ema_update", False), ema_update=kw.get("ema_update"))
                return q.to(xs.dtype), ind, None
            if not dense:
                r = cb.quantize(xs.detach(), mask=rmask, embed_override=embed_eff, update_usage=update_usage, **kw)
                # value: the rows of the PRE-update codebook (the EMA fold inside quantize() runs after the gather, like
                # vqp.py:766 vs :783); gradient: that of a gather from the parameter (vqp.py:710, 766)
                if not (embed_eff.requires_grad and torch.is_grad_enabled()):
                    return r["q"], r["idx"], None
                # (affine_param: quantize() searched and gathered the codes mapped onto the batch's moments, vqp.py:721-724; the
                #  parameter's gradient passes through that map -- a per-dimension scale, the moments are buffers)
                codes = cb.affine_codes(embed_eff) if cb.affine_param else embed_eff
                if codes.shape[0] > 1:      # one codebook per head (xs [h, b, n, d]): the same gather, head by head
                    if xs.dtype in (torch.float32, torch.bfloat16) and xs.shape[-1] <= 512 and os.environ.get("VQHIP_GATHER_FN", "1") != "0":
                        q_h = [_CodesOfIndicesFn.apply(codes[h], r["idx"][h], r["q"][h]) for h in range(codes.shape[0])]
                    else:
                        q_h = []
                        for h in range(codes.shape[0]):
                            g = F.embedding(r["idx"][h], codes[h]).to(xs.dtype)
                            q_h.append(r["q"][h] + (g - g.detach()))
                    return torch.stack(q_h), r["idx"], None
                if xs.dtype in (torch.float32, torch.bfloat16) and xs.shape[-1] <= 512 and os.environ.get("VQHIP_GATHER_FN", "1") != "0":
                    return _CodesOfIndicesFn.apply(codes[0], r["idx"], r["q"]), r["idx"], None
                g = F.embedding(r["idx"], codes[0]).to(xs.dtype)
                return r["q"] + (g - g.detach()), r["idx"], None
            if not cb._is_initted():
                Hc = embed_eff.shape[0]
                cb.init_embed_(xs.detach().reshape(Hc, -1, xs.shape[-1]).float(), None if rmask is None else rmask.reshape(1, -1).expand(Hc, -1))
            codes_eff = embed_eff
            if cb.affine_param:                                     # vqp.py:705-706, 721-724: moments first, then the mapped codebook
                Hc = embed_eff.shape[0]
                cb.update_affine(xs.detach().reshape(Hc, -1, xs.shape[-1]).float(), cb.embed,      # (every call of the codebook,
                                 None if rmask is None else rmask.reshape(1, -1).expand(Hc, -1).bool())   # like Codebook.quantize)
                codes_eff = cb.affine_codes(embed_eff)
            if topk_only and L.topk_supported(xs, topk, embed_eff.shape[-2]):
                # top-k is the only consumer of the score row: the K best codes straight from the sweep (vqhip_topk), no N x C tensor
                e2 = codes_eff[0].detach().float().contiguous()
                ind = L.topk(xs.detach(), L.pack_codebook(e2), e2.shape[0], topk, cosine=cb.use_cosine_sim, skip_l2norm=True)
                q = F.embedding(ind, codes_eff[0])
                return q.to(xs.dtype), ind, None
            want_div = self.training and self.has_codebook_diversity_loss
            # one codebook per head: xs [h, b, n, d], embed_eff [h, C, D] -- every head is searched by itself; a shared codebook
            # (heads folded into the batch axis, [(b h), n, d]) is one "head" here
            sep = xs.ndim == 4
            xs_h = list(xs.unbind(0)) if sep else [xs]
            E_h = list(codes_eff.unbind(0)) if sep else [codes_eff[0]]
            assert len(xs_h) == len(E_h)

            def rows_to_codes(xc, E, E_search=None):                                  # xc [b, n', d] -> q, ind, dist, batch-mean softmax
                dist = _ScoresFn.apply(xc, E, cb.use_cosine_sim, E_search)            # vqp.py:740-743, [b, n', C]
                logits = dist
                if self.training and self.stochastic_sample_codes and temp > 0:       # vqp.py:117-119, 132-133
                    u = torch.zeros_like(dist).uniform_(0, 1)
                    logits = dist / temp - torch.log((-torch.log(u.clamp(min=1e-20))).clamp(min=1e-20))
                ind = logits.topk(topk, dim=-1).indices if topk is not None else logits.argmax(dim=-1)   # vqp.py:137-140
                q = F.embedding(ind, E)               # materialised before any EMA fold below touches the codebook
                if self.gumbel_straight_through and temp > 0 and self.training:       # vqp.py:144-148: one_hot + pi - pi.detach()
                    assert topk is None
                    pi = (dist / temp).softmax(dim=-1)
                    q = q + (pi - pi.detach()) @ E.detach()
                avg = None
                if want_div:                                                          # vqp.py:1287-1292, 67-68: softmax averaged over
                    avg = (dist * self.codebook_diversity_temperature).softmax(dim=-1).mean(dim=0)    # the batch, per position
                return q, ind, dist, avg

            def entropy_sum(avgs):                    # the averaged distribution runs over batch AND heads ('... n l -> n l', vqp.py:1290)
                if not want_div:
                    return xs.new_zeros((), dtype=torch.float32)
                avg = avgs[0] if len(avgs) == 1 else torch.stack(avgs).mean(0)
                return (-avg * torch.log(avg.clamp(min=1e-5))).sum()                  # summed over this chunk's positions

            # The options that get here read whole score rows (gumbel noise, the straight-through softmax, the diversity loss'
            # batch-averaged softmax).  `dist` is N x C floats -- 4 GiB at BASELINE cfg 2 -- so beyond SCORE_CHUNK_BYTES it is
            # produced for a slice of the positions at a time (all batch entries -- and heads -- of a position stay together: the
            # diversity loss averages over them), consumed, and recomputed in backward (checkpoint): peak memory = one slice, not
            # N x C.  Callers that want the matrix itself (top-k next to other options, the dense cross-entropy of the in-place-
            # optimizer / transform paths) keep the single call.
            nb, npos, C_ = xs_h[0].shape[0], xs_h[0].shape[1], E_h[0].shape[0]
            pos_per = max(1, SCORE_CHUNK_BYTES() // (4 * C_ * max(nb, 1)))
            if topk is None and not need_dist and npos > pos_per:
                from torch.utils.checkpoint import checkpoint
                grad = torch.is_grad_enabled() and (xs.requires_grad or embed_eff.requires_grad)
                E_search = [E.detach().clone() for E in E_h] if grad else [None] * len(E_h)   # (the recomputation in backward runs after the EMA fold)
                qs, inds, ent_sum = [[] for _ in E_h], [[] for _ in E_h], xs.new_zeros((), dtype=torch.float32)
                for n0 in range(0, npos, pos_per):
                    avgs = []
                    for h, (xh, E) in enumerate(zip(xs_h, E_h)):
                        xc = xh[:, n0:n0 + pos_per].contiguous()
                        if grad:
                            def piece(a, e_, h=h):                                    # (e_: makes the codebook an input of the checkpoint)
                                q_, ind_, _, avg_ = rows_to_codes(a, e_, E_search[h])
                                return q_, ind_, (avg_ if avg_ is not None else a.new_zeros(()))
                            q_, ind_, avg_ = checkpoint(piece, xc, E, use_reentrant=False)
                        else:
                            q_, ind_, _, avg_ = rows_to_codes(xc, E)
                        qs[h].append(q_); inds[h].append(ind_); avgs.append(avg_)
                    ent_sum = ent_sum + entropy_sum(avgs)
                q_h, ind_h, dist_h = [torch.cat(v, 1) for v in qs], [torch.cat(v, 1) for v in inds], None
            else:
                outs_h = [rows_to_codes(xh, E) for xh, E in zip(xs_h, E_h)]
                q_h, ind_h = [o[0] for o in outs_h], [o[1] for o in outs_h]
                dist_h = [o[2] for o in outs_h]
                ent_sum = entropy_sum([o[3] for o in outs_h])
            q = torch.stack(q_h) if sep else q_h[0]
            ind = torch.stack(ind_h) if sep else ind_h[0]
            dist = None if dist_h is None else (torch.stack(dist_h) if sep else dist_h[0])
            self.__dict__["_diversity_from_search"] = (-(ent_sum / npos)) if want_div else None   # -ent.mean() over the positions
            if self.training and update_usage and not freeze_codebook and topk is None:                 # vqp.py:783-784
                cb.update_indices(xs.detach(), ind, mask=rmask, ema_update_weight=kw.get("ema_update_weight"),
                                  accum_ema_update=kw.get("accum_ema_update", False), ema_update=kw.get("ema_update"))
            return q.to(xs.dtype), ind, dist

        quantize, embed_ind, distances = search()
        inplace_loss = self.zero
        if self.in_place_codebook_optimizer is not None and self.training and not freeze_codebook:   # vqp.py:1186-1210
            if rmask is not None:
                l = F.mse_loss(quantize, xs.detach(), reduction='none')[rmask].mean()
            else:
                l = F.mse_loss(quantize, xs.detach())
            l.backward()
            if not self.manual_in_place_optimizer_update:
                self.update_in_place_optimizer()
            inplace_loss = l
            embed_eff = cb.embed if cb.vq_bridge is None else cb.vq_bridge(cb.embed)
            quantize, embed_ind, distances = search(update_usage=False)

        commit_quantize = quantize
        if self.training:
            if not self.learnable_codebook or freeze_codebook:
                commit_quantize = quantize.detach()
            src = xs if topk is None else xs[..., None, :].expand(*xs.shape[:-1], topk, xs.shape[-1])   # vqp.py:1220-1221
            if xs.requires_grad and self.route_gradients_to_input and torch.is_grad_enabled():
                if self.rotation_trick:
                    quantize = _RouteFn.apply(src.contiguous(), quantize.detach().contiguous(), L.ROTATION)   # target detached in the formula (vqp.py:292-316)
                elif self.directional_reparam:                                        # vqp.py:323-330
                    err = quantize - src
                    nrm = err.norm(dim=-1, keepdim=True)
                    noised = err + (self.directional_reparam_variance ** 0.5) * torch.randn_like(err)
                    quantize = src + F.normalize(noised, p=2, dim=-1, eps=1e-6).detach() * nrm
                else:
                    quantize = src + (quantize - src).detach()
            if self.sync_update_v > 0.:                                               # vqp.py:1235-1237
                quantize = quantize + self.sync_update_v * (quantize - quantize.detach())
        return quantize, embed_ind, commit_quantize, inplace_loss, distances

    def _masked_commit_loss(self, commit_quantize, orig_input, mask):
        """The commitment loss of a padded batch as the reference computes it (vqp.py:1317-1326): the squared error against
        `orig_input` -- the tensor the CALLER passed (vqp.py:1108), before the layout change, the projection, the head split and the
        cosine l2norm -- broadcast against the rows' [b, n, d] / [1, (b h), n, d] / [h, b, n, d], then the mean over the valid rows.
        For heads == 1 without a projection that is the plain masked mean.  With several heads or a projection the two shapes only
        broadcast when codebook_dim == dim (every head is then compared with the unprojected input) and raise otherwise; both are
        the reference's behaviour, reproduced by running its own two ops."""
        cq = commit_quantize
        if self.heads > 1 and not self.separate_codebook_per_head:
            cq = cq[None]                                                   # rows [(b h), n, d]: the reference's '1 (b h) n d'
        try:                # the two shapes must broadcast (fail with the reason, not with an opaque broadcast error deep in ATen)
            torch.broadcast_shapes(tuple(cq.shape), tuple(orig_input.shape))
        except RuntimeError:
            raise RuntimeError(
                f"masked commitment loss: the reference compares the codes {tuple(cq.shape)} with the tensor the caller passed "
                f"{tuple(orig_input.shape)} (vector_quantize_pytorch.py:1319: F.mse_loss(commit_quantize, orig_input)) and these do not "
                "broadcast -- with several heads or a projection a padded training batch only has a loss when codebook_dim == dim (and, for "
                "one codebook shared by the heads, batch size 1); the reference raises here too (INTEGRATION.md, behaviour notes)") from None
        if tuple(cq.shape) != tuple(orig_input.shape):       # (they broadcast: exactly the reference's op, its UserWarning about it silenced)
            cq, orig_input = torch.broadcast_tensors(cq, orig_input)
        err = F.mse_loss(cq, orig_input, reduction='none')
        loss_mask = mask
        if self.heads > 1:                                                  # 'b n -> c (b h) n' (vqp.py:1323)
            b, n = mask.shape
            loss_mask = mask[None, :, None, :].expand(err.shape[0], b, err.shape[1] // b, n).reshape(err.shape[0], -1, n)
        return err[loss_mask].mean()

    def _split_heads(self, x):
        if self.heads == 1:
            return x
        b, n, _ = x.shape
        xh = x.reshape(b, n, self.heads, -1)
        if self.separate_codebook_per_head:
            return xh.permute(2, 0, 1, 3).contiguous()                   # h b n d
        return xh.permute(0, 2, 1, 3).reshape(b * self.heads, n, -1)    # (b h) n d

    def expire_codes_(self, x):
        x = self._codebook.transform_input(x)
        x = self._split_heads(x)
        self._codebook.expire_codes_(x if x.ndim == 4 else x[None])

    def update_indices(self, x, indices, mask=None):
        x = self._to_rows_layout(x, check_mask=mask)
        x = self.project_in(x)
        x = self._split_heads(x)
        x = self._codebook.transform_input(x)
        if self.heads > 1:
            b = indices.shape[0]
            if self.separate_codebook_per_head:
                indices = indices.reshape(b, -1, self.heads).permute(2, 0, 1)
            else:
                indices = indices.reshape(b, -1, self.heads).permute(0, 2, 1).reshape(b * self.heads, -1)
        else:
            indices = indices.reshape(x.shape[0], -1)
        self._codebook.update_indices(x, indices, mask=mask)

    update_ema_indices = update_indices

    def _to_rows_layout(self, x, check_mask=None):
        if self.accept_image_fmap or self.accept_3d_fmap:
            assert check_mask is None
            return x.flatten(2).transpose(1, 2)
        if not self.channel_last:
            return x.transpose(1, 2)
        return x

    # ---- forward ----------------------------------------------------------------------------------
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """shard_codebook=True: a checkpoint of the WHOLE codebook (e.g. the reference's, or an unsharded module's) loads as well --
        every rank keeps the rows [lo, hi) it owns.  (A shard-sized checkpoint loads as is.)"""
        if self._sharded is not None:
            sh = self._sharded
            for k in ("initted", "embed", "embed_avg", "cluster_size"):
                # (checkpoints written while the shard was registered under `_sharded` as well carry both key sets: keep one)
                legacy = state_dict.pop(f"{prefix}_sharded._codebook.{k}", None)
                if legacy is not None and f"{prefix}_codebook.{k}" not in state_dict:
                    state_dict[f"{prefix}_codebook.{k}"] = legacy
            for k in ("embed", "embed_avg", "cluster_size"):
                key = f"{prefix}_codebook.{k}"
                t = state_dict.get(key)
                if t is not None and t.ndim >= 2 and t.shape[1] == sh.codebook_size and sh.codebook_size != sh.hi - sh.lo:
                    state_dict[key] = t[:, sh.lo:sh.hi]
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @other_float_dtypes_as_fp32
    def forward(
        self,
        x,
        indices=None,
        mask=None,
        lens=None,
        topk=None,
        sample_codebook_temp=None,
        freeze_codebook=None,
        return_loss_breakdown=False,
        codebook_transform_fn: Optional[Callable] = None,
        ema_update_weight=None,
        accum_ema_update=False,
        ema_update=None,
    ):
        if codebook_transform_fn is not None:
            if self.heads > 1 or self._sharded is not None or self._codebook.affine_param:
                raise NotImplementedError("codebook_transform_fn: heads == 1, unsharded, no affine_param")
            if (indices is not None or topk is not None or self.commitment_use_cross_entropy_loss or self.has_codebook_diversity_loss
                    or self.stochastic_sample_codes or self.gumbel_straight_through or self.in_place_codebook_optimizer is not None):
                raise NotImplementedError("codebook_transform_fn cannot be combined with options that read the whole score row "
                                          "(indices=, topk=, cross-entropy / diversity losses, gumbel sampling) or the in-place optimizer")
        if self._sharded is not None:
            if any(v is not None for v in (indices, mask, lens, topk, sample_codebook_temp, ema_update_weight, ema_update)) or accum_ema_update \
                    or return_loss_breakdown or x.ndim != 3:
                raise NotImplementedError("shard_codebook=True: forward(x [b, n, d]) only")
            L._need_gpu(x)
            xs = x if self.channel_last else x.transpose(1, 2)
            q, ind, loss = self._sharded(self.project_in(xs))
            q = self.project_out(q)
            return (q if self.channel_last else q.transpose(1, 2)), ind, loss
        if topk is not None and self.heads > 1:
            raise NotImplementedError("forward(topk=) is implemented for heads == 1 only")
        L._need_gpu(x)
        return_loss = indices is not None
        orig_input = x
        freeze_codebook = self.freeze_codebook if freeze_codebook is None else freeze_codebook

        assert not (mask is not None and lens is not None)
        if lens is not None:                                                   # vqp.py:108-110
            mask = torch.arange(x.shape[1], device=x.device)[None, :] < lens[:, None]

        only_one = x.ndim == 2
        if only_one:
            assert mask is None
            x = x[:, None, :]

        spatial = x.shape[2:] if (self.accept_image_fmap or self.accept_3d_fmap) else None
        x = self._to_rows_layout(x, check_mask=mask)
        if not x.is_contiguous() and x.stride(-1) != 1:
            x = _rows_of(x)                                                   # channel-first callers: one transposing copy
        x = self.project_in(x)
        b, n = x.shape[0], x.shape[1]
        xs = self._split_heads(x)                                             # [b,n,d] | [(b h),n,d] | [h,b,n,d]
        if not xs.is_contiguous() and xs.stride(-1) != 1:
            xs = xs.contiguous()

        rmask = mask
        if mask is not None and self.heads > 1:
            rmask = mask if self.separate_codebook_per_head else mask[:, None, :].expand(b, self.heads, n).reshape(b * self.heads, n)

        # cosine: gradients must flow through the l2norm (vqp.py:1159), so when the input needs grad the
        # normalisation stays an autograd op and the kernel is told the rows are already unit-norm.
        pre_normalized = False
        needs_grad = self.training and xs.requires_grad and torch.is_grad_enabled()
        # options that read the WHOLE score row as a matrix (top-k, the diversity loss' batch-averaged softmax, gumbel noise / its
        # straight-through softmax) materialise `dist`; the cross-entropy losses alone do not: they stream a log-sum-exp (_CrossEntropyFn)
        need_matrix = (topk is not None or self.has_codebook_diversity_loss or self.stochastic_sample_codes or self.gumbel_straight_through)
        # (with an in-place optimizer the second search runs on the stepped codebook and the reference recomputes `dist` from it,
        #  vqp.py:1186-1210: that case keeps the dense path, which does the same)
        # (likewise affine_param: the scores are those of the codebook mapped onto THIS batch's moments, which the search updates)
        ce_only = ((return_loss or self.commitment_use_cross_entropy_loss) and not need_matrix and codebook_transform_fn is None
                   and self.in_place_codebook_optimizer is None and not self._codebook.affine_param)
        dense = need_matrix or ((return_loss or self.commitment_use_cross_entropy_loss) and not ce_only)
        # A codebook that receives gradients, in the plain training configuration (input with grad, routed output, MSE commitment):
        # the only gradient that reaches the codes is the commitment loss', which is a function of the EMA statistics of x -- the
        # search runs as on the hot path and _QuantizeFn.backward adds one statistics pass (no N x D autograd glue)
        cb_ = self._codebook
        learn_fast = ((cb_.learnable_codebook or cb_.vq_bridge is not None) and needs_grad and self.route_gradients_to_input
                      and not self.directional_reparam and not dense and not ce_only and codebook_transform_fn is None
                      and self.sync_update_v == 0. and self.in_place_codebook_optimizer is None and self.has_commitment_loss
                      and not freeze_codebook and not cb_.affine_param and not cb_.use_ddp and cb_.embed.dtype == torch.float32
                      and xs.dtype in (torch.float32, torch.bfloat16) and xs.shape[-1] <= 512 and xs.is_cuda
                      # (a masked cosine batch: the reference's loss compares the un-detached gather against the ORIGINAL input,
                      #  vqp.py:1214, 1319 -- not the squared error the statistics pass sums, so the codes' gradient takes param_path)
                      and not (self.use_cosine_sim and mask is not None)
                      and not (self.heads > 1 and self.separate_codebook_per_head)
                      and os.environ.get("VQHIP_LEARN_FAST", "1") != "0")
        param_path = not learn_fast and (self._codebook.learnable_codebook or self._codebook.vq_bridge is not None or self.directional_reparam
                                         or dense or ce_only or codebook_transform_fn is not None)
        # (every branch of the autograd-glue path reads xs itself -- commit loss, update_indices, init_embed_, assign_rowwise -- so it
        #  always gets the normalised rows, input with or without grad: the reference normalises first, vqp.py:1159)
        if self.use_cosine_sim and (needs_grad or param_path or (mask is not None and self.training)):
            xs = _l2norm_input(xs)
            pre_normalized = True

        self.__dict__.pop("_diversity_from_search", None)        # (never a value left behind by a forward that raised)
        kw = dict(freeze_codebook=freeze_codebook, ema_update_weight=ema_update_weight,
                  accum_ema_update=accum_ema_update, ema_update=(ema_update if topk is None else False),
                  input_normalized=pre_normalized)
        inplace_loss = orth_loss = diversity_loss = self.zero
        distances = ce_embed = None
        mask_filled = 0
        if param_path:
            if ce_only:     # the codebook the search is about to use, live and as a snapshot (the EMA fold inside the search rewrites
                cb0 = self._codebook                                                  # `embed` in place; see _CrossEntropyFn)
                if not cb0._is_initted():     # k-means init runs BEFORE the reference computes `dist` (vqp.py:718-720): snapshot after it
                    Hc = cb0.num_codebooks                                # (one codebook per head: xs is [h, b, n, d], the mask [b, n])
                    cb0.init_embed_(xs.detach().reshape(Hc, -1, xs.shape[-1]).float(),
                                    None if rmask is None else rmask.reshape(1, -1).expand(Hc, -1))
                ce_embed = cb0.embed if cb0.vq_bridge is None else cb0.vq_bridge(cb0.embed)        # [H, C, D]
                if not cb0.learnable_codebook:
                    ce_embed = ce_embed.detach()
                ce_embed_at_search = ce_embed.detach().clone()
            quantize, embed_ind, commit_quantize, inplace_loss, distances = self._forward_general(
                xs, rmask, freeze_codebook, kw, dense=dense, topk=topk, temp=sample_codebook_temp, need_dist=return_loss and dense,
                transform_fn=codebook_transform_fn)
        else:
            fold = (mask is None and self.training and self.has_commitment_loss)     # sq_sum then already is the mean
            embed_param = None
            if learn_fast:
                embed_param = cb_.embed if cb_.vq_bridge is None else cb_.vq_bridge(cb_.embed)              # [H, C, D], requires grad
                if not self.learnable_codebook:      # (orthogonal regularisation alone makes the CODEBOOK learnable, vqp.py:939, but the
                    embed_param = embed_param.detach()   # commitment loss detaches `quantize` unless the module's own flag is set, :1214)
            # a padded batch on the plain layout (rows = the caller's tensor): the padding rows of output and indices are written by
            # the node itself (vqhip_mask_fill_rows, padding rows only) instead of two torch.where passes over the batch at the end
            wants = xs.requires_grad and torch.is_grad_enabled()
            if (mask is not None and xs is orig_input and self.heads == 1 and topk is None and not only_one
                    and (not wants or (needs_grad and self.route_gradients_to_input)) and os.environ.get("VQHIP_MASK_FILL", "1") != "0"):
                mask_filled = 2 if self.return_zeros_for_masked_padding else 1
            quantize, embed_ind, sq_sum = _QuantizeFn.apply(xs, self, rmask, kw, 1.0 / float(max(xs.numel(), 1)) if fold else 1.0, embed_param,
                                                            mask_filled)

        # ---- loss (vqp.py:1282-1348) --------------------------------------------------------------
        # the reference's loss starts as a leaf that requires grad in training (vqp.py:1282).  Here: a cached CONSTANT zero per
        # (device, inference mode) -- no fill kernel per forward, and no shared autograd leaf (a cached leaf would accumulate .grad
        # across backward() calls and be shared between concurrent forwards); `requires_grad` is restored on the result below
        if self.training:
            key = (x.device, torch.is_inference_mode_enabled())
            cache = self.__dict__.setdefault("_loss_anchor", {})
            anchor = cache.get(key)
            if anchor is None:
                anchor = cache[key] = torch.zeros((), device=x.device, dtype=torch.float32)
            loss = anchor
        else:
            anchor = None
            loss = torch.zeros((), device=x.device, dtype=torch.float32)
        commit_loss = self.zero
        def ce_loss(codes):                                                          # vqp.py:1242-1256
            if distances is not None:                                                # (dense path; vqp.py:1244-1254)
                if self.heads == 1:
                    return F.cross_entropy(distances.permute(0, 2, 1), codes, ignore_index=-1)
                if not self.separate_codebook_per_head:                              # '1 (b h) n l -> b l n h': rows [(b h), n], codes [b, n, h]
                    codes_bh = codes.permute(0, 2, 1).reshape(distances.shape[0], distances.shape[1])
                    return F.cross_entropy(distances.permute(0, 2, 1), codes_bh, ignore_index=-1)
                return F.cross_entropy(distances.permute(1, 3, 2, 0), codes, ignore_index=-1)   # 'c b n l -> b l n c'

            if self.heads == 1:
                return _CrossEntropyFn.apply(xs, ce_embed[0], ce_embed_at_search[0], codes, self.use_cosine_sim)
            # multi-headed: codes [b, n, h]; the reference takes one mean over every (row, head) target
            if not self.separate_codebook_per_head:                                  # rows [(b h), n, d], one codebook
                codes_bh = codes.permute(0, 2, 1).reshape(xs.shape[0], xs.shape[1])
                return _CrossEntropyFn.apply(xs, ce_embed[0], ce_embed_at_search[0], codes_bh, self.use_cosine_sim)
            total = (codes >= 0).sum()                                               # rows [h, b, n, d], one codebook per head
            return sum(_CrossEntropyFn.apply(xs[h], ce_embed[h], ce_embed_at_search[h], codes[..., h].contiguous(), self.use_cosine_sim, total)
                       for h in range(self.heads))

        if return_loss:                                                              # vqp.py:1260-1261: an early return -- `quantize` still in
            if self.heads > 1 and not self.separate_codebook_per_head:               # the codebook's layout, for a shared codebook
                quantize = quantize[None]                                            # [1, (b h), n, d]
            return quantize, ce_loss(indices)

        if self.training and param_path:
            if self.has_codebook_diversity_loss:                                     # vqp.py:1287-1292, 67-68: computed next to the
                diversity_loss = self.__dict__.pop("_diversity_from_search", None)     # scores it reads (_forward_general: rows_to_codes)
                if diversity_loss is None:
                    raise NotImplementedError("codebook_diversity_loss_weight > 0 together with an option whose search does not read the "
                                              "score rows (codebook_transform_fn): not on the MI355X path")
                loss = loss + diversity_loss * self.codebook_diversity_loss_weight
            if self.has_commitment_loss:
                if self.commitment_use_cross_entropy_loss:                           # vqp.py:1297-1305
                    codes = embed_ind
                    if self.heads > 1:                                               # rows layout -> the reference's [b, n, h]
                        codes = (embed_ind.permute(1, 2, 0) if self.separate_codebook_per_head
                                 else embed_ind.reshape(b, self.heads, n).permute(0, 2, 1))
                    if mask is not None:
                        codes = codes.masked_fill(~(mask if self.heads == 1 else mask[..., None]), -1)
                    commit_loss = ce_loss(codes)
                elif topk is not None:                                               # vqp.py:1307-1315
                    rep = orig_input[..., None, :].expand(*orig_input.shape[:-1], topk, orig_input.shape[-1])
                    commit_loss = F.mse_loss(commit_quantize, rep, reduction='none').mean(dim=-1)
                    if mask is not None:
                        mk = mask.reshape(*mask.shape, *([1] * (commit_loss.ndim - mask.ndim)))
                        commit_loss = torch.where(mk, commit_loss, torch.zeros_like(commit_loss))
                elif mask is not None:
                    commit_loss = self._masked_commit_loss(commit_quantize, orig_input, mask)
                else:
                    commit_loss = F.mse_loss(commit_quantize, xs)
                term = commit_loss * self.commitment_weight
                if term.ndim > 0:      # (top-k on bf16 rows: the reference adds this [b, n, k] bf16 tensor to its fp32 `loss` of shape [1],
                    term = term.float()   # vqp.py:1282, 1329 -- two dimensioned tensors promote to fp32; `loss` here is 0-dim, which would not)
                loss = loss + term
        elif self.training and self.has_commitment_loss:
            d = xs.shape[-1]
            if mask is None:
                commit_loss = sq_sum                                  # mean((q - x)^2): the 1 / numel is folded into the reduction
            elif not self.use_cosine_sim and self.heads == 1 and xs.shape == orig_input.shape:
                commit_loss = sq_sum / (rmask.sum() * d).to(torch.float32)
            else:
                # reference quirk (vqp.py:1319): the masked loss compares against the caller's tensor -- not l2-normalised, not
                # projected, not split into heads (this path's codebook is not learnable: commit_quantize is detached, vqp.py:1214)
                commit_loss = self._masked_commit_loss(quantize.detach(), orig_input, mask)
            # (`loss` still is the constant zero here and 0 + v == v bit for bit: the term itself, not an add kernel per forward)
            loss = commit_loss if self.commitment_weight == 1. else commit_loss * self.commitment_weight

        if self.training and self.has_codebook_orthogonal_loss:                                     # vqp.py:1331-1348, 340-345
            codebook = self._codebook.embed
            if self.orthogonal_reg_active_codes_only:
                ids = torch.unique(embed_ind)
                if mask_filled:             # the node has already written -1 into the padding rows (vqhip_mask_fill_rows); the reference
                    ids = ids[ids >= 0]     # takes `unique` before its where(mask, ind, -1), vqp.py:1336 vs :1386-1394
                codebook = codebook[:, ids]
            ncodes = codebook.shape[-2]
            if self.orthogonal_reg_max_codes is not None and ncodes > self.orthogonal_reg_max_codes:
                codebook = codebook[:, torch.randperm(ncodes, device=x.device)[:self.orthogonal_reg_max_codes]]
            normed = F.normalize(codebook, p=2, dim=-1, eps=1e-6)
            cos = torch.einsum('hid,hjd->hij', normed, normed)
            h_, n_ = codebook.shape[:2]
            orth_loss = (cos ** 2).sum() / (h_ * n_ ** 2) - (1 / n_)
            loss = loss + orth_loss * self.orthogonal_reg_weight

        if self.training:
            if loss is anchor:
                loss = loss.clone()                  # no loss term was added: hand out a fresh tensor, not the cached constant
            if torch.is_grad_enabled() and not loss.requires_grad:
                loss = loss.detach().requires_grad_(True)        # as the reference's loss (a view: no kernel), a fresh leaf per forward

        # ---- indices / quantized back to the caller's layout (vqp.py:1265-1396) -------------------
        if self.heads > 1:
            if self.separate_codebook_per_head:
                embed_ind = embed_ind.permute(1, 2, 0)
                quantize = quantize.permute(1, 2, 0, 3).reshape(b, n, -1)
            else:
                embed_ind = embed_ind.reshape(b, self.heads, n).permute(0, 2, 1)
                quantize = quantize.reshape(b, self.heads, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
        if spatial is not None:
            embed_ind = embed_ind.reshape(b, *spatial, *embed_ind.shape[2:])
        if only_one:
            embed_ind = embed_ind[:, 0]

        quantize = self.project_out(quantize)
        if spatial is not None:
            quantize = quantize.transpose(1, 2).reshape(b, -1, *spatial)
        elif not self.channel_last:
            quantize = quantize.transpose(1, 2)
        if only_one:
            quantize = quantize[:, 0]

        if mask is not None and not mask_filled:
            fill = torch.zeros_like(orig_input) if self.return_zeros_for_masked_padding else orig_input
            if topk is not None:
                fill = fill[..., None, :]
            mq = mask.reshape(*mask.shape, *([1] * (quantize.ndim - mask.ndim)))
            quantize = torch.where(mq, quantize, fill)
            m = mask.reshape(*mask.shape, *([1] * (embed_ind.ndim - mask.ndim)))
            embed_ind = torch.where(m, embed_ind, torch.full_like(embed_ind, -1))

        if not return_loss_breakdown:
            return quantize, embed_ind, loss
        return quantize, embed_ind, loss, LossBreakdown(commit_loss, diversity_loss, orth_loss, inplace_loss)
