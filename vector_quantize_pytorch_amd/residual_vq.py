"""`ResidualVQ` / `GroupedResidualVQ` for MI355X (reference: residual_vq.py:166-724).

Same constructor / forward keywords and outputs `(quantized_out, indices [b, n, Q], losses [Q])`,
same state_dict keys (`layers.{i}._codebook.*`, aliased under shared_codebook).  The per-quantizer
loop stays on the device: no host synchronisation happens between stages (the only `.item()` of the
reference -- the quantize-dropout seed, rvq.py:96-102 -- is taken only when quantize_dropout is on).
"""
from __future__ import annotations

import contextlib
import os
import random
from math import ceil
from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import _lib as L
from .vector_quantize import VectorQuantize, _rows_of, other_float_dtypes_as_fp32


def _round_up(n, m):
    return ceil(n / m) * m


def _draw_seed(device, need_value: bool, max_size=10_000):
    """Consumes the device RNG exactly like get_maybe_sync_seed (rvq.py:96-102); syncs only if asked."""
    r = torch.randint(0, max_size, (), device=device)
    if not need_value:
        return None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(r)
    return int(r.item())


class MLP(nn.Module):
    """QINCo's code transform (rvq.py:107-162; Huijben et al., arXiv 2401.14732): every code of a quantizer's codebook is
    rewritten as a function of the reconstruction so far.  codes [c, d] (or [h, c, d]), condition [b, n, d] (or [b, d]) ->
    [b, n, c, d] ([h, b, n, c, d]).  Same parameter names as the reference (`proj_in`, `layers.{i}.0/2`): state_dicts load.
    Plain PyTorch: this is the user-side network, not the search; the search over its output is vqhip_assign_rowwise."""

    def __init__(self, dim, dim_hidden=None, depth=4, l2norm_output=False):
        super().__init__()
        dim_hidden = dim if dim_hidden is None else dim_hidden
        self.proj_in = nn.Linear(2 * dim, dim)
        self.layers = nn.ModuleList([nn.Sequential(nn.Linear(dim, dim_hidden), nn.SiLU(), nn.Linear(dim_hidden, dim)) for _ in range(depth)])
        self.l2norm_output = l2norm_output

    def forward(self, codes, *, condition):
        one_headed = codes.ndim == 2
        if one_headed:
            codes = codes[None]
        cond = condition.reshape(condition.shape[0], -1, condition.shape[-1])                  # [b, n, d] ([b, d] -> n = 1)
        h, c, b, n = codes.shape[0], codes.shape[-2], cond.shape[0], cond.shape[1]
        both = torch.cat((cond[None, :, :, None, :].expand(h, b, n, c, -1), codes[:, None, None].expand(h, b, n, c, -1)), dim=-1)
        x = self.proj_in(both)
        for layer in self.layers:
            x = layer(x) + x
        if self.l2norm_output:
            x = torch.nn.functional.normalize(x, dim=-1)
        return x[0] if one_headed else x


_GROUPS_CONCURRENT = [False]      # True while GroupedResidualVQ.forward runs its groups on side streams: the groups interleave with each
                                  # other already, so their residual chains are not split into row chunks as well (a context, not an
                                  # attribute written onto the child modules: a child used on its own keeps its chunking)


class ResidualVQ(nn.Module):
    concurrent_stats = True       # fused loop: stage statistics on a side HIP stream beside the later searches (class attribute)
    chunk_rows = True             # fused loop: big batches run as interleaved row chunks (L.rvq_row_chunks)

    def __init__(
        self,
        *,
        dim,
        num_quantizers: Optional[int] = None,
        codebook_size,
        codebook_dim=None,
        shared_codebook=False,
        diveq=False,
        heads=1,
        quantize_dropout=False,
        quantize_dropout_cutoff_index=0,
        quantize_dropout_multiple_of=1,
        accept_image_fmap=False,
        implicit_neural_codebook=False,
        mlp_kwargs: dict = dict(),
        beam_size=None,
        eval_beam_size=None,
        beam_score_quantizer_weights=None,
        quant_grad_frac=0.,
        **vq_kwargs,
    ):
        super().__init__()
        assert heads == 1, 'residual vq is not compatible with multi-headed codes'
        assert num_quantizers is not None or isinstance(codebook_size, tuple)
        assert not (eval_beam_size is not None and beam_size is None)
        if implicit_neural_codebook and (beam_size is not None):
            raise NotImplementedError("implicit_neural_codebook with beam search")

        codebook_dim = dim if codebook_dim is None else codebook_dim
        self.codebook_dim = codebook_dim
        requires_projection = codebook_dim != dim
        self.project_in = nn.Linear(dim, codebook_dim) if requires_projection else nn.Identity()
        self.project_out = nn.Linear(codebook_dim, dim) if requires_projection else nn.Identity()
        self.has_projections = requires_projection
        self.accept_image_fmap = accept_image_fmap
        self.implicit_neural_codebook = implicit_neural_codebook
        if implicit_neural_codebook:                          # rvq.py:207-211
            vq_kwargs.update(learnable_codebook=True, ema_update=False)
        self.diveq = diveq
        if diveq:                                             # rvq.py:226-232: DiVeQ learns the codebook through the reparametrised output
            vq_kwargs.update(ema_update=False, learnable_codebook=True, route_gradients_to_input=False, commitment_weight=0.)

        if shared_codebook:                                   # rvq.py:213-217
            vq_kwargs.update(manual_ema_update=True, manual_in_place_optimizer_update=True)

        sizes = codebook_size if isinstance(codebook_size, tuple) else (codebook_size,) * num_quantizers
        num_quantizers = len(sizes) if num_quantizers is None else num_quantizers
        assert len(sizes) == num_quantizers
        self.num_quantizers = num_quantizers
        self.codebook_sizes = sizes
        self.uniform_codebook_size = len(set(sizes)) == 1

        self.layers = nn.ModuleList([
            VectorQuantize(dim=codebook_dim, codebook_size=c, codebook_dim=codebook_dim,
                           accept_image_fmap=accept_image_fmap, **vq_kwargs) for c in sizes])
        assert all(not vq.has_projections for vq in self.layers)

        self.quantize_dropout = quantize_dropout and num_quantizers > 1
        assert quantize_dropout_cutoff_index >= 0
        self.quantize_dropout_cutoff_index = quantize_dropout_cutoff_index
        self.quantize_dropout_multiple_of = quantize_dropout_multiple_of
        self.vq_is_ema_updating = self.layers[0].ema_update
        assert not (self.vq_is_ema_updating and diveq), 'Only one of ema_update or self.diveq must be used for updating the codebook'
        self.quant_grad_frac = quant_grad_frac if not diveq else 1.
        self.beam_size = beam_size
        self.eval_beam_size = beam_size if eval_beam_size is None else eval_beam_size
        weights = [1.] * num_quantizers if beam_score_quantizer_weights is None else beam_score_quantizer_weights
        assert len(weights) == num_quantizers
        self.register_buffer('beam_score_weights', torch.tensor(weights), persistent=False)
        if implicit_neural_codebook:                          # rvq.py:288-289: one transform per quantizer after the first
            self.mlps = nn.ModuleList([MLP(dim=codebook_dim, l2norm_output=self.layers[0].use_cosine_sim, **mlp_kwargs)
                                       for _ in range(num_quantizers - 1)])
        else:
            self.mlps = (None,) * (num_quantizers - 1)

        self.shared_codebook = shared_codebook
        if shared_codebook:                                   # rvq.py:300-306: every layer aliases layer 0's codebook
            assert self.uniform_codebook_size
            first = self.layers[0]._codebook
            for vq in self.layers[1:]:
                vq._codebook = first

    @property
    def codebook_size(self):
        return self.layers[0].codebook_size

    @property
    def codebooks(self):
        cbs = tuple(layer._codebook.embed[0] for layer in self.layers)
        return torch.stack(cbs) if self.uniform_codebook_size else cbs

    def get_codes_from_indices(self, indices):
        """[b, ..., q] -> [q, b, ..., d]; -1 (dropped-out quantizer) decodes to zeros (rvq.py:324-377)."""
        qdim = indices.shape[-1]
        if qdim < self.num_quantizers:
            assert self.quantize_dropout > 0., 'quantize dropout must be greater than 0 if you wish to reconstruct from a signal with less fine quantizations'
            indices = torch.nn.functional.pad(indices, (0, self.num_quantizers - qdim), value=-1)
        cbs = self.codebooks
        if self.implicit_neural_codebook:                     # rvq.py:347-366: layer by layer, every transform sees the sum so far
            out, so_far = [], 0.
            for q, mlp in enumerate((None, *self.mlps)):
                ind = indices[..., q]
                safe = ind.clamp(min=0)
                if mlp is None:
                    codes = cbs[q][safe]
                else:
                    te = mlp(cbs[q], condition=so_far)                                        # [b, n, c, d]
                    te = te.reshape(*ind.shape, te.shape[-2], te.shape[-1])
                    codes = te.gather(-2, safe[..., None, None].expand(*ind.shape, 1, te.shape[-1]))[..., 0, :]
                out.append(codes)
                so_far = so_far + codes                       # (the reference adds the un-masked code as well, rvq.py:364)
            allc = torch.stack(out)
            return allc.masked_fill((indices == -1).movedim(-1, 0)[..., None], 0.)
        # every stage's gather writes its slice of the stacked result (no torch.stack pass over Q x N x D floats afterwards)
        allc = torch.empty(self.num_quantizers, *indices.shape[:-1], cbs[0].shape[-1], dtype=torch.float32, device=indices.device)
        for q in range(self.num_quantizers):
            L.decode_sum(indices[..., q:q + 1].contiguous(), cbs[q].contiguous(), out=allc[q])
        return allc

    def get_output_from_indices(self, indices):
        qdim = indices.shape[-1]
        if qdim < self.num_quantizers:
            assert self.quantize_dropout > 0., 'quantize dropout must be greater than 0 if you wish to reconstruct from a signal with less fine quantizations'
            indices = torch.nn.functional.pad(indices, (0, self.num_quantizers - qdim), value=-1)
        if self.uniform_codebook_size and not self.implicit_neural_codebook:
            cb = self.layers[0]._codebook.embed[0] if self.shared_codebook else self.codebooks.contiguous()
            summed = L.decode_sum(indices.contiguous(), cb.contiguous())           # one fused gather + sum over q
        else:
            summed = self.get_codes_from_indices(indices).sum(0)
        return self.project_out(summed)

    @other_float_dtypes_as_fp32
    def forward(
        self,
        x,
        mask=None,
        indices=None,
        return_all_codes=False,
        sample_codebook_temp=None,
        freeze_codebook=False,
        beam_size=None,
        rand_quantize_dropout_fixed_seed=None,
    ):
        if indices is not None:
            # (the reference itself cannot run this: rvq.py:493 unpacks three values from a layer call that returns two with indices,
            #  vqp.py:1260-1261 -- v1.31.0 raises "ValueError: not enough values to unpack (expected 3, got 2)"; checked against the live
            #  reference, tests/golden/make_golden.py has no fixture to make)
            raise NotImplementedError("ResidualVQ.forward(indices=): the reference (v1.31.0) raises ValueError here (rvq.py:493 unpacks three "
                                      "values from VectorQuantize.forward(indices=), which returns two); call the layers' forward(indices=) directly")
        L._need_gpu(x)
        beam_size = (self.beam_size if self.training else self.eval_beam_size) if beam_size is None else beam_size
        is_beam = beam_size is not None and beam_size > 1
        Q = self.num_quantizers
        x = self.project_in(x)

        drop_at = None
        if self.training and self.quantize_dropout:                # rvq.py:423-439
            seed = rand_quantize_dropout_fixed_seed
            if seed is None:
                seed = _draw_seed(x.device, need_value=True)
            drop_at = random.Random(seed).randrange(self.quantize_dropout_cutoff_index, Q)
            if self.quantize_dropout_multiple_of != 1:
                drop_at = _round_up(drop_at + 1, self.quantize_dropout_multiple_of) - 1

        # Feature maps (rvq.py: every layer is built with accept_image_fmap and rearranges 'b d h w -> b (h w) d' and back, Q times
        # each way): the residual arithmetic is row-wise, so the rows are formed ONCE (one tiled transposing copy), the stages run on
        # them as for a channel-last input -- on the fused loop -- and output and indices go back to the map's shape as views.
        fmap = None
        if (self.accept_image_fmap and x.ndim >= 4 and not is_beam and not self.has_projections and mask is None
                and self._fused_eligible(x.flatten(2).transpose(1, 2), mask, rows_of_fmap=True)):
            fmap, x_map = x.shape[2:], x
            x = _rows_of(x.flatten(2).transpose(1, 2))
            if self._wants_input_grad(x) and self._route_mode() != 0 and not self._chain_eligible(x, freeze_codebook, routed=True):
                fmap, x = None, x_map                                  # (the per-stage path takes the map itself)

        if is_beam:
            quantized_out, all_indices, all_losses = self._forward_beam(x, mask, sample_codebook_temp, freeze_codebook, beam_size, drop_at)
        elif self._fused_eligible(x, mask, rows_of_fmap=fmap is not None) and not (self._wants_input_grad(x) and self._route_mode() != 0
                                                    and not self._chain_eligible(x, freeze_codebook, routed=True)):
            if self._wants_input_grad(x):
                # the same on-device loop, gradients to the input in closed form (one kernel forward, one backward).  With
                # routed gradients every layer RETURNS the straight-through / rotation-trick value and rvq.py:524 subtracts
                # that from the residual, so the searches run as the chain with route_mode (anything the chain does not
                # cover -- bf16 rows, D = 512 -- took the per-stage path above).
                quantized_out, all_indices, all_losses = _RvqFusedFn.apply(x, self, mask, freeze_codebook, drop_at)
            else:
                quantized_out, all_indices, all_losses = self._forward_fused(x, mask, freeze_codebook, drop_at)
        else:
            quantized_out, all_indices, all_losses = self._forward_staged(x, mask, sample_codebook_temp, freeze_codebook, drop_at)

        if self.diveq:                                        # rvq.py:605-606, vqp.py:323-330 (noise from the device RNG)
            err = quantized_out - x
            noised = err + (5e-3 ** 0.5) * torch.randn_like(err)
            quantized_out = x + torch.nn.functional.normalize(noised, p=2, dim=-1, eps=1e-6).detach() * err.norm(dim=-1, keepdim=True)

        quantized_out = self.project_out(quantized_out)
        if fmap is not None:
            b = quantized_out.shape[0]
            quantized_out = quantized_out.transpose(1, 2).reshape(b, -1, *fmap)
            all_indices = all_indices.reshape(b, *fmap, all_indices.shape[-1])
        ret = (quantized_out, all_indices, all_losses)
        if return_all_codes:
            ret = (*ret, self.get_codes_from_indices(ret[1]))
        return ret

    # ---- fused on-device residual loop (csrc: vq_rvq_kernel) -----------------------------------------
    def _fused_eligible(self, x, mask, rows_of_fmap=False):
        vq0 = self.layers[0]
        cb0 = vq0._codebook
        if vq0.use_cosine_sim or not self.uniform_codebook_size or (self.accept_image_fmap and not rows_of_fmap):
            return False
        if self.codebook_dim % 32 != 0 or x.ndim != 3 or x.dtype not in (torch.float32, torch.bfloat16):
            return False
        if self.codebook_dim > 512:          # (wide dims, csrc/vq_wide.hip: the layers one by one)
            return False
        if self.quant_grad_frac > 0:
            return False
        # The fused loop searches the stored `embed` with a plain argmin under no_grad: only plain EMA / frozen codebooks
        # qualify.  Anything that changes the searched codebook (vq_bridge, affine_param), needs autograd (learnable
        # codebook, in-place optimizer, orthogonal / diversity / cross-entropy losses, DiVeQ / directional reparam) or
        # reads the whole score row (gumbel sampling, gumbel straight-through) takes the per-stage path.
        for layer in self.layers:
            cb = layer._codebook
            if (cb.learnable_codebook or cb.vq_bridge is not None or cb.affine_param
                    or layer.in_place_codebook_optimizer is not None or layer.directional_reparam
                    or layer.stochastic_sample_codes or layer.gumbel_straight_through
                    or layer.commitment_use_cross_entropy_loss or layer.has_codebook_diversity_loss
                    or layer.has_codebook_orthogonal_loss or layer.sync_update_v > 0.):
                return False
        # shared codebook + dead-code replacement: the reference expires inside every stage's update (vqp.py:635-641),
        # which changes `embed` before the next stage's search -- a stage-by-stage dependency the fused loop cannot honour
        if self.shared_codebook and cb0.has_dead_code_replacement and self.training:
            return False
        if mask is not None and not all(layer.return_zeros_for_masked_padding for layer in self.layers):
            return False                                     # (masked rows keep their input: the per-stage path's fill)
        return all(layer._codebook._is_initted() for layer in self.layers)     # k-means runs in the staged path

    def _wants_input_grad(self, x):
        return self.training and x.requires_grad and torch.is_grad_enabled()

    def _route_mode(self):
        """what a layer returns as `quantized` for an input that requires grad (vqp.py:1225-1233)"""
        vq0 = self.layers[0]
        if not vq0.route_gradients_to_input:
            return 0
        return L.ROTATION if vq0.rotation_trick else L.STRAIGHT_THROUGH

    def _update_and_loss(self, freeze_codebook):
        vq0 = self.layers[0]
        update = self.training and not (freeze_codebook or vq0.freeze_codebook) and \
            (vq0._codebook.ema_update or vq0._codebook.has_dead_code_replacement)
        return update, self.training and vq0.has_commitment_loss

    def _chain_eligible(self, x, freeze_codebook, routed=False):
        """the residual chain (csrc: vqhip_assign_screened_chain): fp32 rows, D in {32, 64, 128, 256}; routed (a training step whose
        input requires grad: every stage's input comes from vqhip_route_residual and is searched like a first stage) also bf16 rows
        and D = 512.  The per-stage commitment loss comes from the statistics pass, so a loss without an EMA update has no producer
        there"""
        update, want_loss = self._update_and_loss(freeze_codebook)
        dims = (32, 64, 128, 256, 512) if routed else (32, 64, 128, 256)
        return bool(L.screening_enabled() and self.codebook_dim in dims and L.rvq_chain_supported(x, self.codebook_size, routed=routed)
                    and (update or not want_loss))

    @torch.no_grad()
    def _forward_fused(self, x, mask, freeze_codebook, drop_at, aux=None):
        """aux (dict, filled for _RvqFusedFn): the codebook(s) the search used (a snapshot: the EMA fold below rewrites `embed` in
        place), the number of active stages and d loss_q / d (sum of squared errors of stage q)."""
        Q = self.num_quantizers if drop_at is None else drop_at + 1
        vq0 = self.layers[0]
        D, C = self.codebook_dim, self.codebook_size
        train = self.training
        if self.shared_codebook:
            embed = vq0._codebook.embed[0]
            packed = L.pack_codebook(embed)
        else:
            embed = torch.stack([layer._codebook.embed[0] for layer in self.layers[:Q]]).contiguous()
            packed = L.pack_codebook_batched(embed)            # the Q codebooks in two launches (was 2 Q and a torch.stack)
        update, want_loss = self._update_and_loss(freeze_codebook)
        route_mode = self._route_mode() if aux is not None else 0       # aux: called from _RvqFusedFn (the input requires grad)
        buf = side = None
        if update:
            # statistics of ALL stages in one buffer [Q, C D + C] (embed_sum || count per stage): under data parallelism ONE
            # all-reduce per forward (the reference issues two per stage, vqp.py:603, 607); the per-stage folds stay sequential
            # (a shared codebook is lerp-ed Q times, rvq.py:213-217 + vqp.py:616-617)
            buf = torch.zeros(Q, (C * D + C + 3) // 4 * 4, dtype=torch.float32, device=x.device)   # stage slices stay 16-byte aligned

        # Residual chain (csrc: vqhip_assign_screened_chain): every stage forms its input x_prev - code in its own prologue, so no
        # stage re-reads its input to write a residual; the commitment loss' squared error then comes from the per-stage
        # statistics pass, which reads every row next to its code anyway (needs `update`; without a loss nothing is needed)
        chain = self._chain_eligible(x, freeze_codebook, routed=route_mode != 0)
        assert chain or route_mode == 0, "routed residuals need the chain (ResidualVQ.forward sends the rest to the per-stage path)"
        sq_parts = None
        if chain and want_loss:             # one [Q, P] buffer of loss partials: one batched reduction after the loop
            nrows = x.numel() // D
            sq_parts = torch.empty(Q, L.lib().vqhip_ema_sqerr_partials(nrows, C), dtype=torch.float64, device=x.device)

        # the statistics workspaces of all stages, their histograms zeroed in one launch (a memset per stage queued on the statistics
        # stream showed up as 14 % of the summed kernel time of a cfg-3 profile: it waits there for a workgroup slot)
        stats_ws = L.ema_workspaces(Q, x.numel() // D, C, x.device) if (update and x.is_cuda and x.numel() > 0) else None

        def accumulate(q, stage_input, idx_all):
            kw = dict(row_mask=mask, count=buf[q, C * D: C * D + C], embed_sum=buf[q, : C * D].view(C, D), idx_offset=q, idx_stride=Q)
            if stats_ws is not None and not vq0._codebook.use_cosine_sim:
                kw["ws"] = stats_ws[q]
            if chain and want_loss:
                e_q = embed if self.shared_codebook else embed[q]
                L.ema_accumulate(stage_input, idx_all, C, sqerr_from=(packed if self.shared_codebook else packed[q], e_q),
                                 sqerr_out=sq_parts[q], **kw)
            else:
                L.ema_accumulate(stage_input, idx_all, C, **kw)

        # Batched stage statistics (round 5): on the chain the inputs of stages 1 .. Q - 1 sit one behind the other, so their
        # statistics are ONE set of launches (vqhip_ema_accumulate_stages, grid dimension y = stage) behind the loop, beside the decode,
        # instead of Q chains of four short launches that each wait for a workgroup slot beside a screening kernel filling every
        # register file -- and slow it down (profiles/r5_rvq_cfg3).  VQHIP_RVQ_BATCH_STATS: 0 (default) per-stage passes beside the
        # loop, 1 stage 0 beside the loop + the rest batched, 2 everything behind the loop.  Measured at cfg 3: 2.86 / 2.93 / 2.95 ms
        # (cfg 5: 14.7 / 15.0 / 14.9) -- the loop alone gets 0.35 ms shorter without the statistics beside it, and the batched passes
        # (HBM-bound at 5.1 TB/s: 0.45 ms for the 1.9 GB of stages 1 .. 7) then cost more than that behind it.
        batch_mode = int(os.environ.get("VQHIP_RVQ_BATCH_STATS", "0")) if (chain and update and Q > 1 and x.is_cuda and x.numel() > 0
                                                                            and not vq0._codebook.use_cosine_sim) else 0
        concurrent = bool(update and x.is_cuda and self.concurrent_stats and not torch.cuda.is_current_stream_capturing())
        main = torch.cuda.current_stream(x.device) if x.is_cuda else None
        native = False
        if L.screening_enabled() and D in (32, 64, 128, 256, 512) and x.data_ptr() % 16 == 0:
            # Q screened searches on the f16 MFMA pipe (csrc/vq_screen.hip), each writing the next stage's input; beats
            # the fused exact-fp32 kernel, which keeps the dims the screen does not cover (96, 160, ...)
            hook = None
            if concurrent:
                # no search reads a codebook this forward changes (shared or not, embed is only rewritten after the loop), so
                # stage q's statistics pass runs on a side stream beside the searches of the later stages.  (Not while a HIP graph
                # is being captured: a fork nested inside GroupedResidualVQ's per-group fork crashed hipStreamEndCapture on
                # ROCm 7.2; either fork alone captures fine, the group fork is the one kept.)
                side = _stats_stream(x.device, main)
                side.wait_stream(main)

                def hook(q, stage_input, idx_all, ready=None):
                    if batch_mode == 2 or (batch_mode == 1 and q > 0):
                        return
                    if ready is None:       # one chain on the caller's stream
                        ready = [torch.cuda.Event()]
                        ready[0].record(main)
                    for ev in ready:        # row chunks: one event per chunk stream, recorded behind its stage q
                        side.wait_event(ev)
                    with torch.cuda.stream(side):
                        accumulate(q, stage_input, idx_all)
            # (with a side-stream hook the -1 of the masked rows is written only after that stream has been joined below: its
            #  statistics passes read `idx`)
            if chain:
                # (row chunks interleave two chains so that one's exact passes run beside the other's screening kernel; a grouped
                #  module's groups already do that for each other.  Not under graph capture: nested forks, see above.)
                capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
                no_chunks = capturing or not self.chunk_rows or (_GROUPS_CONCURRENT[0] and os.environ.get("VQHIP_GRVQ_CHUNK_ROWS", "0") != "1")
                K = 1 if no_chunks else L.rvq_row_chunks(x.numel() // D)
                native = (batch_mode == 0 and x.numel() > 0 and not (update and vq0._codebook.use_cosine_sim)
                          and os.environ.get("VQHIP_RVQ_NATIVE", "1") != "0")
                if native:
                    # the whole loop -- searches, routed residuals, per-stage statistics -- as ONE library call
                    # (vqhip_rvq_chain_forward): the same launches on the same streams, issued from C (VQHIP_RVQ_NATIVE=0: from here)
                    # Decode split around the last stage (round 6: stages 0 .. Q - 2 summed on the statistics stream beside the last
                    # stage's search): for per-stage codebooks, whose decode gathers Q code rows per output row from L2; one shared
                    # codebook of <= 1024 codes decodes from LDS in one pass (vq_decode_lds_kernel) and keeps that (VQHIP_CHAIN_DECODE=2
                    # forces the split, 0 turns it off)
                    dmode = os.environ.get("VQHIP_CHAIN_DECODE", "0")       # measured slower (cfg 3 2.73 -> 2.95 ms forced, cfg 5 14.07 -> 14.66): off
                    split_out = None
                    if (aux is None and update and hook is not None and Q >= 2 and mask is None and x.dtype == torch.float32 and route_mode == 0
                            and (dmode == "2" or (dmode == "1" and not (self.shared_codebook and C <= 1024)))):
                        split_out = torch.empty_like(x)
                    r = L.rvq_chain_forward(x, packed, embed, Q, row_mask=mask, route_mode=route_mode, row_chunks=K,
                                            stats=buf if update else None, stats_ws=stats_ws if update else None,
                                            sq_parts=sq_parts if (update and want_loss) else None,
                                            stats_stream=side if (update and hook is not None) else None, decode_out=split_out)
                    r["decoded"] = split_out
                    self.last_counts = r["counts"]      # per stage: (open rows, pair rows) device counters, one per row chunk (diagnostic)
                    if mask is not None and hook is None:
                        L.mask_fill_indices(r["idx"], mask)
                else:
                    r = L.rvq_forward_chained(x, packed, embed, Q, row_mask=mask, stage_hook=hook, fill_masked=hook is None,
                                              route_mode=route_mode, row_chunks=K)
            else:
                r = L.rvq_forward_screened(x, packed, embed, Q, want_resid=update, want_sqerr=want_loss, row_mask=mask, stage_hook=hook,
                                           fill_masked=hook is None)
            if hook is None:
                side = None
            if batch_mode:
                # stages 1 .. Q - 1 (and stage 0 unless the hook took it) in one set of launches, on the statistics stream beside the
                # decode when there is one
                run = contextlib.nullcontext()
                if side is not None:
                    side.wait_stream(main)
                    run = torch.cuda.stream(side)
                with run:
                    if batch_mode == 2 or side is None:
                        accumulate(0, x, r["idx"])
                    L.ema_accumulate_stages(r["bufs"][: Q - 1].view(Q - 1, *x.shape), r["idx"], 1, C, buf[1:Q], stats_ws[1:Q], row_mask=mask,
                                            sqerr_from=None if not want_loss else ((packed, embed) if self.shared_codebook else (packed[1:Q], embed[1:Q])),
                                            sqerr_out=None if not want_loss else sq_parts[1:Q])
        else:
            r = L.rvq_forward(x, packed, embed, Q, want_resid=update, want_sqerr=want_loss, row_mask=mask)
        idx = r["idx"]
        if side is not None and mask is not None:
            torch.cuda.current_stream(x.device).wait_stream(side)       # the side stream's statistics passes have read idx
            L.mask_fill_indices(idx, mask)
        quantized_out = None
        if aux is None:
            quantized_out = r.get("decoded") if r.get("decoded") is not None else L.decode_sum(idx, embed, out_dtype=x.dtype)
        if aux is not None:
            aux["embed"] = embed.clone() if self.shared_codebook else embed      # (torch.stack above already copied)
            aux["Q"] = Q

        stage_in = None
        reduced = False
        if update:
            resid = r.get("resid")
            stage_in = (lambda q: r["inputs"][q]) if r.get("inputs") is not None else (lambda q: resid[..., q, :])
            if side is not None:
                if vq0._codebook.use_ddp:
                    # the ONE all-reduce of the forward, issued on the statistics stream: it queues behind the last stage's
                    # statistics pass and runs beside the decode on the main stream (the fold below waits for both)
                    with torch.cuda.stream(side):
                        dist.all_reduce(buf)
                    reduced = True
                torch.cuda.current_stream(x.device).wait_stream(side)
            elif not batch_mode and not native:
                for q in range(Q):
                    accumulate(q, stage_in(q), idx)

        losses = torch.zeros(self.num_quantizers, device=x.device, dtype=torch.float32)
        if want_loss:
            if chain:
                sums = L.reduce_partials_rows(sq_parts)
            else:
                sums = torch.stack([L.reduce_partials(r["sqerr_partials"][q], r["sqerr_partials"].shape[1], 1.0) for q in range(Q)])
            denom = float(x.numel()) if mask is None else (mask.sum() * D).to(torch.float32)
            losses[:Q] = sums / denom * vq0.commitment_weight
            if aux is not None:
                aux["loss_scale"] = vq0.commitment_weight / denom            # float, or a 0-dim device tensor under a mask
        if train and aux is None:
            losses = torch.zeros(self.num_quantizers, device=x.device, requires_grad=True) + losses   # as vqp.py:1282

        if update:
            if vq0._codebook.use_ddp and not reduced:
                dist.all_reduce(buf)
            cb0 = vq0._codebook
            if self.shared_codebook and cb0.cluster_size.grad is None and cb0.embed_avg.grad is None:
                # the Q folds of the one codebook (stage order, vqp.py:616-617) and its renormalisation (rvq.py:593-598) in one call
                cs, ea, e = cb0._views(0)
                L.ema_fold_many(cs, ea, e, buf, decay=cb0.decay, eps=cb0.eps, cosine=cb0.use_cosine_sim,
                                do_update_ema=bool(self.vq_is_ema_updating))
            else:
                cbs = [self.layers[q]._codebook for q in range(Q)]
                for q, cb in enumerate(cbs):
                    cb._fold_stats(0, buf[q, C * D: C * D + C], buf[q, : C * D].view(C, D), None, False, cb.ema_update)
                if not self.shared_codebook:                            # vqp.py:641: expire_codes_(flatten, seq_mask = mask)
                    # (after all the folds: a layer's expiry reads only its own cluster_size; replacement draws stay in layer order.
                    #  Layers on the reference's control flow share ONE host read of their any(expired) flags.)
                    known = {}
                    host = [q for q, cb in enumerate(cbs) if cb.expiry_reads_host()]
                    if len(host) > 1:
                        known = dict(zip(host, type(cbs[0]).any_expired_many([cbs[q] for q in host])))
                    for q, cb in enumerate(cbs):
                        cb.expire_codes_(stage_in(q).reshape(1, -1, D), seq_mask=None if mask is None else mask.reshape(1, -1).bool(),
                                         any_expired=known.get(q))
                if self.shared_codebook and self.vq_is_ema_updating:    # rvq.py:593-598 (dead-code replacement never gets here:
                    vq0._codebook.update_ema()                           # _fused_eligible sends it to the per-stage path)

        if Q < self.num_quantizers:
            pad = torch.full((*idx.shape[:-1], self.num_quantizers - Q), -1, device=x.device, dtype=torch.long)
            idx = torch.cat((idx, pad), -1)
        return quantized_out, idx, losses

    # ---- beam search over the quantizers (rvq.py:445-590): every stage keeps the `beam_size` best partial code
    #      sequences by accumulated (weighted) negative commit loss; the per-stage top-k comes from the dense-score path ----
    def _forward_beam(self, x, mask, sample_codebook_temp, freeze_codebook, beam_size, drop_at):
        lead = x.shape[:-1]
        J = 1
        scores = torch.zeros(*lead, 1, device=x.device, dtype=x.dtype)
        residual = x[..., None, :]                                  # [b, n, J, d]
        out = torch.zeros_like(residual)
        inputs = torch.empty(*lead, 1, 0, x.shape[-1], device=x.device, dtype=x.dtype)     # [b, n, J, l, d] stage inputs
        idxs = torch.empty(*lead, 1, 0, device=x.device, dtype=torch.long)                  # [b, n, J, l]
        losses = torch.empty(*lead, 1, 0, device=x.device, dtype=torch.float32)             # [b, n, J, l]
        last = (len(self.layers) - 1) if drop_at is None else drop_at

        def pick(t, sel):                                           # t [b, n, J', ...] -> rows `sel` [b, n, keep] of the beam axis
            extra = t.ndim - sel.ndim
            s_ = sel.reshape(*sel.shape, *([1] * extra)).expand(*sel.shape, *t.shape[sel.ndim:])
            return t.gather(sel.ndim - 1, s_)

        for qi, vq in enumerate(self.layers):
            if drop_at is not None and qi > drop_at:
                idxs = torch.nn.functional.pad(idxs, (0, 1), value=-1)
                losses = torch.nn.functional.pad(losses, (0, 1), value=0.)
                continue
            inputs = torch.cat((inputs, residual[..., None, :]), dim=-2)
            quantized, ind, loss = vq(residual, mask=mask, sample_codebook_temp=sample_codebook_temp,
                                      freeze_codebook=freeze_codebook, topk=beam_size)       # [b,n,J,K,d], [b,n,J,K], [b,n,J,K]
            K = ind.shape[-1]
            scores = (scores[..., None] - loss * self.beam_score_weights[qi]).flatten(-2)   # [b, n, J K]
            step = quantized.detach() if self.quant_grad_frac <= 0 else (
                self.quant_grad_frac * quantized + (1. - self.quant_grad_frac) * quantized.detach())
            residual = (residual[..., None, :] - step).flatten(-3, -2)
            out = (out[..., None, :] + quantized).flatten(-3, -2)
            inputs = inputs[..., None, :, :].expand(*inputs.shape[:-2], K, *inputs.shape[-2:]).flatten(-4, -3)
            idxs = torch.cat((idxs[..., None, :].expand(*idxs.shape[:-1], K, idxs.shape[-1]), ind[..., None]), dim=-1).flatten(-3, -2)
            losses = torch.cat((losses[..., None, :].expand(*losses.shape[:-1], K, losses.shape[-1]), loss[..., None].float()), dim=-1).flatten(-3, -2)
            keep = beam_size if qi != last else 1
            if scores.shape[-1] > keep:
                scores, sel = scores.topk(keep, dim=-1)
                residual, out, idxs, losses, inputs = (pick(t, sel) for t in (residual, out, idxs, losses, inputs))

        out, idxs, losses, inputs = out[..., 0, :], idxs[..., 0, :], losses[..., 0, :], inputs[..., 0, :, :]
        if mask is not None:
            losses = torch.where(mask[..., None], losses, torch.zeros_like(losses))
            losses = losses.flatten(0, -2).sum(0) / mask.sum().clamp_min(1e-4)
        else:
            losses = losses.flatten(0, -2).mean(0)
        if self.training:
            # rvq.py:574, 586-589.  REFERENCE QUIRK, replicated for parity: `all_residuals[..., 0, :]` selects stage 0 (not beam 0)
            # of the [.., beam, stage, d] tensor, the following unbind runs over the beam axis (length 1), and the zip therefore
            # stops after the FIRST quantizer: only layer 0 receives EMA statistics after a beam search.
            self.layers[0].update_indices(inputs[..., 0, :], idxs[..., 0], mask=mask)
            if self.shared_codebook:
                shared = self.layers[0]
                if self.vq_is_ema_updating:
                    shared._codebook.update_ema()
                if shared._codebook.has_dead_code_replacement:
                    shared.expire_codes_(inputs.reshape(inputs.shape[0], -1, inputs.shape[-1]))
        return out, idxs, losses

    # ---- per-stage path: autograd to the input, cosine metric, k-means first step, ragged sizes --------
    def _forward_staged(self, x, mask, sample_codebook_temp, freeze_codebook, drop_at):
        quantized_out = torch.zeros_like(x)
        residual = x
        all_idx, all_loss, stage_inputs = [], [], []
        idx_shape = x.shape[:-1] if not self.accept_image_fmap else (x.shape[0], *x.shape[2:])

        transforms = (None, *self.mlps)                             # rvq.py:460-465
        for qi, vq in enumerate(self.layers):                      # rvq.py:469-568
            if drop_at is not None and qi > drop_at:
                all_idx.append(torch.full(idx_shape, -1, device=x.device, dtype=torch.long))
                all_loss.append(torch.zeros((), device=x.device, dtype=torch.float32))
                continue
            if self.shared_codebook and self.training:
                stage_inputs.append(residual.detach())
            fn = None
            if transforms[qi] is not None:                          # QINCo: this stage's codes are a function of the sum so far
                fn = (lambda codes, _m=transforms[qi], _c=quantized_out: _m(codes, condition=_c))
            quantized, ind, loss = vq(residual, mask=mask, sample_codebook_temp=sample_codebook_temp,
                                      freeze_codebook=freeze_codebook, codebook_transform_fn=fn)
            step = quantized.detach() if self.quant_grad_frac <= 0 else (
                self.quant_grad_frac * quantized + (1. - self.quant_grad_frac) * quantized.detach())
            residual = residual - step
            quantized_out = quantized_out + quantized
            all_idx.append(ind)
            all_loss.append(loss)

        if self.training and self.shared_codebook:                 # rvq.py:593-601
            shared = self.layers[0]
            if self.vq_is_ema_updating:
                shared._codebook.update_ema()
            if shared._codebook.has_dead_code_replacement:
                stacked = torch.stack(stage_inputs, -2)
                if self.accept_image_fmap:
                    stacked = torch.stack([s.flatten(2).transpose(1, 2) for s in stage_inputs], -2)
                shared.expire_codes_(stacked.reshape(stacked.shape[0], -1, stacked.shape[-1]))
        return quantized_out, torch.stack(all_idx, -1), torch.stack(all_loss)


class _RvqFusedFn(torch.autograd.Function):
    """ResidualVQ's on-device loop for an input that requires grad.  Forward = the same chained search / statistics / EMA fold as
    the no-grad path, except that every stage's input is the previous one minus the previous layer's ROUTED value (the chain's
    route_mode: rvq.py:524 subtracts `quantized.detach()`, which is the straight-through / rotation-trick value here); the
    output is formed by vq_rvq_route_kernel (sum over the stages of that value, rvq.py:525 + vqp.py:1225-1233); backward = the
    closed form of the reference's graph with quant_grad_frac = 0 (every stage's input is x minus DETACHED values, so
    d r_q / d x = I):
        dL/dx = sum_q J_q^T g_out + sum_q g_loss[q] * commitment_weight * 2 (r_q - c_q) / count
    in one kernel that recomputes r_q from x and the saved indices (nothing per stage is kept)."""

    @staticmethod
    def forward(ctx, x, rvq, mask, freeze_codebook, drop_at):
        aux = {}
        _, idx, losses = rvq._forward_fused(x, mask, freeze_codebook, drop_at, aux=aux)
        mode = rvq._route_mode()
        out = L.rvq_route(x, aux["embed"], idx, aux["Q"], mode, resid_routed=True)
        ctx.mode, ctx.Q, ctx.loss_scale, ctx.has_mask = mode, aux["Q"], aux.get("loss_scale"), mask is not None
        ctx.save_for_backward(x, idx, aux["embed"], *([mask] if mask is not None else []))
        ctx.mark_non_differentiable(idx)
        return out, idx, losses

    @staticmethod
    def backward(ctx, g_out, g_idx, g_losses):
        x, idx, embed = ctx.saved_tensors[:3]
        mask = ctx.saved_tensors[3] if ctx.has_mask else None
        coef = None
        if g_losses is not None and ctx.loss_scale is not None:
            coef = (g_losses[:ctx.Q].to(torch.float32) * ctx.loss_scale).contiguous()
        use_g = ctx.mode != 0 and g_out is not None
        if not use_g and coef is None:
            return None, None, None, None, None
        gx = L.rvq_route(x, embed, idx, ctx.Q, ctx.mode, g_out=L.rows_contiguous(g_out) if use_g else None,
                         loss_coef=coef, row_mask=mask, backward=True, resid_routed=True, loss_only=not use_g)
        return gx, None, None, None, None


class _GrvqFusedFn(torch.autograd.Function):
    """GroupedResidualVQ's batched chain for an input that requires grad (round 6): _RvqFusedFn for all groups at once -- ONE chain with
    routed residuals for the G groups (GroupedResidualVQ._forward_batched), then every group's routed output / gradient written into
    its feature chunk of one tensor by vq_rvq_route_kernel (the groups share nothing: rvq.py:706)."""

    @staticmethod
    def forward(ctx, x, grvq, mask, freeze_codebook):
        aux = {}
        _, idx, losses = grvq._forward_batched(x, mask, freeze_codebook, aux=aux)
        G, D = grvq.groups, grvq.rvqs[0].codebook_dim
        mode = grvq.rvqs[0]._route_mode()
        xc = x if x.is_contiguous() else x.contiguous()
        out = torch.empty_like(xc)
        for g in range(G):
            L.rvq_route(xc[..., g * D:(g + 1) * D], aux["embed"][g], idx[g], aux["Q"], mode, resid_routed=True, out=out[..., g * D:(g + 1) * D])
        ctx.mode, ctx.Q, ctx.G, ctx.D, ctx.loss_scale, ctx.has_mask = mode, aux["Q"], G, D, aux.get("loss_scale"), mask is not None
        ctx.save_for_backward(xc, idx, aux["embed"], *([mask] if mask is not None else []))
        ctx.mark_non_differentiable(idx)
        return out, idx, losses

    @staticmethod
    def backward(ctx, g_out, g_idx, g_losses):
        x, idx, embed = ctx.saved_tensors[:3]
        mask = ctx.saved_tensors[3] if ctx.has_mask else None
        G, D, Q = ctx.G, ctx.D, ctx.Q
        coef = None
        if g_losses is not None and ctx.loss_scale is not None:
            coef = (g_losses[:, :Q].to(torch.float32) * ctx.loss_scale).contiguous()          # [G, Q]
        use_g = ctx.mode != 0 and g_out is not None
        if not use_g and coef is None:
            return None, None, None, None
        if use_g:
            g_out = L.rows_contiguous(g_out)
        gx = torch.empty_like(x)
        for g in range(G):
            sl = slice(g * D, (g + 1) * D)
            L.rvq_route(x[..., sl], embed[g], idx[g], Q, ctx.mode, g_out=g_out[..., sl] if use_g else None,
                        loss_coef=None if coef is None else coef[g], row_mask=mask, backward=True, resid_routed=True, loss_only=not use_g,
                        out=gx[..., sl])
        return gx, None, None, None


_SIDE_STREAMS = {}
_STATS_STREAMS = {}


def _stats_stream(device, main):
    """The statistics stream that pairs with `main` (one per caller stream, so concurrent groups do not share one)."""
    key = (torch.device(device).index, main.cuda_stream)
    while key not in _STATS_STREAMS and len(_STATS_STREAMS) >= L._CACHE_CAP:      # bounded: oldest caller stream out first
        _STATS_STREAMS.pop(next(iter(_STATS_STREAMS)))
    if key not in _STATS_STREAMS:
        # (default priority.  The statistics chain is five short launches behind one another -- memset, histogram, scan, scatter,
        #  segmented sum -- and beside a search that fills every CU each of them waits for a workgroup slot: rocprofv3 shows the
        #  4 KB memset at 124 us wall, the chain at 340 us for ~100 us of work.  A high-priority stream was tried in round 3:
        #  cfg 3 2.996 -> 2.974 ms, cfg 5 15.5 -> 18.9 ms -- the statistics then push the searches of the other groups aside.
        #  Round 4 confined this stream to 16 / 32 / 64 CUs (hipExtStreamCreateWithCUMask through torch.cuda.ExternalStream):
        #  cfg 3 2.93 -> 4.73 / 4.43 / 4.43 ms, cfg 5 15.5 -> 27.1 ms -- a masked queue loses its concurrency with the search.
        #  Also round 4: stage q's pass held back until stage q + 1's screening kernel has finished (an event recorded by the
        #  library between that kernel and its exact passes), so that it runs in the shadow of those short kernels: cfg 3 3.01 ->
        #  3.03 ms, cfg 5 16.0 -> 16.4 ms -- the pass outlasts that shadow and meets the next screen anyway.  And the opposite of the
        #  round-3 experiment, the SEARCH chain on a high-priority stream: cfg 3 2.88 -> 3.06 ms, cfg 5 15.7 -> 20.7 ms.)
        _STATS_STREAMS[key] = torch.cuda.Stream(device=device)
    return _STATS_STREAMS[key]


class GroupedResidualVQ(nn.Module):
    """G independent ResidualVQs on feature chunks (rvq.py:634-724).  Chunks are passed to the kernels as
    strided row views -- no per-group copy of the input."""

    def __init__(self, *, dim, groups=1, accept_image_fmap=False, **kwargs):
        super().__init__()
        self.dim = dim
        self.groups = groups
        assert dim % groups == 0
        self.accept_image_fmap = accept_image_fmap
        self.rvqs = nn.ModuleList([ResidualVQ(dim=dim // groups, accept_image_fmap=accept_image_fmap, **kwargs)
                                   for _ in range(groups)])

    concurrent_groups = True      # one HIP stream per group in forward (class attribute: set False to serialise on the caller's stream)

    def _side_streams(self, device):
        pool = _SIDE_STREAMS.setdefault(torch.device(device).index, [])      # per device, shared by every module: no stream in module state
        while len(pool) < self.groups - 1:
            pool.append(torch.cuda.Stream(device=device))
        return pool

    @property
    def codebooks(self):
        return torch.stack(tuple(r.codebooks for r in self.rvqs))

    @property
    def split_dim(self):
        return 1 if self.accept_image_fmap else -1

    def get_codes_from_indices(self, indices):
        return torch.stack(tuple(r.get_codes_from_indices(i) for r, i in zip(self.rvqs, indices)))

    def get_output_from_indices(self, indices):
        return torch.cat(tuple(r.get_output_from_indices(i) for r, i in zip(self.rvqs, indices)), dim=self.split_dim)

    # ---- the G groups as ONE launch set (round 6; csrc/vq_rvq_chain.hip, vqhip_rvq_chain_t.groups) ----------------------------------
    batched_groups = True         # class attribute; VQHIP_GRVQ_BATCHED=0 keeps the groups on side streams

    def _batched_eligible(self, x, chunks, mask, freeze_codebook):
        """can the groups' residual loops run as one batched chain?  The no-grad on-device loop of every group (ResidualVQ._forward_fused on
        the native chain) with the same stage count: fp32 rows, codebook dim in {32, 64, 128, 256}, plain EMA / frozen codebooks,
        no quantize dropout, no beam search, no projections, channel-last rows."""
        if not (self.batched_groups and os.environ.get("VQHIP_GRVQ_BATCHED", "1") != "0" and x.is_cuda and self.groups > 1
                and x.ndim == 3 and not self.accept_image_fmap and x.numel() > 0 and L.screening_enabled()):
            return False
        if os.environ.get("VQHIP_RVQ_NATIVE", "1") == "0" or int(os.environ.get("VQHIP_RVQ_BATCH_STATS", "0")) != 0:
            return False
        r0 = self.rvqs[0]
        for r, c in zip(self.rvqs, chunks):
            beam = r.beam_size if r.training else r.eval_beam_size
            if (r.has_projections or r.diveq or (beam is not None and beam > 1) or (r.training and r.quantize_dropout)
                    or r.training != r0.training or not r._fused_eligible(c, mask) or r.layers[0]._codebook.use_cosine_sim):
                return False
            # an input that requires grad: the chain with routed residuals (_GrvqFusedFn below), the same mode in every group
            grad = r._wants_input_grad(c)
            if not r._chain_eligible(c, freeze_codebook, routed=grad and r._route_mode() != 0) or r._route_mode() != r0._route_mode():
                return False
            for layer in r.layers:
                cb = layer._codebook
                if cb.cluster_size.grad is not None or cb.embed_avg.grad is not None:
                    return False
        return True

    def _fold_table(self, Q):
        """device table of the (cluster_size, embed_avg, embed) pointers of every layer, group-major -- rebuilt only when a buffer moved"""
        bufs = [r.layers[q]._codebook._views(0) for r in self.rvqs for q in range(Q)]
        key = tuple(t.data_ptr() for b in bufs for t in b)
        cached = getattr(self, "_fold_table_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, L.pointer_table(bufs))
            self._fold_table_cache = cached
        return cached[1]

    @torch.no_grad()
    def _forward_batched(self, x, mask, freeze_codebook, aux=None):
        """rvq.py:634-724 with the loop over the groups (:706) as grid dimension y of ONE residual chain: Q x (screen + exact passes +
        statistics) launches for all groups, the decode of every group straight into its feature chunk of the output (no torch.cat),
        all G x Q EMA folds in three launches.
        aux (dict, filled for _GrvqFusedFn: the input requires grad): every stage's input is the previous one minus the previous layer's
        ROUTED value (the chain's route_mode, as ResidualVQ._forward_fused), no output is decoded here (the caller routes it), and aux
        receives the codebooks the searches used, Q and d loss / d (sum of squared errors)."""
        G, r0 = self.groups, self.rvqs[0]
        Q, D, C = r0.num_quantizers, r0.codebook_dim, r0.codebook_size
        vq0 = r0.layers[0]
        dev = x.device
        x = x if x.is_contiguous() else x.contiguous()
        N = x.numel() // x.shape[-1]
        shared = r0.shared_codebook
        if shared:
            embed = torch.stack([r.layers[0]._codebook.embed[0] for r in self.rvqs])                       # [G, C, D]
            packed = L.pack_codebook_batched(embed)                                                        # [G, P]
        else:
            embed = torch.stack([layer._codebook.embed[0] for r in self.rvqs for layer in r.layers]).view(G, Q, C, D)
            packed = L.pack_codebook_batched(embed.view(G * Q, C, D)).view(G, Q, -1)
        update, want_loss = r0._update_and_loss(freeze_codebook)
        stride = (C * D + C + 3) // 4 * 4
        buf = torch.zeros(G, Q, stride, dtype=torch.float32, device=dev) if update else None
        sq_parts = torch.empty(G, Q, L.lib().vqhip_ema_sqerr_partials(N, C), dtype=torch.float64, device=dev) if (update and want_loss) else None
        stats_ws = L.ema_workspaces(G * Q, N, C, dev).view(G, Q, -1) if update else None
        main = torch.cuda.current_stream(dev)
        capturing = torch.cuda.is_current_stream_capturing()
        side = None
        if update and r0.concurrent_stats and not capturing:
            side = _stats_stream(dev, main)
            side.wait_stream(main)
        env = os.environ.get("VQHIP_GRVQ_CHUNKS")
        K = 1 if capturing else (int(env) if env else L.grvq_row_chunks(N, G))
        out = torch.empty_like(x)
        # the decode split around the last stage: stages 0 .. Q - 2 are summed on the statistics stream beside the last stage's search
        # (cfg 5: four 0.2 ms gathers of eight code rows per output row were the tail of every step), the last stage is added behind it
        # -- built, bit-identical (test_residual_chain_decode_split_...), and MEASURED SLOWER: cfg 5 14.07 -> 14.66 ms (the gathers of 7 x 4
        # stages beside the last screening kernel outlast it and hold back the last stage's statistics behind them), cfg 3 2.73 -> 2.95.
        # Off by default (VQHIP_CHAIN_DECODE=1 / 2 turn it on).
        route_mode = r0._route_mode() if aux is not None else 0
        split = (aux is None and side is not None and Q >= 2 and mask is None and x.dtype == torch.float32
                 and os.environ.get("VQHIP_CHAIN_DECODE", "0") != "0")
        r = L.rvq_chain_forward(x, packed, embed, Q, row_mask=mask, route_mode=route_mode, row_chunks=K, stats=buf, stats_ws=stats_ws,
                                sq_parts=sq_parts, stats_stream=side, groups=G, decode_out=out if split else None)
        self.last_counts = r["counts"]                  # per stage: (open rows, pair rows) counters [chunks, G] (device, diagnostic)
        idx = r["idx"]                                  # [G, b, n, Q]
        if mask is not None:
            if side is not None:
                main.wait_stream(side)                  # the statistics passes have read idx
            idx.masked_fill_(~mask.reshape(1, *idx.shape[1:-1], 1).bool(), -1)
        if aux is not None:
            aux["embed"], aux["Q"] = embed, Q           # (torch.stack above copied: the folds below rewrite the module buffers)
            out = None
        elif not split:
            for g in range(G):
                L.decode_sum(idx[g], embed[g], out=out[..., g * D:(g + 1) * D])

        if update:
            if side is not None:
                if vq0._codebook.use_ddp:
                    with torch.cuda.stream(side):       # ONE all-reduce for every group and stage, beside the decode
                        dist.all_reduce(buf)
                main.wait_stream(side)
            elif vq0._codebook.use_ddp:
                dist.all_reduce(buf)
        losses = torch.zeros(G, Q, device=dev, dtype=torch.float32)
        if want_loss and sq_parts is not None:
            sums = L.reduce_partials_rows(sq_parts.view(G * Q, -1)).view(G, Q)
            denom = float(N * D) if mask is None else (mask.sum() * D).to(torch.float32)
            losses = sums / denom * vq0.commitment_weight
            if aux is not None:
                aux["loss_scale"] = vq0.commitment_weight / denom            # float, or a 0-dim device tensor under a mask
        if self.training and aux is None:
            losses = torch.zeros(G, Q, device=dev, requires_grad=True) + losses                            # as vqp.py:1282

        if update:
            cb0 = vq0._codebook
            if shared:
                for g, rv in enumerate(self.rvqs):      # the Q folds of a group's one codebook in stage order + its renormalisation
                    cs, ea, e = rv.layers[0]._codebook._views(0)
                    L.ema_fold_many(cs, ea, e, buf[g], decay=cb0.decay, eps=cb0.eps, cosine=False, do_update_ema=bool(r0.vq_is_ema_updating))
            else:
                L.ema_finalize_table(self._fold_table(Q), buf.view(G * Q, stride), C, D, decay=cb0.decay, eps=cb0.eps, cosine=False,
                                     do_update_ema=bool(cb0.ema_update and not cb0.manual_ema_update))
                if cb0.has_dead_code_replacement:       # vqp.py:641, per layer, after all the folds (as ResidualVQ._forward_fused)
                    cbs = [(g, q, rv.layers[q]._codebook) for g, rv in enumerate(self.rvqs) for q in range(Q)]
                    host = [i for i, (_, _, cb) in enumerate(cbs) if cb.expiry_reads_host()]
                    known = dict(zip(host, type(cb0).any_expired_many([cbs[i][2] for i in host]))) if len(host) > 1 else {}
                    for i, (g, q, cb) in enumerate(cbs):
                        stage_in = x[..., g * D:(g + 1) * D] if q == 0 else r["bufs"][q - 1, g]
                        cb.expire_codes_(stage_in.reshape(1, -1, D), seq_mask=None if mask is None else mask.reshape(1, -1).bool(),
                                         any_expired=known.get(i))
        return out, idx, losses

    @other_float_dtypes_as_fp32
    def forward(self, x, indices=None, return_all_codes=False, sample_codebook_temp=None, freeze_codebook=False, mask=None):
        if indices is not None and len(indices) > 0:
            raise NotImplementedError("GroupedResidualVQ.forward(indices=): every group's ResidualVQ.forward(indices=) raises in the reference "
                                      "(v1.31.0, rvq.py:493: ValueError) -- there is no behaviour to reproduce")
        assert x.shape[self.split_dim] == self.dim
        chunks = x.chunk(self.groups, dim=self.split_dim)
        seed = None
        if self.training:   # same RNG consumption as rvq.py:701; the value (host sync) only if dropout needs it
            seed = _draw_seed(x.device, need_value=any(r.quantize_dropout for r in self.rvqs))
        kw = dict(return_all_codes=return_all_codes, sample_codebook_temp=sample_codebook_temp, mask=mask,
                  freeze_codebook=freeze_codebook, rand_quantize_dropout_fixed_seed=seed)
        if self._batched_eligible(x, chunks, mask, freeze_codebook):
            if self.rvqs[0]._wants_input_grad(x):
                ret = _GrvqFusedFn.apply(x, self, mask, freeze_codebook)
            else:
                ret = self._forward_batched(x, mask, freeze_codebook)
            if return_all_codes:
                ret = (*ret, self.get_codes_from_indices(ret[1]))
            return ret
        if x.is_cuda and self.groups > 1 and self.concurrent_groups:
            # The groups share nothing (own codebooks, own feature chunk): each runs on its own HIP stream, forked from and
            # joined back into the caller's stream, so that one group's short exact-pass / statistics kernels run beside another
            # group's search instead of leaving the chip mostly idle.  Host code order (and with it RNG consumption) is unchanged.
            cur = torch.cuda.current_stream(x.device)
            side = self._side_streams(x.device)
            fork = torch.cuda.Event()
            fork.record(cur)
            outs = []
            _GROUPS_CONCURRENT[0] = True    # (the groups interleave with each other already: no row chunks inside them)
            try:
                for g, (r, c) in enumerate(zip(self.rvqs, chunks)):
                    if g == 0:
                        outs.append(r(c, **kw))
                        continue
                    side[g - 1].wait_event(fork)
                    with torch.cuda.stream(side[g - 1]):
                        outs.append(r(c, **kw))
            finally:
                _GROUPS_CONCURRENT[0] = False
            for g in range(1, self.groups):
                cur.wait_stream(side[g - 1])
                for t in outs[g]:
                    if torch.is_tensor(t):
                        t.record_stream(cur)          # allocated on the side stream, consumed (and freed) on the caller's
        else:
            outs = [r(c, **kw) for r, c in zip(self.rvqs, chunks)]
        quantized, all_idx, losses, *codes = zip(*outs)
        ret = (torch.cat(quantized, dim=self.split_dim), torch.stack(all_idx), torch.stack(losses))
        if codes:
            ret = (*ret, torch.stack(codes[0]))
        return ret
