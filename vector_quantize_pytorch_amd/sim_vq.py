"""`SimVQ` / `ResidualSimVQ` for MI355X (reference: sim_vq.py:37-138, residual_sim_vq.py:49-220).

SimVQ keeps a frozen random codebook and learns a map on top of it; the codebook that is searched is
`code_transform(frozen_codebook)`.  The search itself is the same nearest-code problem as VectorQuantize's, so it
runs on `vqhip_assign` (fp32 MFMA); everything that carries gradients (the differentiable gather into the learned
map, the two-sided commit loss) is ordinary autograd, and the rotation trick / straight-through value and its
gradient to the input are `vq_route_kernel` (the reference detaches the target inside those formulas, sim_vq.py:126-131).

The reference uses `torch.cdist(x, codebook).argmin(-1)` (sim_vq.py:111-113).  torch.cdist folds the norms into one
sgemm (`[-2x, |x|^2, 1] . [c, 1, |c|^2]`), whose summation order is MKL's; like the dot product of §2 in DESIGN.md it
cannot be reproduced bit for bit, only the real nearest code can -- indices agree except on rows whose two best codes
are closer than fp32 round-off (none in the golden fixtures).
"""
from __future__ import annotations

import os
import random
from typing import Callable, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from .residual_vq import _draw_seed, _round_up
from .vector_quantize import _RouteFn, _rows_of


class _SimQuantizeFn(torch.autograd.Function):
    """SimVQ's search, gather, both commitment terms and the routed output (sim_vq.py:107-131) as the hot path runs them: the search
    returns indices and the sum of squared errors, the routed value gathers the code rows by index, and backward is two kernels --
    the routing kernel for x (rotation trick / straight-through Jacobian of the upstream gradient + the x-side commitment term
    `w * 2 (x - q) / numel`) and one statistics pass for the codes (`2 (count_c q_c - sum of the rows quantized to c) / numel`, the
    gradient of mse(x.detach(), codes[idx]); the routed output detaches the codes, sim_vq.py:126-131).  Replaces F.embedding, two
    F.mse_loss graphs over N x D tensors and embedding's backward.  Third output: mse(x, codes[idx]); the module multiplies it by
    (1 + w) -- both commitment terms have that value -- and this node splits the incoming gradient 1 : w between the two routes."""

    @staticmethod
    def forward(ctx, rows, implicit, mode, w):
        searched = implicit.detach().float().contiguous()
        numel = max(rows.numel(), 1)
        r = L.assign(rows, L.pack_codebook(searched), searched, want_q=False, want_sqerr=True)          # sim_vq.py:111-113
        idx = r["idx"].contiguous()
        mse = L.reduce_partials(r["sqerr_partials"], r["nblk"], 1.0 / numel)
        codes = searched.to(rows.dtype)
        out = L.route_fwd_gather(rows, codes, idx, mode)
        ctx.mode, ctx.w, ctx.scale, ctx.idt = mode, float(w), 1.0 / numel, implicit.dtype
        ctx.save_for_backward(rows, codes, idx, searched)
        ctx.mark_non_differentiable(idx)
        return out, idx, mse

    @staticmethod
    def backward(ctx, g_out, g_idx, g_loss):
        rows, codes, idx, searched = ctx.saved_tensors
        g_codes = None
        coef_x = None
        if g_loss is not None:
            g32 = g_loss.to(torch.float32)
            share = 1.0 / (1.0 + ctx.w)
            if ctx.needs_input_grad[1]:
                count, esum = L.ema_accumulate(rows, idx.reshape(-1), searched.shape[0])
                g_codes = ((count[:, None] * searched - esum) * (g32 * (2.0 * ctx.scale * share))).to(ctx.idt)
            coef_x = g32 * (ctx.scale * ctx.w * share)
        if not ctx.needs_input_grad[0] or (g_out is None and coef_x is None):
            return None, g_codes, None, None
        gx = L.route_bwd_gather(rows, codes, idx, None if g_out is None else L.rows_contiguous(g_out), coef_x, None,
                                ctx.mode if g_out is not None else 0)
        return gx, g_codes, None, None


class SimVQ(nn.Module):
    def __init__(
        self,
        dim,
        codebook_size,
        codebook_transform: Optional[nn.Module] = None,
        init_fn: Callable = lambda t: t,
        channel_first=False,
        rotation_trick=True,
        input_to_quantize_commit_loss_weight=0.25,
        commitment_weight=1.,
        frozen_codebook_dim=None,
    ):
        super().__init__()
        self.codebook_size = codebook_size
        self.channel_first = channel_first
        frozen_codebook_dim = dim if frozen_codebook_dim is None else frozen_codebook_dim
        codebook = init_fn(torch.randn(codebook_size, frozen_codebook_dim) * (frozen_codebook_dim ** -0.5))   # sim_vq.py:55-56
        self.code_transform = nn.Linear(frozen_codebook_dim, dim, bias=False) if codebook_transform is None else codebook_transform
        self.register_buffer('frozen_codebook', codebook)
        self.rotation_trick = rotation_trick
        self.input_to_quantize_commit_loss_weight = input_to_quantize_commit_loss_weight
        self.commitment_weight = commitment_weight

    @property
    def codebook(self):
        return self.code_transform(self.frozen_codebook)

    def indices_to_codes(self, indices):
        quantized = self.code_transform(self.frozen_codebook[indices])
        return quantized.movedim(-1, 1) if self.channel_first else quantized

    def forward(self, x):
        L._need_gpu(x)
        if self.channel_first:
            x = x.movedim(1, -1)
        lead = x.shape[:-1]
        rows = x.reshape(lead[0], -1, x.shape[-1])                               # 'b * d'

        implicit = self.codebook                                                # [C, D], carries grad to the learned map
        if (rows.is_cuda and rows.dtype in (torch.float32, torch.bfloat16) and rows.shape[-1] <= 512 and rows.numel() > 0
                and os.environ.get("VQHIP_SIM_FAST", "1") != "0"):
            w = self.input_to_quantize_commit_loss_weight
            mode = L.ROTATION if self.rotation_trick else L.STRAIGHT_THROUGH
            quantized, idx, mse = _SimQuantizeFn.apply(rows if rows.is_contiguous() else _rows_of(rows), implicit, mode, w)
            quantized = quantized.reshape(*lead, -1)
            if self.channel_first:
                quantized = quantized.movedim(-1, 1)
            return quantized, idx.reshape(lead), mse * ((1.0 + w) * self.commitment_weight)
        searched = implicit.detach().float().contiguous()
        with torch.no_grad():
            idx = L.assign(rows.detach(), L.pack_codebook(searched), searched, want_q=False)["idx"]   # sim_vq.py:111-113
        quantized = F.embedding(idx, implicit).to(rows.dtype)                  # sim_vq.py:117

        commit_loss = (F.mse_loss(rows.detach(), quantized) +
                       F.mse_loss(rows, quantized.detach()) * self.input_to_quantize_commit_loss_weight)   # sim_vq.py:121-124
        mode = L.ROTATION if self.rotation_trick else L.STRAIGHT_THROUGH
        quantized = _RouteFn.apply(rows, quantized.detach(), mode)             # sim_vq.py:126-131

        quantized = quantized.reshape(*lead, -1)
        idx = idx.reshape(lead)
        if self.channel_first:
            quantized = quantized.movedim(-1, 1)
        return quantized, idx, commit_loss * self.commitment_weight


class ResidualSimVQ(nn.Module):
    def __init__(
        self,
        *,
        dim,
        num_quantizers,
        codebook_size,
        heads=1,
        quantize_dropout=False,
        quantize_dropout_cutoff_index=0,
        quantize_dropout_multiple_of=1,
        channel_first=False,
        rotation_trick=True,
        **sim_vq_kwargs,
    ):
        super().__init__()
        assert heads == 1, 'residual vq is not compatible with multi-headed codes'
        self.channel_first = channel_first
        self.num_quantizers = num_quantizers
        self.layers = nn.ModuleList([SimVQ(dim=dim, codebook_size=codebook_size, rotation_trick=rotation_trick,
                                           channel_first=channel_first, **sim_vq_kwargs) for _ in range(num_quantizers)])
        self.quantize_dropout = quantize_dropout and num_quantizers > 1
        assert quantize_dropout_cutoff_index >= 0
        self.quantize_dropout_cutoff_index = quantize_dropout_cutoff_index
        self.quantize_dropout_multiple_of = quantize_dropout_multiple_of

    @property
    def codebook_size(self):
        return self.layers[0].codebook_size

    @property
    def codebooks(self):
        return torch.stack([layer.codebook for layer in self.layers])

    def get_codes_from_indices(self, indices):
        """[b, ..., q] -> [q, b, ..., d] ([q, b, d, ...] when channel_first); -1 decodes to zeros (residual_sim_vq.py:95-132)."""
        qdim = indices.shape[-1]
        if qdim < self.num_quantizers:
            assert self.quantize_dropout > 0., 'quantize dropout must be greater than 0 if you wish to reconstruct from a signal with less fine quantizations'
            indices = F.pad(indices, (0, self.num_quantizers - qdim), value=-1)
        cbs = self.codebooks.detach().float().contiguous()
        codes = torch.stack([L.decode_sum(indices[..., q:q + 1].contiguous(), cbs[q].contiguous()) for q in range(self.num_quantizers)])
        return codes.movedim(-1, 2) if self.channel_first else codes

    def get_output_from_indices(self, indices):
        qdim = indices.shape[-1]
        if qdim < self.num_quantizers:
            assert self.quantize_dropout > 0., 'quantize dropout must be greater than 0 if you wish to reconstruct from a signal with less fine quantizations'
            indices = F.pad(indices, (0, self.num_quantizers - qdim), value=-1)
        out = L.decode_sum(indices.contiguous(), self.codebooks.detach().float().contiguous())   # fused gather + sum over q
        return out.movedim(-1, 1) if self.channel_first else out

    def forward(self, x, return_all_codes=False, rand_quantize_dropout_fixed_seed=None):
        Q = self.num_quantizers
        drop_at = None
        if self.training and self.quantize_dropout:                              # residual_sim_vq.py:158-172
            seed = rand_quantize_dropout_fixed_seed
            if seed is None:
                seed = _draw_seed(x.device, need_value=True)
            drop_at = random.Random(seed).randrange(self.quantize_dropout_cutoff_index, Q)
            if self.quantize_dropout_multiple_of != 1:
                drop_at = _round_up(drop_at + 1, self.quantize_dropout_multiple_of) - 1
        idx_shape = (x.shape[0], *x.shape[2:]) if self.channel_first else tuple(x.shape[:-1])

        quantized_out = 0.
        residual = x
        all_idx, all_loss = [], []
        for qi, layer in enumerate(self.layers):                                 # residual_sim_vq.py:183-205
            if drop_at is not None and qi > drop_at:
                all_idx.append(torch.full(idx_shape, -1, device=x.device, dtype=torch.long))
                all_loss.append(torch.zeros((), device=x.device, dtype=x.dtype))
                continue
            quantized, ind, loss = layer(residual)
            residual = residual - quantized.detach()
            quantized_out = quantized_out + quantized
            all_idx.append(ind)
            all_loss.append(loss)
        ret = (quantized_out, torch.stack(all_idx, -1), torch.stack(all_loss, -1))
        if return_all_codes:
            ret = (*ret, self.get_codes_from_indices(ret[1]))
        return ret
