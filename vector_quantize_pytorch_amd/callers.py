"""Callers of the hot path: wiring around `VectorQuantize`, nothing else.

* `RandomProjectionQuantizer` (reference: random_projection_quantizer.py:11-66): LayerNorm + frozen random projection in PyTorch,
  the multi-head cosine search on the HIP kernels; `forward(indices=)` streams the cross-entropy (vqhip_scores_lse per head).
* `HierarchicalVQ` (reference: hierarchical_vq.py:28-170): one shared `VectorQuantize` quantizing the residual image map pooled
  to s x s for growing s -- the small-N, launch-latency-bound caller of §8f-2 (k-means on a 1 x 1 map, dead-code replacement,
  a few hundred rows per call).  Pooling, bilinear up-sampling and the residual 3 x 3 conv are PyTorch ops; every search, EMA
  update and expiry runs through the same `VectorQuantize` as everywhere else.  Same constructor keywords, sub-module names
  (`vq`, `phi_shared` | `phi_levels.{i}.conv`: reference state_dicts load strictly) and outputs."""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn.functional as F
from torch import nn

from .vector_quantize import VectorQuantize


class RandomProjectionQuantizer(nn.Module):
    """BEST-RQ style frozen quantizer (arXiv:2202.01855): LayerNorm -> frozen random projection to
    `num_codebooks` heads -> cosine-similarity VQ with one frozen codebook per head, always in eval mode."""

    def __init__(self, *, dim, codebook_size, codebook_dim, num_codebooks=1, norm=True, **kwargs):
        super().__init__()
        self.num_codebooks = num_codebooks
        proj = torch.empty(num_codebooks, dim, codebook_dim)
        nn.init.xavier_normal_(proj)
        self.register_buffer('rand_projs', proj)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False) if norm else nn.Identity()
        self.vq = VectorQuantize(dim=codebook_dim * num_codebooks, heads=num_codebooks, codebook_size=codebook_size,
                                 use_cosine_sim=True, separate_codebook_per_head=True, **kwargs)

    def forward(self, x, indices=None):
        """-> indices [b, n, h]; with `indices` given: the cross-entropy of the score rows to them (random_projection_quantizer.py:46-66)"""
        x = self.norm(x)
        x = torch.einsum('bnd,hde->bnhe', x, self.rand_projs).flatten(2)
        self.vq.eval()
        out = self.vq(x, indices=indices)
        return out[1]


class _ResidualMix(nn.Module):
    """(1 - ratio) t + ratio conv3x3(t)  (hierarchical_vq.py:16-25); identity for ratio ~ 0"""

    def __init__(self, dim, ratio):
        super().__init__()
        self.resi_ratio = abs(float(ratio))
        self.conv = nn.Conv2d(dim, dim, kernel_size=3, padding=1)

    def forward(self, t):
        if self.resi_ratio <= 1e-8:
            return t
        return (1. - self.resi_ratio) * t + self.resi_ratio * self.conv(t)


class HierarchicalVQ(nn.Module):
    def __init__(self, *, dim: int, codebook_size: int, scales: Sequence[int], decay: float = 0.99, commitment_weight: float = 1.,
                 rotation_trick: bool = False, kmeans_init: bool = True, kmeans_iters: int = 10, threshold_ema_dead_code: int = 2,
                 stochastic_sample_codes: bool = False, sample_codebook_temp: float = 0.1, orthogonal_reg_weight: float = 0.,
                 orthogonal_reg_max_codes: int = 128, orthogonal_reg_active_codes_only: bool = False, quant_resi: float = 0.5,
                 share_quant_resi: int = 1, accept_image_fmap: bool = False):
        super().__init__()
        assert accept_image_fmap, 'HierarchicalVQ currently expects accept_image_fmap = True'
        scales = [int(s) for s in scales]
        assert len(scales) > 0 and scales == sorted(scales) and all(s > 0 for s in scales)
        self.dim, self.scales, self.accept_image_fmap = dim, tuple(scales), True
        self.vq = VectorQuantize(
            dim=dim, codebook_size=codebook_size, decay=decay, commitment_weight=commitment_weight, rotation_trick=rotation_trick,
            kmeans_init=kmeans_init, kmeans_iters=kmeans_iters, threshold_ema_dead_code=threshold_ema_dead_code,
            stochastic_sample_codes=stochastic_sample_codes, sample_codebook_temp=sample_codebook_temp,
            orthogonal_reg_weight=orthogonal_reg_weight, orthogonal_reg_max_codes=orthogonal_reg_max_codes,
            orthogonal_reg_active_codes_only=orthogonal_reg_active_codes_only, accept_image_fmap=True)
        # one mixer for all scales, one per scale (share_quant_resi <= 0), or `share_quant_resi` of them spread evenly over the scales
        if share_quant_resi == 1:
            self.phi_shared, self.phi_levels = _ResidualMix(dim, quant_resi), None
        else:
            n = len(scales) if share_quant_resi <= 0 else min(len(scales), int(share_quant_resi))
            self.phi_shared, self.phi_levels = None, nn.ModuleList([_ResidualMix(dim, quant_resi) for _ in range(n)])

    def _mixer(self, i):
        if self.phi_shared is not None:
            return self.phi_shared
        m, n = len(self.phi_levels), len(self.scales)
        if m == n:
            return self.phi_levels[i]
        if n == 1:
            return self.phi_levels[0]
        return self.phi_levels[max(0, min(m - 1, round(i / float(n - 1) * (m - 1))))]

    def _to_full(self, q, size, i):
        if tuple(q.shape[-2:]) != tuple(size):
            q = F.interpolate(q, size=size, mode='bilinear', align_corners=False)
        return self._mixer(i)(q)

    def forward(self, x, indices=None, sample_codebook_temp=None, **kwargs):
        assert indices is None, 'reconstruction-from-indices path not implemented in forward'
        assert x.ndim == 4 and x.shape[1] == self.dim, 'expected image fmap of shape (batch, channels, height, width)'
        size = tuple(x.shape[-2:])
        kw = {} if sample_codebook_temp is None else dict(sample_codebook_temp=sample_codebook_temp)
        rest, total, all_idx, all_loss = x, torch.zeros_like(x), [], []
        for i, s in enumerate(self.scales):
            q, idx, loss = self.vq(F.adaptive_avg_pool2d(rest, (s, s)), **kw)
            q = self._to_full(q, size, i)
            total, rest = total + q, rest - q
            all_idx.append(idx)
            all_loss.append(loss)
        return total, tuple(all_idx), torch.stack(all_loss).mean()

    def get_output_from_indices(self, indices):
        assert isinstance(indices, (tuple, list)) and len(indices) == len(self.scales) and indices[0].ndim == 3
        size = (self.scales[-1], self.scales[-1])
        out = None
        for i, ind in enumerate(indices):
            q = self._to_full(self.vq.get_output_from_indices(ind), size, i)
            out = q if out is None else out + q
        return out
