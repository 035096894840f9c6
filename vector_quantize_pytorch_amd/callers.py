"""Thin callers of the hot path: `RandomProjectionQuantizer`
(reference: random_projection_quantizer.py:11-66) and `HierarchicalVQ` (hierarchical_vq.py:28-170).
Both only arrange tensors around `VectorQuantize`; the projection / pooling / interpolation / conv stay PyTorch ops on
the GPU, the nearest-code search runs on the HIP kernels."""
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
        if indices is not None:
            raise NotImplementedError("the cross-entropy-to-given-indices path reads the full distance matrix; not on the "
                                      "MI355X hot path (SURVEY.md §8f)")
        x = self.norm(x)
        x = torch.einsum('bnd,hde->bnhe', x, self.rand_projs).flatten(2)
        self.vq.eval()
        return self.vq(x)[1]


class _Phi2D(nn.Module):
    """residual 3x3 conv mix applied to an up-sampled quantized map (hierarchical_vq.py:15-25)"""

    def __init__(self, dim: int, resi_ratio: float):
        super().__init__()
        self.resi_ratio = float(abs(resi_ratio))
        self.conv = nn.Conv2d(dim, dim, 3, padding=1)

    def forward(self, x):
        if self.resi_ratio <= 1e-8:
            return x
        return (1. - self.resi_ratio) * x + self.resi_ratio * self.conv(x)


class HierarchicalVQ(nn.Module):
    """multi-scale (VAR-style) residual VQ of an image feature map with ONE shared codebook: for every scale the
    residual is average-pooled to (s, s), quantized, bilinearly up-sampled, passed through phi and subtracted."""

    def __init__(self, *, dim: int, codebook_size: int, scales: Sequence[int], decay: float = 0.99, commitment_weight: float = 1.,
                 rotation_trick: bool = False, kmeans_init: bool = True, kmeans_iters: int = 10, threshold_ema_dead_code: int = 2,
                 stochastic_sample_codes: bool = False, sample_codebook_temp: float = 0.1, orthogonal_reg_weight: float = 0.,
                 orthogonal_reg_max_codes: int = 128, orthogonal_reg_active_codes_only: bool = False, quant_resi: float = 0.5,
                 share_quant_resi: int = 1, accept_image_fmap: bool = False):
        super().__init__()
        assert accept_image_fmap, 'HierarchicalVQ currently expects accept_image_fmap = True'
        scales = [int(s) for s in scales]
        assert len(scales) > 0 and scales == sorted(scales) and all(s > 0 for s in scales)
        self.dim = dim
        self.scales = tuple(scales)
        self.accept_image_fmap = True
        self.vq = VectorQuantize(dim=dim, codebook_size=codebook_size, decay=decay, commitment_weight=commitment_weight,
                                 rotation_trick=rotation_trick, kmeans_init=kmeans_init, kmeans_iters=kmeans_iters,
                                 threshold_ema_dead_code=threshold_ema_dead_code, stochastic_sample_codes=stochastic_sample_codes,
                                 sample_codebook_temp=sample_codebook_temp, orthogonal_reg_weight=orthogonal_reg_weight,
                                 orthogonal_reg_max_codes=orthogonal_reg_max_codes,
                                 orthogonal_reg_active_codes_only=orthogonal_reg_active_codes_only, accept_image_fmap=True)
        if share_quant_resi == 1:
            self.phi_shared, self.phi_levels = _Phi2D(dim, quant_resi), None
        else:
            n = len(self.scales) if share_quant_resi <= 0 else min(len(self.scales), int(share_quant_resi))
            self.phi_shared, self.phi_levels = None, nn.ModuleList([_Phi2D(dim, quant_resi) for _ in range(n)])

    def _phi(self, scale_index: int):
        if self.phi_shared is not None:
            return self.phi_shared
        if len(self.phi_levels) == len(self.scales):
            return self.phi_levels[scale_index]
        if len(self.scales) == 1:
            return self.phi_levels[0]
        pos = scale_index / float(len(self.scales) - 1)
        k = round(pos * (len(self.phi_levels) - 1))
        return self.phi_levels[max(0, min(len(self.phi_levels) - 1, k))]

    def _to_full(self, q, full_hw, scale_index: int):
        if q.shape[-2:] != full_hw:
            q = F.interpolate(q, size=full_hw, mode='bilinear', align_corners=False)
        return self._phi(scale_index)(q)

    def forward(self, x, indices=None, sample_codebook_temp=None, **kwargs):
        assert indices is None, 'reconstruction-from-indices path not implemented in forward'
        assert x.ndim == 4 and x.shape[1] == self.dim, 'expected image fmap of shape (batch, channels, height, width)'
        hw = tuple(x.shape[-2:])
        residual, recon = x, torch.zeros_like(x)
        all_idx, all_loss = [], []
        for si, s in enumerate(self.scales):
            down = F.adaptive_avg_pool2d(residual, output_size=(s, s))
            q, idx, loss = self.vq(down) if sample_codebook_temp is None else self.vq(down, sample_codebook_temp=sample_codebook_temp)
            q = self._to_full(q, hw, si)
            recon = recon + q
            residual = residual - q
            all_idx.append(idx)
            all_loss.append(loss)
        return recon, tuple(all_idx), torch.stack(all_loss).mean()

    def get_output_from_indices(self, indices):
        assert isinstance(indices, (tuple, list)) and len(indices) == len(self.scales)
        assert indices[0].ndim == 3
        full_hw = (self.scales[-1], self.scales[-1])
        out = None
        for si, idx in enumerate(indices):
            q = self._to_full(self.vq.get_output_from_indices(idx), full_hw, si)
            out = q if out is None else out + q
        return out
