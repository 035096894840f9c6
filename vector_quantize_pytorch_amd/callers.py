"""Thin caller of the hot path: `RandomProjectionQuantizer` (reference: random_projection_quantizer.py:11-66)
only arranges tensors around `VectorQuantize`; the projection stays a PyTorch op on the GPU, the nearest-code search runs on the
HIP kernels.

`HierarchicalVQ` (hierarchical_vq.py) is NOT provided: it is pooling / interpolation / 3x3-conv glue around an unchanged
`VectorQuantize` (out of scope), i.e. nothing in it belongs to the accelerated path.  Its call pattern -- the one
`VectorQuantize` quantizing image maps of growing size inside a residual loop, k-means on a 1 x 1 map, dead-code replacement --
is still pinned by the golden fixtures `hvq` / `hvq_nokmeans` through a test-side harness (tests/golden_util.py)."""
from __future__ import annotations

import torch
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
