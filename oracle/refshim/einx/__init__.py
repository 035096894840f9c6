"""Stand-in for the `einx` package, TEST INFRASTRUCTURE ONLY.

The live reference (/root/reference, lucidrains/vector-quantize-pytorch v1.31.0) hard-imports
`einx` (vector_quantize_pytorch.py:16, residual_vq.py:19-20), which is not installed in this
image and cannot be installed (no network).  This module implements exactly the patterns the
VQ / RVQ path uses so that the reference can be imported *in the build container* to
(a) validate the restatement oracle and (b) generate the golden fixtures under tests/golden/.

It is never imported by the product package and is not needed on the GPU box.

Patterns covered (reference call sites):
  where('b n, b n ... d, b n d -> b n ... d')   vector_quantize_pytorch.py:1384
  where('b n, b n ..., -> b n ...')             vector_quantize_pytorch.py:1391
  where('..., ... k, -> ... k')                 vector_quantize_pytorch.py:1315
  where('..., ... l,')                          residual_vq.py:579
  add('... j, ... j k -> ... (j k)')            residual_vq.py:515
  get_at('q [c] d, b n q -> q b n d')           residual_vq.py:346
  get_at('b n [c] d, b n -> b n d')             residual_vq.py:360
  get_at('[c] d, b n -> b n d') / 'b ... -> b ... d'   residual_vq.py:362, sim_vq.py:92,117
"""
import torch


def _squash(pattern):
    return pattern.replace(' ', '')


def where(pattern, cond, a, b):
    p = _squash(pattern)
    if not torch.is_tensor(a):
        a = torch.as_tensor(a)
    while cond.ndim < a.ndim:
        cond = cond.unsqueeze(-1)
    if torch.is_tensor(b):
        if p == 'bn,bn...d,bnd->bn...d':
            # b is [b, n, d]; a may carry extra axes between n and d
            while b.ndim < a.ndim:
                b = b.unsqueeze(-2)
        b = b.to(a.dtype)
        return torch.where(cond, a, b)
    return torch.where(cond, a, torch.as_tensor(b, dtype=a.dtype, device=a.device))


def add(pattern, a, b):
    assert _squash(pattern) == '...j,...jk->...(jk)', pattern
    return (a.unsqueeze(-1) + b).flatten(-2)


def get_at(pattern, table, idx):
    p = _squash(pattern)
    if p == 'q[c]d,bnq->qbnd':
        if not torch.is_tensor(table):
            table = torch.stack(tuple(table))
        return torch.stack([table[i][idx[..., i]] for i in range(idx.shape[-1])])
    if p == 'bn[c]d,bn->bnd':
        d = table.shape[-1]
        g = idx[..., None, None].expand(*idx.shape, 1, d)
        return table.gather(-2, g).squeeze(-2)
    if p in ('[c]d,bn->bnd', '[c]d,b...->b...d'):
        return table[idx]
    raise NotImplementedError(pattern)
