"""
oracle/vq_oracle.py -- TEST INFRASTRUCTURE ONLY (the parity oracle; never shipped, never measured
as the product).  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg import it.

A CPU restatement of the VectorQuantize / ResidualVQ forward path of
lucidrains/vector-quantize-pytorch v1.31.0.  Citations: "vqp.py" =
/root/reference/vector_quantize_pytorch/vector_quantize_pytorch.py, "rvq.py" = .../residual_vq.py.

Two assignment back-ends:
  mode="aten"  : distances with the same ATen/MKL ops the reference issues (bit-identical to the
                 live reference on the same host; with quantize_mode="onehot" this is also what
                 `cpu_baseline` times, because it is the reference's real CPU cost: 3 N*C*D
                 contractions -- cdist, the one-hot gather einsum of a training forward, the EMA
                 einsum -- + the N*C temporaries; the default quantize_mode="gather" runs 2 of them).
  mode="chain" : the deterministic C restatement (oracle/vq_oracle.c) whose x.c^T is one fp32 FMA
                 chain in ascending k -- host-independent, and bit-identical to the HIP kernel.

Parity pinning: tests/golden/*.npz were produced by the LIVE reference (tests/golden/make_golden.py,
run in the build container where /root/reference is mounted); tests/test_oracle.py checks both
back-ends against them.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvqoracle.so")
_lib = None


def lib():
    """ctypes handle of the C oracle (oracle/Makefile builds it)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, p = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        L.vqo_row_sumsq.argtypes = [p, i64, i32, i64, p]
        L.vqo_l2norm_rows.argtypes = [p, i64, i32, i64, p, i64]
        L.vqo_assign.argtypes = [p, i64, i32, i64, p, i32, i32, p, p]
        L.vqo_scores.argtypes = [p, i64, i32, i64, p, i32, i32, p]
        L.vqo_ema_stats.argtypes = [p, i64, i32, i64, p, i32, p, p]
        L.vqo_version.restype = ctypes.c_char_p
        for f in (L.vqo_row_sumsq, L.vqo_l2norm_rows, L.vqo_assign, L.vqo_scores, L.vqo_ema_stats):
            f.restype = None
        _lib = L
    return _lib


def _f32c(t):
    t = t.detach()
    assert t.device.type == "cpu"
    return t.to(torch.float32).contiguous()


# ----------------------------------------------------------------------------------------------
# C back-end wrappers
# ----------------------------------------------------------------------------------------------
def c_row_sumsq(x2d: torch.Tensor) -> torch.Tensor:
    x2d = _f32c(x2d)
    out = torch.empty(x2d.shape[0], dtype=torch.float32)
    lib().vqo_row_sumsq(x2d.data_ptr(), x2d.shape[0], x2d.shape[1], x2d.stride(0), out.data_ptr())
    return out


def c_l2norm(x2d: torch.Tensor) -> torch.Tensor:
    x2d = _f32c(x2d)
    out = torch.empty_like(x2d)
    lib().vqo_l2norm_rows(x2d.data_ptr(), x2d.shape[0], x2d.shape[1], x2d.stride(0), out.data_ptr(), out.stride(0))
    return out


def c_assign(x2d: torch.Tensor, embed2d: torch.Tensor, cosine: bool = False):
    """-> (idx int64 [N], best fp32 [N])  (best = distance, or similarity when cosine)."""
    x2d, embed2d = _f32c(x2d), _f32c(embed2d)
    N, D = x2d.shape
    idx = torch.empty(N, dtype=torch.int64)
    best = torch.empty(N, dtype=torch.float32)
    lib().vqo_assign(x2d.data_ptr(), N, D, x2d.stride(0), embed2d.data_ptr(), embed2d.shape[0],
                     int(cosine), idx.data_ptr(), best.data_ptr())
    return idx, best


def c_scores(x2d: torch.Tensor, embed2d: torch.Tensor, cosine: bool = False) -> torch.Tensor:
    x2d, embed2d = _f32c(x2d), _f32c(embed2d)
    N, D = x2d.shape
    out = torch.empty(N, embed2d.shape[0], dtype=torch.float32)
    lib().vqo_scores(x2d.data_ptr(), N, D, x2d.stride(0), embed2d.data_ptr(), embed2d.shape[0], int(cosine), out.data_ptr())
    return out


def c_ema_stats(x2d: torch.Tensor, idx: torch.Tensor, C: int):
    x2d = _f32c(x2d)
    idx = idx.to(torch.int64).contiguous()
    N, D = x2d.shape
    count = torch.empty(C, dtype=torch.float32)
    esum = torch.empty(C, D, dtype=torch.float32)
    lib().vqo_ema_stats(x2d.data_ptr(), N, D, x2d.stride(0), idx.data_ptr(), C, count.data_ptr(), esum.data_ptr())
    return count, esum


# ----------------------------------------------------------------------------------------------
# torch restatement of the helpers (vqp.py L0 layer)
# ----------------------------------------------------------------------------------------------
def l2norm(t, eps=1e-6):                       # vqp.py:37-38
    return F.normalize(t, p=2, dim=-1, eps=eps)


def neg_cdist(x, y, eps=1e-8):                 # vqp.py:58-62 + negation at :743
    """x [H,N,D], y [H,C,D] -> -dist [H,N,C]; association exactly (x2 + y2) + (-2 xy)."""
    x2 = (x ** 2).sum(-1)
    y2 = (y ** 2).sum(-1)
    xy = torch.einsum('hid,hjd->hij', x, y) * -2
    return -((x2[:, :, None] + y2[:, None, :] + xy).clamp(min=eps).sqrt())


def scores(flat, embed, cosine):               # vqp.py:740-743
    if cosine:
        return torch.einsum('hnd,hcd->hnc', flat, embed)
    return neg_cdist(flat, embed)


def lerp_inplace(old, new, decay, weight=None):    # vqp.py:76-97 (ema_inplace, no .grad folding)
    w = 1.0 if weight is None else weight
    if torch.is_tensor(w):
        if w.ndim == 1:
            w = w[None]
        while w.ndim < old.ndim:
            w = w[..., None]
    old.lerp_(new.to(old), (1.0 - decay) * w)


def laplace(cs, C, eps):                       # vqp.py:152-154
    return (cs + eps) / (cs.sum(-1, keepdim=True) + C * eps)


def rotate_to(src, tgt):                       # vqp.py:287-318 (rotation trick, arXiv:2410.06424)
    shp = src.shape
    e = src.reshape(-1, shp[-1])
    q = tgt.reshape(-1, shp[-1])
    ne = e.norm(dim=-1, keepdim=True)
    nq = q.norm(dim=-1, keepdim=True)
    u = (e / ne.clamp(min=1e-6)).detach()
    qh = (q / nq.clamp(min=1e-6)).detach()
    w = l2norm(u + qh).detach()
    e1 = e[:, None, :]
    out = e1 - 2 * (e1 @ w[:, :, None] @ w[:, None, :]) + 2 * (e1 @ u[:, :, None] @ qh[:, None, :])
    out = out[:, 0, :] * (nq / ne.clamp(min=1e-6)).detach()
    return out.reshape(shp)


def rotate_to_bf16_ops(src, tgt):
    """rotate_to (vqp.py:287-318) on bf16 rows, written as fp32 arithmetic with an explicit round-to-bf16 after every TENSOR OP of the
    reference -- the sequence the routing kernels apply (csrc/vq_route_math.h, BF16 = true).  src, tgt: fp32 tensors holding bf16
    values.  A reduction (norm, bmm) accumulates in fp32 and rounds once, like ATen's; `2 * t` is exact.  Pinned against torch's own bf16
    tensor ops -- which is what the reference executes -- by tests/test_oracle.py (bit-identical but for rows where the order of a
    row's fp32 sum moves a rounding)."""
    rb = lambda t: t.to(torch.bfloat16).float()
    e, q = src.reshape(-1, src.shape[-1]), tgt.reshape(-1, src.shape[-1])
    ne = rb(e.pow(2).sum(-1, keepdim=True).sqrt())            # src.norm(dim = -1, keepdim = True)              :292
    nq = rb(q.pow(2).sum(-1, keepdim=True).sqrt())            # tgt.norm(...)                                    :293
    de, dq = rb(ne.clamp(min=1e-6)), rb(nq.clamp(min=1e-6))   # safe_div's den.clamp(min = eps)                  :40-41
    u, qh = rb(e / de), rb(q / dq)                            # safe_div(src, norm_src), safe_div(tgt, norm_tgt) :296-297
    s = rb(u + qh)                                            # u + q                                            :305
    dn = rb(rb(s.pow(2).sum(-1, keepdim=True).sqrt()).clamp(min=1e-6))        # F.normalize: norm, clamp_min(eps)
    w = rb(s / dn)
    a1 = rb((e * w).sum(-1, keepdim=True))                    # e @ w^T  (bmm, [b, 1, d] @ [b, d, 1])            :309
    a2 = rb((e * u).sum(-1, keepdim=True))                    # e @ u^T                                          :310
    out = rb(rb(e - 2 * rb(a1 * w)) + 2 * rb(a2 * qh))        # e - 2 (...) + 2 (...), left to right             :307-311
    return rb(out * rb(nq / de)).reshape(src.shape)           # rotated_tgt * safe_div(norm_tgt, norm_src)       :316


def sample_rows(samples, num):                 # vqp.py:156-163 (consumes torch's global RNG)
    n = samples.shape[0]
    if n >= num:
        ind = torch.randperm(n, device=samples.device)[:num]
    else:
        ind = torch.randint(0, n, (num,), device=samples.device)
    return samples[ind]


def batched_sample_rows(samples, num):         # vqp.py:165-166
    return torch.stack([sample_rows(s, num) for s in samples.unbind(0)], 0)


def kmeans(samples, C, iters=10, cosine=False, sample_fn=batched_sample_rows, assign_mode="aten"):
    """vqp.py:238-278.  samples [H,N,D] -> (means [H,C,D], bins [H,C] int64)."""
    H, N, D = samples.shape
    means = sample_fn(samples, C)
    bins = None
    for _ in range(iters):
        buckets = _assign(samples, means, cosine, assign_mode)          # :251-256
        bins = torch.zeros(H, C, dtype=torch.int64)
        bins.scatter_add_(-1, buckets, torch.ones_like(buckets))        # :231-236
        zero = bins == 0
        denom = bins.masked_fill(zero, 1)
        new = torch.zeros(H, C, D, dtype=samples.dtype)
        new.scatter_add_(1, buckets[..., None].expand(-1, -1, D), samples)   # :265
        new = new / denom[..., None]
        if cosine:
            new = l2norm(new)
        means = torch.where(zero[..., None], means, new)                 # :272-276
    return means, bins


def _assign(flat, embed, cosine, mode):
    """argmax of the score row, first occurrence on ties (vqp.py:140). flat [H,N,D] -> [H,N]."""
    if mode == "aten":
        return scores(flat, embed, cosine).argmax(-1)
    if mode == "chain":
        return torch.stack([c_assign(f, e, cosine)[0] for f, e in zip(flat.unbind(0), embed.unbind(0))], 0)
    raise ValueError(mode)


# ----------------------------------------------------------------------------------------------
# Codebook + VectorQuantize restatement (only the options on the north-star hot path)
# ----------------------------------------------------------------------------------------------
@dataclass
class VQConfig:
    dim: int
    codebook_size: int
    use_cosine_sim: bool = False
    decay: float = 0.8
    eps: float = 1e-5
    threshold_ema_dead_code: int = 0        # VectorQuantize default (vqp.py:818)
    reset_cluster_size: Optional[float] = None
    kmeans_init: bool = False
    kmeans_iters: int = 10
    commitment_weight: float = 1.0
    rotation_trick: bool = True             # default when dim > 1 (vqp.py:856)
    ema_update: bool = True
    manual_ema_update: bool = False
    num_codebooks: int = 1


@dataclass
class VQState:
    """Mirror of the reference's buffers (vqp.py:415-423) -- same names as the state_dict keys."""
    embed: torch.Tensor            # [H, C, D]
    embed_avg: torch.Tensor        # [H, C, D]
    cluster_size: torch.Tensor     # [H, C]
    initted: bool = True

    @staticmethod
    def from_state_dict(sd, prefix="_codebook."):
        return VQState(embed=sd[prefix + "embed"].clone().float(),
                       embed_avg=sd[prefix + "embed_avg"].clone().float(),
                       cluster_size=sd[prefix + "cluster_size"].clone().float(),
                       initted=bool(sd[prefix + "initted"]))

    def clone(self):
        return VQState(self.embed.clone(), self.embed_avg.clone(), self.cluster_size.clone(), self.initted)


def update_ema(st: VQState, cfg: VQConfig):               # vqp.py:576-584
    cs = laplace(st.cluster_size, cfg.codebook_size, cfg.eps) * st.cluster_size.sum(-1, keepdim=True)
    e = st.embed_avg / cs[..., None]
    if cfg.use_cosine_sim:
        e = l2norm(e)
    st.embed.copy_(e)


def expire_codes(st: VQState, cfg: VQConfig, batch_samples, sample_fn=batched_sample_rows, seq_mask=None):
    """vqp.py:564-574 + replace :544-562.  batch_samples [H, n, D]."""
    if cfg.threshold_ema_dead_code <= 0:
        return
    expired = st.cluster_size < cfg.threshold_ema_dead_code
    if not bool(expired.any()):
        return
    reset = cfg.threshold_ema_dead_code if cfg.reset_cluster_size is None else cfg.reset_cluster_size
    if cfg.use_cosine_sim:
        batch_samples = l2norm(batch_samples)
    for h in range(batch_samples.shape[0]):
        samples = batch_samples[h]
        if seq_mask is not None:
            samples = samples[seq_mask[h]]
        if samples.numel() == 0:
            continue
        m = expired[h]
        picked = sample_fn(samples[None], int(m.sum().item()))[0].to(st.embed)
        st.embed[h][m] = picked
        st.cluster_size[h][m] = reset
        st.embed_avg[h][m] = picked * reset


def codebook_forward(st: VQState, cfg: VQConfig, x, *, training=True, mask=None, freeze_codebook=False,
                     ema_update_weight=None, assign_mode="aten", sample_fn=batched_sample_rows,
                     replace_sample_fn=batched_sample_rows, stats_mode="aten", quantize_mode="gather"):
    """Codebook.forward, vqp.py:673-791, for num_codebooks == 1 input [b, n, d] (or [h,b,n,d]).
    Returns (quantize fp32, embed_ind int64).  Mutates `st` like the reference mutates its buffers.
    quantize_mode: "gather" -- embed[ind], what the reference's eval branch does (:779-781) and bit-identical to its training
    branch; "onehot" -- the training branch AS THE REFERENCE RUNS IT: F.one_hot(ind, C).type(dtype) (:142, an N x C int64 tensor
    cast to fp32) contracted with the codebook (:766, the second N*C*D contraction of a training forward), the same one-hot tensor
    then feeding the EMA statistics (:602-606, the third).  bench.py's cpu_baseline times this mode."""
    needs_h = x.ndim < 4
    x = x.float()                                                   # :692
    if needs_h:
        x = x[None]
    H = x.shape[0]
    lead = x.shape[1:-1]
    flat = x.reshape(H, -1, x.shape[-1])                            # :698
    fmask = None
    if mask is not None:                                            # :700-701
        fmask = mask.reshape(1, -1).expand(H, -1)

    if not st.initted:                                              # :451-473
        data = flat
        if fmask is not None:
            data = flat[fmask].reshape(H, -1, flat.shape[-1])
        means, bins = kmeans(data, cfg.codebook_size, cfg.kmeans_iters, cfg.use_cosine_sim, sample_fn, assign_mode)
        st.embed_avg.copy_(means * bins[..., None])
        st.cluster_size.copy_(bins.to(st.cluster_size))
        update_ema(st, cfg)
        st.initted = True

    embed = st.embed.detach()                                       # :710-712
    ind = _assign(flat.detach(), embed, cfg.use_cosine_sim, assign_mode)      # :740-747

    # quantize = exact copy of the pre-update codebook row (:766 one-hot einsum / :779-781 gather)
    onehot_ref = None
    if quantize_mode == "onehot" and training:
        onehot_ref = F.one_hot(ind, cfg.codebook_size).type(flat.dtype)                     # :142
        quant = torch.einsum('hnc,hcd->hnd', onehot_ref, embed)                             # :766
    else:
        quant = torch.stack([embed[h][ind[h]] for h in range(H)], 0)

    if training and not freeze_codebook and (cfg.ema_update or cfg.threshold_ema_dead_code > 0):   # :783, :630
        C = cfg.codebook_size
        if stats_mode == "aten":                                    # :602-606, the reference's one-hot contraction
            onehot = onehot_ref if onehot_ref is not None else F.one_hot(ind, C).to(flat.dtype)
            if fmask is not None:
                onehot = onehot.masked_fill(~fmask[..., None], 0.)
            count = onehot.sum(1)
            esum = torch.einsum('hnd,hnc->hcd', flat.detach(), onehot).contiguous()
        else:                                                       # order-free (double) statistics
            cs, es = [], []
            for h in range(H):
                ii = ind[h] if fmask is None else torch.where(fmask[h], ind[h], torch.full_like(ind[h], -1))
                c_, e_ = c_ema_stats(flat[h], ii, C)
                cs.append(c_); es.append(e_)
            count, esum = torch.stack(cs), torch.stack(es)
        lerp_inplace(st.cluster_size, count, cfg.decay, ema_update_weight)     # :616
        lerp_inplace(st.embed_avg, esum, cfg.decay, ema_update_weight)         # :617
        if cfg.ema_update and not cfg.manual_ema_update:                        # :638-639
            update_ema(st, cfg)
        expire_codes(st, cfg, flat.detach(), replace_sample_fn, seq_mask=fmask)  # :641

    quant = quant.reshape(H, *lead, x.shape[-1])
    ind = ind.reshape(H, *lead)
    if needs_h:
        quant, ind = quant[0], ind[0]
    return quant, ind


def vq_forward(st: VQState, cfg: VQConfig, x, *, training=True, mask=None, lens=None, freeze_codebook=False,
               assign_mode="aten", stats_mode="aten", **cb_kw):
    """VectorQuantize.forward, vqp.py:1093-1403, channel-last [b, n, d] input, heads == 1,
    no projections.  -> (quantize [b,n,d] in x.dtype, indices [b,n] int64, loss scalar fp32)."""
    orig = x
    requires_grad = x.requires_grad
    if lens is not None:                                            # :1118-1119
        mask = torch.arange(x.shape[1])[None, :] < lens[:, None]
    only_one = x.ndim == 2
    if only_one:
        x = x[:, None, :]
    dtype = x.dtype
    if cfg.use_cosine_sim:                                          # :1159
        x = l2norm(x)
    quant, ind = codebook_forward(st, cfg, x, training=training, mask=mask, freeze_codebook=freeze_codebook,
                                  assign_mode=assign_mode, stats_mode=stats_mode, **cb_kw)
    quant = quant.type(dtype)                                       # :1178
    loss = torch.zeros((), dtype=torch.float32)
    if training:
        commit_q = quant.detach()                                   # :1214-1216
        if requires_grad:                                           # :1225-1233
            if cfg.rotation_trick:
                quant = rotate_to(x, quant)
            else:
                quant = x + (quant - x).detach()
        if cfg.commitment_weight > 0:                               # :1296-1329
            if mask is not None:
                l = F.mse_loss(commit_q, orig if not only_one else x, reduction='none')
                commit = l[mask].mean()
            else:
                commit = F.mse_loss(commit_q, x)
            loss = loss + commit * cfg.commitment_weight
    if only_one:
        quant, ind = quant[:, 0], ind[:, 0]
    if mask is not None:                                            # :1378-1396
        quant = torch.where(mask[..., None], quant, torch.zeros_like(quant))
        ind = torch.where(mask, ind, torch.full_like(ind, -1))
    return quant, ind, loss


# ----------------------------------------------------------------------------------------------
# ResidualVQ restatement (rvq.py:384-630; no dropout / beam / implicit codebook)
# ----------------------------------------------------------------------------------------------
def rvq_forward(states, cfg: VQConfig, x, *, shared_codebook=False, training=True, mask=None,
                freeze_codebook=False, assign_mode="aten", stats_mode="aten", **cb_kw):
    """states: list of Q VQState (for shared_codebook pass the SAME object Q times and
    cfg.manual_ema_update=True, as rvq.py:213-217, 302-306 do).
    -> (quantized_out [b,n,d], indices [b,n,Q] int64, losses [Q] fp32)."""
    out = torch.zeros_like(x)
    residual = x
    all_idx, all_loss, all_res = [], [], []
    for st in states:                                               # rvq.py:469
        all_res.append(residual.detach())
        q, ind, loss = vq_forward(st, cfg, residual, training=training, mask=mask,
                                  freeze_codebook=freeze_codebook, assign_mode=assign_mode,
                                  stats_mode=stats_mode, **cb_kw)
        residual = residual - q.detach()                            # :524 (quant_grad_frac = 0)
        out = out + q                                               # :525
        all_idx.append(ind)
        all_loss.append(loss)
    if training and shared_codebook:                                # :593-601
        st = states[0]
        if cfg.ema_update:
            update_ema(st, cfg)
        stacked = torch.stack(all_res, -2)                          # [b, n, Q, d]
        flat = stacked.reshape(1, -1, x.shape[-1])
        if cfg.use_cosine_sim:
            flat = l2norm(flat)
        expire_codes(st, cfg, flat, cb_kw.get("replace_sample_fn", batched_sample_rows))
    return out, torch.stack(all_idx, -1), torch.stack(all_loss)


def decode(embed2d, indices):                                       # vqp.py:1003, rvq.py:341-371
    """indices [...]; -1 -> zero row."""
    m = indices < 0
    q = embed2d[indices.masked_fill(m, 0)]
    return q.masked_fill(m[..., None], 0.)
