"""GPU parity tests of the C-ABI entry points against the oracle (oracle/vq_oracle.{c,py}).

Bars: indices bit-exact vs the deterministic ("chain") oracle; the winning distance bit-exact too
(which pins the MFMA k-order, the ATen-order norms, the association and the sqrt rounding);
floating-point statistics within 1e-5 relative (fp32) as BASELINE.json's north_star states.
"""
import os

import pytest
import torch

from oracle import vq_oracle as O

pytestmark = pytest.mark.gpu


def _mk(N, C, D, dtype=torch.float32, unit=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, D, generator=g)
    if unit:
        e = torch.randn(C, D, generator=g)
    else:  # kaiming-uniform-like tiny codebook: the worst case for near-ties (SURVEY §7 hard part 1)
        bound = (6.0 / (C * D)) ** 0.5
        e = (torch.rand(C, D, generator=g) * 2 - 1) * bound
    x = x.to(dtype)
    return x, e


@pytest.mark.parametrize("N,D", [(1000, 256), (777, 100), (64, 8), (333, 512), (5, 2), (4096, 128), (130, 40),
                                 (300, 520), (257, 768), (1000, 1024), (130, 1531), (64, 2048)])      # > 512: ATen's second cascade level
def test_row_sumsq_matches_aten_order(dev, N, D):
    from vector_quantize_pytorch_amd import _lib as L
    x, _ = _mk(N, 4, D)
    got = L.row_sumsq(x.to(dev)).cpu()
    want = (x ** 2).sum(-1)                      # live ATen on this host
    want_c = O.c_row_sumsq(x)                    # C restatement
    assert torch.equal(want, want_c)
    assert torch.equal(got, want_c)


@pytest.mark.parametrize("N,C,D,dtype,unit", [
    (1024, 512, 256, torch.float32, False),      # BASELINE cfg 1 shape
    (1024, 512, 256, torch.float32, True),
    (4099, 1024, 256, torch.bfloat16, False),    # cfg 2 dtype, ragged N
    (2048, 1024, 256, torch.float32, False),
    (513, 100, 64, torch.float32, True),         # C not a multiple of 32
    (300, 37, 100, torch.float32, True),         # odd D -> pre-pass norms + scalar loads
    (257, 5, 2, torch.float32, True),            # test_tiger-like tiny dims (rvq.py codebook sizes (5,128,256), dim 2)
    (1000, 4096, 128, torch.float32, False),     # cfg 5 per-group shape
    (640, 2048, 512, torch.float32, False),      # cfg 4 dim
    (1, 16, 32, torch.float32, True),
    # wide dims (csrc/vq_wide.hip, round 5): the plain exact kernel, same contract
    (1000, 300, 768, torch.float32, False),
    (513, 512, 1024, torch.float32, True),
    (300, 100, 640, torch.bfloat16, True),
    (129, 37, 2048, torch.float32, False),
    (200, 65, 1001, torch.float32, True),        # odd wide D
    (1000, 260, 776, torch.float32, False),      # a partial 32-feature slab AND a partial group of 128 codes on the 16-byte staging path
    (700, 130, 1000, torch.bfloat16, True),
])
def test_assign_euclid_bitexact(dev, N, C, D, dtype, unit):
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(N, C, D, dtype, unit)
    xd, ed = x.to(dev), e.to(dev)
    packed = L.pack_codebook(ed)
    r = L.assign(xd, packed, ed, want_q=True, want_sqerr=True, want_best=True, want_rnorm=True)
    idx_o, best_o = O.c_assign(x.float(), e)
    idx = r["idx"].cpu()
    mism = (idx != idx_o).sum().item()
    assert mism == 0, f"{mism}/{N} index mismatches vs chain oracle"
    assert torch.equal(r["best"].cpu(), best_o), "winning distance differs bitwise"
    assert torch.equal(r["rnorm"].cpu(), O.c_row_sumsq(x.float()))
    q = r["q"].cpu()
    want_q = e[idx_o].to(dtype)
    assert torch.equal(q, want_q)
    sq = r["sqerr_partials"][: r["nblk"]].sum().item()
    want_sq = ((want_q.double() - x.double()) ** 2).sum().item()
    assert abs(sq - want_sq) <= 1e-5 * max(want_sq, 1e-12)


@pytest.mark.parametrize("N,C,D,dtype", [(1000, 300, 768, torch.float32), (333, 70, 1024, torch.bfloat16), (129, 37, 1001, torch.float32)])
def test_wide_assign_q_rows_without_the_loss(dev, N, C, D, dtype):
    """The training forward of a wide-dim codebook asks for indices + q rows only (the loss comes from the statistics pass): the q rows
    are written four rows' 16-byte pieces at a time (odd D: element by element) -- the same rows as embed[idx], bit for bit."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(N, C, D, dtype, False)
    xd, ed = x.to(dev), e.to(dev)
    r = L.assign(xd, L.pack_codebook(ed), ed, want_q=True)
    idx_o, _ = O.c_assign(x.float(), e)
    assert torch.equal(r["idx"].cpu(), idx_o)
    assert torch.equal(r["q"].cpu(), e[idx_o].to(dtype))


@pytest.mark.parametrize("N,C,D,dtype", [(1024, 512, 256, torch.float32), (999, 1000, 512, torch.float32),
                                         (2048, 1024, 256, torch.bfloat16), (300, 37, 100, torch.float32),
                                         (400, 200, 1024, torch.float32), (300, 70, 768, torch.bfloat16)])      # wide dims
def test_assign_cosine_bitexact(dev, N, C, D, dtype):
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(N, C, D, dtype, unit=True)
    e = O.l2norm(e)
    xd, ed = x.to(dev), e.to(dev)
    packed = L.pack_codebook(ed)
    r = L.assign(xd, packed, ed, cosine=True, want_q=True, want_sqerr=True, want_best=True, want_rnorm=True)
    xn = O.c_l2norm(x.float())
    if dtype == torch.bfloat16:   # the reference normalises bf16 inputs in bf16 (vqp.py:1159 before :692)
        nrm = O.c_row_sumsq(x.float()).sqrt().bfloat16().float().clamp(min=1e-6)
        xn = (x.float() / nrm[:, None]).bfloat16().float()
    idx_o, best_o = O.c_assign(xn, e, cosine=True)
    assert (r["idx"].cpu() != idx_o).sum().item() == 0
    assert torch.equal(r["best"].cpu(), best_o)
    want_q = e[idx_o].to(dtype)
    assert torch.equal(r["q"].cpu(), want_q)
    sq = r["sqerr_partials"][: r["nblk"]].sum().item()
    want_sq = ((want_q.double() - xn.double()) ** 2).sum().item()
    assert abs(sq - want_sq) <= 1e-5 * max(want_sq, 1e-12)


def test_assign_ties_lowest_index(dev):
    """duplicate codes: the first copy must win (ATen argmax first-occurrence, vqp.py:140)."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(512, 96, 64, unit=True)
    e = torch.cat([e, e, e[:7]], 0)              # every code appears twice (or three times)
    ed = e.to(dev)
    r = L.assign(x.to(dev), L.pack_codebook(ed), ed)
    idx = r["idx"].cpu()
    assert int(idx.max()) < 96
    assert torch.equal(idx, O.c_assign(x, e)[0])


def test_assign_masked_rows_excluded_from_sqerr(dev):
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(700, 64, 32, unit=True)
    m = torch.rand(700) < 0.5
    ed = e.to(dev)
    r = L.assign(x.to(dev), L.pack_codebook(ed), ed, want_sqerr=True, row_mask=m.to(dev))
    idx = r["idx"].cpu()
    want = (((e[idx] - x) ** 2).sum(-1).double() * m).sum().item()
    assert abs(r["sqerr_partials"][: r["nblk"]].sum().item() - want) <= 1e-5 * want


def test_assign_strided_rows(dev):
    """feature-chunk views (GroupedResidualVQ's x.chunk(groups, -1), rvq.py:690) need no copy."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(600, 128, 64, unit=True)
    big = torch.randn(600, 256)
    big[:, 64:128] = x
    xd = big.to(dev)[:, 64:128]
    ed = e.to(dev)
    r = L.assign(xd, L.pack_codebook(ed), ed)
    assert torch.equal(r["idx"].cpu(), O.c_assign(x, e)[0])


@pytest.mark.parametrize("N,C,D,dtype,cos", [(5000, 1024, 256, torch.float32, False), (3000, 100, 40, torch.float32, False),
                                             (4096, 2500, 128, torch.bfloat16, False), (2000, 512, 256, torch.float32, True),
                                             (3000, 200, 768, torch.float32, False), (2000, 100, 2048, torch.bfloat16, False)])   # wide dims
def test_ema_accumulate(dev, N, C, D, dtype, cos):
    from vector_quantize_pytorch_amd import _lib as L
    x, _ = _mk(N, C, D, dtype, unit=True)
    idx = torch.randint(0, C, (N,))
    idx[::17] = -1                                # masked / dropped rows are skipped
    rn = None
    xf = x.float()
    if cos:
        rn = O.c_row_sumsq(xf).sqrt().clamp(min=1e-6)
        xf = xf / rn[:, None]
    cnt, es = L.ema_accumulate(x.to(dev), idx.to(dev), C, cosine=cos, rnorm=None if rn is None else rn.to(dev))
    cnt_o, es_o = O.c_ema_stats(xf, idx, C)
    assert torch.equal(cnt.cpu(), cnt_o)
    scale = es_o.abs().max().item()
    assert (es.cpu() - es_o).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize("N,C,D,dtype", [(20000, 1024, 256, torch.bfloat16), (20000, 512, 256, torch.float32),
                                         (5000, 300, 512, torch.float32), (5000, 64, 32, torch.bfloat16), (777, 1000, 128, torch.bfloat16),
                                         (3000, 100, 1024, torch.float32), (2000, 64, 640, torch.bfloat16)])      # wide dims
def test_ema_accumulate_sqerr_equals_the_search_kernels_loss(dev, N, C, D, dtype):
    """vqhip_ema_accumulate_sqerr: the statistics pass also sums ||q - x||^2 (reference: F.mse_loss numerator, vqp.py:1327).  Against
    (a) the oracle formula on the CPU in double, (b) the sum the exact search kernel produces for the same rows and indices;
    count / embed_sum must be what the plain entry point gives; masked rows (mask, idx < 0) contribute nothing."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(N, C, D, dtype)
    xd, ed = x.to(dev), e.to(dev).contiguous()
    packed = L.pack_codebook(ed)
    mask = torch.rand(N) > 0.1
    r = L.assign(xd, packed, ed, want_q=True, want_sqerr=True, row_mask=mask.to(dev))
    idx = r["idx"]
    assert L.stats_sqerr_supported(xd)
    cnt, es, parts = L.ema_accumulate(xd, idx, C, row_mask=mask.to(dev), sqerr_from=(packed, ed))
    cnt0, es0 = L.ema_accumulate(xd, idx, C, row_mask=mask.to(dev))
    assert torch.equal(cnt, cnt0) and (es - es0).abs().max().item() <= 1e-5 * es0.abs().max().item()   # (fp32 atomics over a code's chunks)
    got = parts.sum().item()
    q = (ed.to(dtype) if dtype == torch.bfloat16 else ed)[idx].double().cpu()         # the q rows of this dtype
    want = ((q - x.double()) ** 2).sum(-1)[mask].sum().item()
    assert abs(got - want) <= 1e-6 * want
    from_search = r["sqerr_partials"][: r["nblk"]].sum().item()
    assert abs(got - from_search) <= 1e-6 * from_search                                  # fp32 chains per batch of rows, then double
    # rows dropped through idx < 0 are skipped by both the statistics and the loss
    idx2 = idx.clone(); idx2[::5] = -1
    _, _, parts2 = L.ema_accumulate(xd, idx2, C, sqerr_from=(packed, ed))
    keep = torch.ones(N, dtype=torch.bool); keep[::5] = False
    want2 = ((q - x.double()) ** 2).sum(-1)[keep].sum().item()
    assert abs(parts2.sum().item() - want2) <= 1e-6 * want2


@pytest.mark.parametrize("C,D,cos", [(512, 256, False), (1024, 256, False), (1000, 100, False), (4096, 128, True), (37, 2, False),
                                     (300, 768, False), (64, 2048, False), (100, 1024, True)])      # wide dims: vq_wide_embed_kernel
def test_ema_finalize(dev, C, D, cos):
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    cfg = O.VQConfig(dim=D, codebook_size=C, use_cosine_sim=cos)
    st = O.VQState(embed=torch.randn(1, C, D, generator=g), embed_avg=torch.randn(1, C, D, generator=g),
                   cluster_size=torch.rand(1, C, generator=g) * 5)
    count = torch.randint(0, 9, (1, C), generator=g).float()
    esum = torch.randn(1, C, D, generator=g) * 3
    d = {k: v.clone().to(dev) for k, v in dict(cs=st.cluster_size[0], ea=st.embed_avg[0], e=st.embed[0]).items()}
    L.ema_finalize(d["cs"], d["ea"], d["e"], count[0].to(dev), esum[0].to(dev), decay=cfg.decay, eps=cfg.eps, cosine=cos)
    O.lerp_inplace(st.cluster_size, count, cfg.decay)
    O.lerp_inplace(st.embed_avg, esum, cfg.decay)
    O.update_ema(st, cfg)
    assert torch.equal(d["cs"].cpu(), st.cluster_size[0])          # ATen lerp_ restated exactly
    assert torch.equal(d["ea"].cpu(), st.embed_avg[0])
    if cos:
        assert (d["e"].cpu() - st.embed[0]).abs().max().item() <= 1e-6
    else:
        assert torch.equal(d["e"].cpu(), st.embed[0])              # incl. ATen-order sum of cluster_size


@pytest.mark.parametrize("C,D,Q,cos", [(1024, 256, 8, False), (300, 100, 3, False), (64, 32, 5, True)])
def test_ema_fold_many_equals_successive_folds(dev, C, D, Q, cos):
    """vqhip_ema_fold_many (the Q folds of a codebook shared by the stages of a residual VQ + one renormalisation, rvq.py:213-217,
    593-598) == Q calls of vqhip_ema_finalize(do_lerp) followed by one update_ema, bit for bit; vqhip_reduce_partials_rows == rows
    of vqhip_reduce_partials."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator(device=dev).manual_seed(C + Q)
    cs = torch.rand(C, device=dev, generator=g) * 5
    ea = torch.randn(C, D, device=dev, generator=g)
    e = torch.randn(C, D, device=dev, generator=g)
    stride = (C * D + C + 3) // 4 * 4
    stats = torch.zeros(Q, stride, device=dev)
    stats[:, : C * D] = torch.randn(Q, C * D, device=dev, generator=g) * 3
    stats[:, C * D: C * D + C] = torch.randint(0, 9, (Q, C), device=dev, generator=g).float()
    a = [t.clone() for t in (cs, ea, e)]
    b = [t.clone() for t in (cs, ea, e)]
    L.ema_fold_many(*a, stats, decay=0.8, eps=1e-5, cosine=cos, do_update_ema=True)
    for q in range(Q):
        L.ema_finalize(*b, stats[q, C * D: C * D + C], stats[q, : C * D].view(C, D), decay=0.8, eps=1e-5, cosine=cos, do_lerp=True, do_update_ema=False)
    L.ema_finalize(*b, None, None, decay=0.8, eps=1e-5, cosine=cos, do_lerp=False, do_update_ema=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    c2 = [t.clone() for t in (cs, ea, e)]
    L.ema_fold_many(*c2, stats, decay=0.8, eps=1e-5, cosine=cos, do_update_ema=False)      # folds only: embed untouched
    assert torch.equal(c2[2], e) and torch.equal(c2[0], a[0]) and torch.equal(c2[1], a[1])
    parts = torch.randn(Q, 777, device=dev, generator=g, dtype=torch.float64)
    rows = L.reduce_partials_rows(parts, 0.5)
    one = torch.stack([L.reduce_partials(parts[q], 777, 0.5) for q in range(Q)])
    assert torch.equal(rows, one)


@pytest.mark.parametrize("D", [96, 768, 2048])          # > 512: vq_wide_decode_kernel
def test_decode_sum(dev, D):
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    Q, C = 4, 50
    cb = torch.randn(Q, C, D, generator=g)
    idx = torch.randint(0, C, (3, 70, Q), generator=g)
    idx[0, :5, 2:] = -1
    out = L.decode_sum(idx.to(dev), cb.to(dev)).cpu()
    want = torch.zeros(3, 70, D)
    for q in range(Q):
        want = want + O.decode(cb[q], idx[..., q])
    assert torch.equal(out, want)
    out1 = L.decode_sum(idx.to(dev), cb[0].contiguous().to(dev)).cpu()
    want1 = sum(O.decode(cb[0], idx[..., q]) for q in range(Q))
    assert (out1 - want1).abs().max().item() <= 1e-6


@pytest.mark.parametrize("N,Q,C,D,out_dtype", [(20000, 8, 1024, 256, torch.float32), (16385, 3, 37, 64, torch.float32), (70001, 11, 512, 128, torch.bfloat16),
                                               (32768, 2, 1000, 512, torch.float32)])
def test_decode_sum_shared_codebook_from_lds(dev, N, Q, C, D, out_dtype):
    """vq_decode_lds_kernel (shared codebook, column slices of the codes in LDS): bit-equal to the stage-order running sum of the
    gathered rows (rvq.py:525), dropped stages (-1) add nothing, ragged row counts, more than 8 stages."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator(device=dev).manual_seed(N)
    e = torch.randn(C, D, device=dev, generator=g)
    idx = torch.randint(0, C, (N, Q), device=dev, generator=g)
    idx[::7, Q - 1] = -1
    idx[5, :] = -1
    out = L.decode_sum(idx, e, out_dtype=out_dtype)
    want = torch.zeros(N, D, device=dev)
    for q in range(Q):
        ok = (idx[:, q] >= 0)[:, None]
        want = torch.where(ok, want + e[idx[:, q].clamp(min=0)], want)
    assert torch.equal(out, want.to(out_dtype))


@pytest.mark.parametrize("N,C,D,Q,dtype,shared", [(300, 64, 512, 3, torch.float32, True), (515, 100, 128, 4, torch.float32, False),
                                                   (1000, 256, 256, 8, torch.bfloat16, True), (129, 33, 32, 2, torch.float32, False)])
def test_fused_rvq_kernel_vs_stagewise_oracle(dev, N, C, D, Q, dtype, shared):
    """vqhip_rvq_forward == Q successive oracle assignments on the running residual (rvq.py:469-568),
    incl. the bf16 arithmetic of the reference when the input is bf16 (quantized and residual are bf16 tensors)."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, D, generator=g).to(dtype)
    cbs = torch.randn(1 if shared else Q, C, D, generator=g)
    m = torch.rand(N, generator=g) < 0.9
    xd = x.to(dev)
    if shared:
        e = cbs[0].contiguous().to(dev)
        pk = L.pack_codebook(e)
    else:
        e = cbs.contiguous().to(dev)
        pk = torch.stack([L.pack_codebook(e[q]) for q in range(Q)])
    r = L.rvq_forward(xd, pk, e, Q, want_resid=True, want_sqerr=True, row_mask=m.to(dev))
    res = x.clone()
    for q in range(Q):
        cb = cbs[0 if shared else q]
        assert torch.equal(r["resid"][:, q].cpu(), res), f"stage {q} input residual"
        idx_o, _ = O.c_assign(res.float(), cb)
        want_idx = torch.where(m, idx_o, torch.full_like(idx_o, -1))
        assert torch.equal(r["idx"][:, q].cpu(), want_idx), f"stage {q} indices"
        quant = cb[idx_o].to(dtype)
        want_sq = (((quant.double() - res.double()) ** 2).sum(-1) * m).sum().item()
        got_sq = r["sqerr_partials"][q].sum().item()
        assert abs(got_sq - want_sq) <= 1e-5 * max(want_sq, 1e-12), f"stage {q} squared error"
        res = torch.where(m[:, None], res - quant, res)
    out = L.decode_sum(r["idx"], e, out_dtype=torch.float32).cpu()
    want = torch.zeros(N, D)
    for q in range(Q):
        want = want + O.decode(cbs[0 if shared else q], r["idx"][:, q].cpu())
    assert torch.equal(out, want)


def test_assign_fuzz_shapes_against_chain_oracle(dev):
    """60 random (N, C, D, dtype, metric, scale) draws, including C = 1, N = 1, D = 1 .. 512 (odd sizes take the
    pre-pass / scalar-load paths), tiny and huge magnitudes: indices and winning scores bit-exact every time."""
    import random
    from vector_quantize_pytorch_amd import _lib as L
    rnd = random.Random(1234)
    g = torch.Generator().manual_seed(99)
    for it in range(60):
        N = rnd.choice([1, 2, 31, 33, 127, 129, 500, 1000])
        C = rnd.choice([1, 2, 31, 32, 33, 100, 257, 1000])
        D = rnd.choice([1, 2, 3, 7, 8, 31, 32, 33, 64, 96, 100, 128, 200, 256, 300, 512])
        cosine = rnd.random() < 0.3
        dtype = torch.bfloat16 if rnd.random() < 0.3 else torch.float32
        scale = rnd.choice([1e-4, 1.0, 30.0])
        x = (torch.randn(N, D, generator=g) * scale).to(dtype)
        e = torch.randn(C, D, generator=g) * rnd.choice([1e-3, 1.0])
        if cosine:
            e = O.l2norm(e)
        xd, ed = x.to(dev), e.to(dev)
        r = L.assign(xd, L.pack_codebook(ed), ed, cosine=cosine, want_q=True, want_best=True)
        xf = x.float()
        if cosine:
            if dtype == torch.bfloat16:
                nrm = O.c_row_sumsq(xf).sqrt().bfloat16().float().clamp(min=1e-6)
                xf = (xf / nrm[:, None]).bfloat16().float()
            else:
                xf = O.c_l2norm(xf)
        io, bo = O.c_assign(xf, e, cosine)
        tag = f"draw {it}: N={N} C={C} D={D} cosine={cosine} {dtype} scale={scale}"
        assert torch.equal(r["idx"].cpu(), io), tag
        assert torch.equal(r["best"].cpu(), bo), tag
        assert torch.equal(r["q"].cpu(), e[io].to(dtype)), tag


def test_assign_exact_duplicates_and_zero_rows(dev):
    """rows equal to a code (distance clamps to sqrt(1e-8)), all-zero rows, all-equal codebook."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    e = torch.randn(64, 64, generator=g)
    x = torch.cat([e[[3, 17, 63]], torch.zeros(2, 64), torch.randn(27, 64, generator=g)])
    ed = e.to(dev)
    r = L.assign(x.to(dev), L.pack_codebook(ed), ed, want_best=True)
    io, bo = O.c_assign(x, e)
    assert torch.equal(r["idx"].cpu(), io) and torch.equal(r["best"].cpu(), bo)
    assert r["idx"][:3].tolist() == [3, 17, 63]
    e2 = torch.ones(40, 32) * 0.5                     # every code identical: index 0 must win everywhere
    r2 = L.assign(torch.randn(100, 32, generator=g).to(dev), L.pack_codebook(e2.to(dev)), e2.to(dev))
    assert int(r2["idx"].max()) == 0


@pytest.mark.parametrize("N,C,D,cos", [(300, 100, 64, False), (257, 33, 256, False), (128, 512, 128, True), (77, 40, 100, False)])
def test_dense_scores_match_oracle_bitwise(dev, N, C, D, cos):
    """vqhip_scores == the reference's `dist` tensor in the oracle's chain arithmetic (-cdist / cosine similarity)."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(N, C, D, unit=True)
    if cos:
        e = O.l2norm(e)
    ed = e.to(dev)
    dist, idx, _ = L.scores(x.to(dev), L.pack_codebook(ed), ed, cosine=cos)
    xo = O.c_l2norm(x) if cos else x
    want = O.c_scores(xo, e, cos)
    assert torch.equal(dist.cpu(), want)
    assert torch.equal(idx.cpu(), want.argmax(-1))


# ---- screened assignment (csrc/vq_screen.hip): bf16 rows, bf16-MFMA screen + exact pass on the uncertified rows --------
def _screen_case(N, C, D, kind, seed=0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, D, generator=g).to(dtype)
    if kind == "kaiming":      # the reference's default init (vqp.py:28-31): tiny codes, the near-tie worst case
        e = (torch.rand(C, D, generator=g) * 2 - 1) * (6.0 / D) ** 0.5
    elif kind == "unit":
        e = torch.randn(C, D, generator=g)
    elif kind == "rows":       # codes drawn from the data, as after k-means init / dead-code replacement
        e = x[torch.randperm(N, generator=g)[:C]].float().contiguous()
    elif kind == "dups":       # duplicated codes: no row whose best code has a twin can be certified
        e = torch.randn(C, D, generator=g)
        e[C // 2:] = e[: C - C // 2]
    elif kind == "tiny":       # collapsed codebook, score gaps close to the rounding level
        e = torch.randn(C, D, generator=g) * 1e-3
    elif kind == "bigcode":    # the reference's default init with ONE code 100 x larger (round 6: the certificate is charged per code)
        e = (torch.rand(C, D, generator=g) * 2 - 1) * (6.0 / D) ** 0.5
        e[7 % C] *= 100.0
    elif kind == "normspread":  # code norms spread over three decades, rows scattered around the codes at their code's scale
        sc = torch.logspace(-1.5, 1.5, C)[torch.randperm(C, generator=g)]
        e = torch.randn(C, D, generator=g) * sc[:, None]
        pick = torch.randint(0, C, (N,), generator=g)
        x = (e[pick] + 0.3 * sc[pick][:, None] * torch.randn(N, D, generator=g)).to(dtype)
    elif kind == "zeros":      # several (near-)zero codes among ordinary ones: k-means seeds drawn from residuals that vanished
        e = torch.randn(C, D, generator=g)
        e[::7] *= 1e-5
        e[3] = 0.0
    else:
        raise ValueError(kind)
    return x, e


@pytest.mark.parametrize("N,C,D,kind", [
    (4099, 1024, 256, "kaiming"),    # cfg 2 shape, ragged N
    (5000, 1000, 256, "unit"),       # C not a multiple of 32
    (300, 37, 128, "unit"),
    (20000, 512, 64, "rows"),
    (8192, 1024, 256, "dups"),
    (8192, 1024, 256, "tiny"),
    (1000, 2, 64, "unit"),
    (3000, 4096, 128, "kaiming"),    # cfg 5 per-group shape
    (5000, 8192, 32, "unit"),        # low-dimensional codebook (codebook_dim = 32)
    (2500, 100, 32, "rows"),
    (3000, 2048, 512, "unit"),       # cfg 4's dimension: one row block per wave
    (1500, 300, 512, "kaiming"),
    (8192, 1024, 256, "bigcode"),    # round 6: per-code certificate
    (8192, 1024, 256, "normspread"),
    (6000, 1000, 128, "normspread"),
    (4000, 500, 512, "normspread"),
    (8192, 1024, 128, "zeros"),
    (3000, 300, 64, "bigcode"),
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_screened_assign_matches_chain_oracle(dev, N, C, D, kind, dtype):
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _screen_case(N, C, D, kind, dtype=dtype)
    xd, ed = x.to(dev), e.to(dev)
    r = L.assign(xd, L.pack_codebook(ed), ed, want_q=True, want_sqerr=True)
    assert r.get("n_exact") is not None, "rows with D in {32,64,128,256,512} must take the screened path"
    idx_o, _ = O.c_assign(x.float(), e)
    mism = (r["idx"].cpu() != idx_o).sum().item()
    assert mism == 0, f"{mism}/{N} index mismatches vs chain oracle"
    want_q = e[idx_o].to(dtype)
    assert torch.equal(r["q"].cpu(), want_q)
    sq = r["sqerr_partials"][: r["nblk"]].sum().item()
    want_sq = ((want_q.double() - x.double()) ** 2).sum().item()
    assert abs(sq - want_sq) <= 1e-5 * max(want_sq, 1e-12)
    n_exact = int(r["n_exact"].item()) + int(r["n_pair"].item())      # rows of the full exact sweep + rows decided between two codes
    if kind == "dups":
        assert n_exact == N            # every best code has an identical twin: no row can be certified by the screen alone
    elif kind in ("kaiming", "unit", "rows"):
        assert n_exact <= 0.1 * N      # the screen certifies the bulk (observed: 0.3 .. 5 %)
    elif kind == "bigcode" and D == 256:
        # one large-norm code must not raise every row's threshold (round 5: 99.8 % of the rows of such a codebook took the exact
        # sweep): the uncertified share stays what the same codebook without the large code has
        x0, e0 = _screen_case(N, C, D, "kaiming", dtype=dtype)
        r0 = L.assign(x0.to(dev), L.pack_codebook(e0.to(dev)), e0.to(dev), want_q=False)
        base = int(r0["n_exact"].item()) + int(r0["n_pair"].item())
        assert n_exact <= 1.5 * base + 0.01 * N, f"one large code: {n_exact} uncertified rows vs {base} without it"
    elif kind == "normspread":
        assert n_exact <= 0.2 * N


def test_screened_scores_stay_inside_certified_bound(dev):
    """|screen score - exact score| must be far below the certification threshold (csrc/vq_screen.hip header):
    the bound is a pessimistic model of the MFMA's internal accumulation, this measures the real thing."""
    from vector_quantize_pytorch_amd import _lib as L
    worst = 0.0
    for (N, C, D, kind, dtype) in [(8192, 1024, 256, "kaiming", torch.bfloat16), (8192, 1024, 256, "unit", torch.bfloat16),
                                   (8192, 512, 64, "rows", torch.bfloat16), (8192, 1024, 128, "tiny", torch.bfloat16),
                                   (8192, 1024, 256, "kaiming", torch.float32), (8192, 1000, 128, "unit", torch.float32),
                                   (8192, 512, 64, "rows", torch.float32),
                                   (8192, 1024, 256, "normspread", torch.bfloat16), (8192, 1024, 128, "normspread", torch.float32),
                                   (8192, 1024, 256, "bigcode", torch.bfloat16), (4096, 512, 512, "normspread", torch.float32)]:
        x, e = _screen_case(N, C, D, kind, seed=3, dtype=dtype)
        xd, ed = x.to(dev), e.to(dev)
        L.screen_debug = True
        try:
            r = L.assign(xd, L.pack_codebook(ed), ed, want_q=False)
        finally:
            L.screen_debug = False
        dbg = r["screen_debug"].double().cpu()
        y2 = O.c_row_sumsq(e).double()
        t = x.double() @ e.double().t() - 0.5 * y2[None, :]       # what the screen approximates
        top = t.topk(2, dim=1).values
        cert = dbg[:, 3] == 0
        if kind in ("normspread", "bigcode"):
            # (the kernels rank the codes by an UPPER bound of their score -- score + the code's own error allowance, round 6 -- so with
            #  code norms decades apart the runner-up by that ranking need not be the runner-up by score: measure the certified winner)
            won = t.gather(1, r["idx"].cpu()[:, None])[:, 0]
            err = (dbg[:, 0] - won).abs()[cert]
            worst = max(worst, float((err / dbg[:, 2][cert]).max()))
            assert bool((won[cert] == top[:, 0][cert]).all())
        else:
            err = torch.maximum((dbg[:, 0] - top[:, 0]).abs(), (dbg[:, 1] - top[:, 1]).abs())
            worst = max(worst, float((err / dbg[:, 2]).max()))
            # every row the screen certified really has a margin above the threshold in exact arithmetic too
            assert bool(((top[:, 0] - top[:, 1])[cert] > 0.5 * dbg[:, 2][cert]).all())
    # typical rows sit far below the threshold; the worst case here is a code that EQUALS the row ("rows" codebooks) under the
    # fp32-row kernel, whose truncating x split errs with the sign of x -- coherent with c = x, so its 2^-20 X Y term is nearly
    # attained (0.28 of the threshold, which also carries the second code's share and the other terms)
    assert worst < 0.5, f"screen error reaches {worst:.3f} of the certified threshold"


def test_screened_masked_strided_rows_and_toggle(dev, monkeypatch):
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _screen_case(3000, 256, 128, "unit", seed=5)
    big = torch.randn(3000, 512).bfloat16()
    big[:, 128:256] = x
    xd = big.to(dev)[:, 128:256]                       # strided view, 16-byte aligned rows
    m = torch.rand(3000) < 0.5
    ed = e.to(dev)
    packed = L.pack_codebook(ed)
    r = L.assign(xd, packed, ed, want_q=True, want_sqerr=True, row_mask=m.to(dev))
    assert r.get("n_exact") is not None
    idx_o, _ = O.c_assign(x.float(), e)
    assert torch.equal(r["idx"].cpu(), idx_o)
    want = (((e[idx_o].bfloat16().double() - x.double()) ** 2).sum(-1) * m).sum().item()
    assert abs(r["sqerr_partials"][: r["nblk"]].sum().item() - want) <= 1e-5 * want
    monkeypatch.setenv("VQHIP_SCREEN", "0")            # the exact kernel on the same input: identical outputs
    r0 = L.assign(xd, packed, ed, want_q=True, want_sqerr=True, row_mask=m.to(dev))
    assert r0.get("n_exact") is None
    assert torch.equal(r0["idx"], r["idx"]) and torch.equal(r0["q"], r["q"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("kind", ["unit", "dups"])
def test_screened_residual_output(dev, dtype, kind):
    """resid_out = x - q in the reference's tensor arithmetic (rvq.py:524), from the screen kernel for certified rows and
    from the finish kernel for the rows of the exact pass ('dups' sends every row there)."""
    from vector_quantize_pytorch_amd import _lib as L
    N, C, D = 3001, 256, 128
    x, e = _screen_case(N, C, D, kind, seed=7, dtype=dtype)
    xd, ed = x.to(dev), e.to(dev)
    big = torch.zeros(N, 3, D, dtype=dtype, device=dev)           # strided destination rows, like the RVQ stage buffers
    r = L.assign(xd, L.pack_codebook(ed), ed, want_q=False, want_sqerr=True, resid_out=big[:, 1, :])
    assert r.get("n_exact") is not None and r["q"] is None
    idx_o, _ = O.c_assign(x.float(), e)
    assert torch.equal(r["idx"].cpu(), idx_o)
    want = x - e[idx_o].to(dtype)                                  # torch bf16 subtraction == fp32 subtract + RNE
    assert torch.equal(big[:, 1, :].cpu(), want)
    assert not big[:, 0, :].any() and not big[:, 2, :].any()


@pytest.mark.parametrize("N,C,D,Q,dtype,shared", [(3000, 256, 256, 4, torch.float32, True), (2049, 100, 128, 3, torch.bfloat16, False),
                                                   (1500, 512, 64, 8, torch.bfloat16, True)])
def test_rvq_screened_loop_equals_fused_exact_kernel(dev, N, C, D, Q, dtype, shared):
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, N // 2, D, generator=g).to(dtype).to(dev)
    e = (torch.randn(C, D, generator=g) if shared else torch.randn(Q, C, D, generator=g)).to(dev)
    packed = L.pack_codebook(e) if shared else torch.stack([L.pack_codebook(e[q]) for q in range(Q)])
    mask = (torch.rand(2, N // 2, generator=g) < 0.8).to(dev)
    a = L.rvq_forward(x, packed, e, Q, want_resid=True, want_sqerr=True, row_mask=mask)
    b = L.rvq_forward_screened(x, packed, e, Q, want_resid=True, want_sqerr=True, row_mask=mask)
    assert torch.equal(a["idx"], b["idx"])
    for q in range(Q):
        rows = mask.reshape(-1)
        assert torch.equal(a["resid"][..., q, :].reshape(-1, D)[rows], b["inputs"][q].reshape(-1, D)[rows])
        sa, sb = a["sqerr_partials"][q].sum().item(), b["sqerr_partials"][q].sum().item()
        assert abs(sa - sb) <= 1e-5 * max(abs(sa), 1e-12)


def _l2norm_ref(x):
    """the reference's l2norm on a tensor of x's dtype (vqp.py:37-38): bf16 tensors round the norm and the quotient to bf16"""
    if x.dtype == torch.bfloat16:
        nrm = O.c_row_sumsq(x.float()).sqrt().bfloat16().float().clamp(min=1e-6)
        return (x.float() / nrm[:, None]).bfloat16()
    return O.c_l2norm(x)


@pytest.mark.parametrize("N,D,dtype", [(4099, 256, torch.bfloat16), (1000, 128, torch.float32), (333, 64, torch.bfloat16),
                                       (2048, 256, torch.float32), (700, 32, torch.bfloat16), (700, 32, torch.float32)])
def test_l2norm_rows_matches_reference_arithmetic(dev, N, D, dtype):
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(N, D, generator=g) * torch.rand(N, 1, generator=g) * 3).to(dtype)
    x[5] = 0                                                       # zero row: divided by the 1e-6 floor
    assert torch.equal(L.l2norm_rows(x.to(dev)).cpu(), _l2norm_ref(x))


@pytest.mark.parametrize("B,R,S,dtype", [(3, 256, 1024, torch.float32), (3, 256, 1024, torch.bfloat16), (2, 64, 128, torch.float32),
                                         (5, 100, 333, torch.float32), (5, 100, 333, torch.bfloat16), (1, 7, 5, torch.float32),
                                         (2, 130, 260, torch.bfloat16), (4, 32, 4096, torch.float16), (2, 512, 72, torch.float32),
                                         (1, 2, 70000, torch.int32)])
def test_transposing_copy_is_the_contiguous_copy_of_a_transposed_view(dev, B, R, S, dtype):
    """vqhip_transpose_batched ([B, R, S] -> [B, S, R], channel-first callers: vqp.py:1136-1147) moves bits: equal to ATen's
    .contiguous() of the transposed view for full tiles (16-byte accesses) and ragged / unaligned shapes (guarded element path),
    4- and 2-byte elements, also from a view with a storage offset."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(4)
    base = torch.randint(-30000, 30000, (B + 1, R, S), generator=g)
    base = base.to(dtype) if dtype == torch.int32 else (base.float() / 7).to(dtype)
    for t in (base[:B].to(dev), base.to(dev)[1:]):
        v = t.transpose(1, 2)
        assert L.is_transposed_view(v)
        out = L.transpose_rows(v)
        assert out.is_contiguous() and out.shape == v.shape and torch.equal(out, v.contiguous())
        assert L.rows_contiguous(v).is_contiguous() and L.rows_contiguous(out) is out
    assert not L.is_transposed_view(base.to(dev)[:, :, ::2].transpose(1, 2))
    if R >= 8:                                                  # a channel group of a wider map: batches further apart than R * S
        grp = base.to(dev)[:, R // 4:R // 4 + R // 2].transpose(1, 2)
        assert L.is_transposed_view(grp) and torch.equal(L.transpose_rows(grp), grp.contiguous())


@pytest.mark.parametrize("N,D,dtype", [(4099, 256, torch.bfloat16), (1000, 128, torch.float32), (333, 64, torch.bfloat16),
                                       (2048, 256, torch.float32), (700, 32, torch.float32), (515, 512, torch.float32),
                                       (515, 512, torch.bfloat16)])
def test_l2norm_rows_backward_is_autograds_normalize_gradient(dev, N, D, dtype):
    """vqhip_l2norm_rows_bwd against autograd through F.normalize (vqp.py:37-38) evaluated in float64 on the same values: fp32 to
    rounding, bf16 to one rounding of the result; a zero row and a row below the eps floor take the clamp's branch (gradient g / eps)."""
    from vector_quantize_pytorch_amd import _lib as L
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(N, D, generator=gen) * torch.rand(N, 1, generator=gen) * 3).to(dtype)
    x[5] = 0
    x[6] = (torch.randn(D, generator=gen) * 1e-9).to(dtype)
    g = torch.randn(N, D, generator=gen).to(dtype)
    x64 = x.double().requires_grad_(True)
    torch.nn.functional.normalize(x64, p=2, dim=-1, eps=1e-6).backward(g.double())
    want = x64.grad
    got = L.l2norm_rows_bwd(x.to(dev), g.to(dev)).cpu().double()
    scale = want.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2       # bf16: the norm is the bf16-rounded one of the forward, the result rounded once
    assert ((got - want).abs() / scale).max().item() < tol
    assert L.l2norm_rows_supported(x.to(dev))


@pytest.mark.parametrize("N,C,D,kind", [(4099, 1024, 256, "unit"), (5000, 1000, 128, "unit"), (3000, 37, 64, "unit"),
                                        (8192, 1024, 256, "dups"), (4000, 2048, 32, "unit"), (2000, 4096, 512, "unit")])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_screened_cosine_matches_chain_oracle(dev, N, C, D, kind, dtype):
    """cosine metric through the screen: l2norm_rows + screened search on unit-norm rows == the exact cosine kernel's
    reference arithmetic (indices, q, squared error against the normalised rows)."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _screen_case(N, C, D, kind, dtype=dtype)
    e = O.l2norm(e)
    xd, ed = x.to(dev), e.to(dev)
    xn_d = L.l2norm_rows(xd)
    r = L.assign(xn_d, L.pack_codebook(ed), ed, cosine=True, skip_l2norm=True, want_q=True, want_sqerr=True)
    assert r.get("n_exact") is not None
    xn = _l2norm_ref(x)
    assert torch.equal(xn_d.cpu(), xn)
    idx_o, _ = O.c_assign(xn.float(), e, cosine=True)
    mism = (r["idx"].cpu() != idx_o).sum().item()
    assert mism == 0, f"{mism}/{N} index mismatches vs chain oracle"
    want_q = e[idx_o].to(dtype)
    assert torch.equal(r["q"].cpu(), want_q)
    sq = r["sqerr_partials"][: r["nblk"]].sum().item()
    want_sq = ((want_q.double() - xn.double()) ** 2).sum().item()
    assert abs(sq - want_sq) <= 1e-5 * max(want_sq, 1e-12)
    if kind == "dups":
        assert int(r["n_exact"].item()) + int(r["n_pair"].item()) == N


def test_screened_large_codebook_and_nonfinite_rows(dev, monkeypatch):
    """C = 65536 (2048 tiles, many near-ties) and rows containing NaN / inf: the screened path must return exactly what
    the exact kernel returns (non-finite rows are never certified and fall to the exact pass)."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(9)
    N, C, D = 3000, 65536, 128
    x = torch.randn(N, D, generator=g).bfloat16()
    x[7, 3] = float("nan")
    x[100, :] = float("inf")
    x[200, 5] = -float("inf")
    e = torch.randn(C, D, generator=g) * 0.3
    xd, ed = x.to(dev), e.to(dev)
    packed = L.pack_codebook(ed)
    r1 = L.assign(xd, packed, ed, want_q=True)
    assert r1.get("n_exact") is not None
    monkeypatch.setenv("VQHIP_SCREEN", "0")
    r0 = L.assign(xd, packed, ed, want_q=True)
    assert r0.get("n_exact") is None
    assert torch.equal(r0["idx"], r1["idx"]) and torch.equal(r0["q"], r1["q"])
    fin = torch.isfinite(x.float()).all(-1)
    idx_o, _ = O.c_assign(x.float()[fin], e)
    assert torch.equal(r1["idx"].cpu()[fin], idx_o)


def test_screen_verify_switch(dev, monkeypatch):
    """VQHIP_SCREEN_VERIFY=1 re-runs the exact kernel behind every screened search and raises on any disagreement."""
    from vector_quantize_pytorch_amd import _lib as L
    monkeypatch.setenv("VQHIP_SCREEN_VERIFY", "1")
    for dtype, cos in ((torch.bfloat16, False), (torch.float32, False), (torch.bfloat16, True)):
        x, e = _screen_case(5000, 1024, 128, "kaiming" if not cos else "unit", seed=13, dtype=dtype)
        xd = x.to(dev)
        if cos:
            e = O.l2norm(e)
            xd = L.l2norm_rows(xd)
        ed = e.to(dev)
        r = L.assign(xd, L.pack_codebook(ed), ed, cosine=cos, skip_l2norm=cos)
        assert r.get("n_exact") is not None


@pytest.mark.parametrize("N,C,D,dtype,cos", [(3000, 1024, 256, torch.bfloat16, False), (2000, 300, 512, torch.float32, True),
                                             (1000, 64, 64, torch.float32, False), (777, 2048, 512, torch.bfloat16, True)])
def test_score_indices_equals_exact_kernels_winner_score(dev, N, C, D, dtype, cos):
    """vqhip_score_indices: the reference-arithmetic score of a given code == what the exact kernel reports for its winner
    (needed by the codebook-sharded merge after a screened search, which certifies indices but produces no scores)."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(N, C, D, dtype, unit=True, seed=21)
    xd, ed = x.to(dev), e.to(dev)
    if cos:
        ed = torch.nn.functional.normalize(ed, dim=-1).contiguous()
        xd = L.l2norm_rows(xd)
    packed = L.pack_codebook(ed)
    r0 = L.assign(xd, packed, ed, cosine=cos, skip_l2norm=cos, want_q=False, want_best=True)     # exact kernel (want_best disables the screen)
    r1 = L.assign(xd, packed, ed, cosine=cos, skip_l2norm=cos, want_q=False)                     # screened
    assert r1.get("n_exact") is not None and torch.equal(r0["idx"], r1["idx"])
    s = L.score_indices(xd, packed, ed, r1["idx"], cosine=cos)
    assert torch.equal(s, r0["best"])


@pytest.mark.parametrize("C,D,frac", [(1024, 256, 0.1), (37, 8, 0.5), (5000, 32, 0.01), (64, 128, 0.0), (300, 100, 1.0)])
def test_expire_scatter_matches_masked_assignment(dev, C, D, frac):
    """vqhip_expire_scatter == the reference's `embed[mask] = sampled; cluster_size[mask] = reset; embed_avg[mask] = sampled * reset`
    (vqp.py:559-562) with sampled[j] going to the j-th expired code in ascending order."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(C + D)
    cs = torch.rand(C, generator=g) * 4 + 2.0
    dead = torch.rand(C, generator=g) < frac
    cs[dead] = torch.rand(int(dead.sum()), generator=g) * 1.9          # below the threshold 2
    e, ea, cand = torch.randn(C, D, generator=g), torch.randn(C, D, generator=g), torch.randn(C, D, generator=g)
    cs_d, e_d, ea_d = cs.to(dev), e.to(dev), ea.to(dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    L.expire_scatter(cs_d, ea_d, e_d, cand.to(dev), 2.0, 2.0, n)
    k = int(dead.sum())
    e[dead] = cand[:k]; ea[dead] = cand[:k] * 2.0; cs[dead] = 2.0
    assert int(n.item()) == k
    assert torch.equal(e_d.cpu(), e) and torch.equal(ea_d.cpu(), ea) and torch.equal(cs_d.cpu(), cs)


@pytest.mark.parametrize("C,D,cos", [(512, 256, False), (100, 40, True), (4096, 128, False)])
def test_kmeans_update_kernel(dev, C, D, cos):
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    means, esum = torch.randn(C, D, generator=g), torch.randn(C, D, generator=g) * 5
    cnt = torch.randint(0, 6, (C,), generator=g).float()
    md = means.to(dev)
    L.kmeans_update(md, esum.to(dev), cnt.to(dev), cosine=cos)
    nm = esum / cnt.clamp(min=1)[:, None]
    if cos:
        nm = torch.nn.functional.normalize(nm, dim=-1, eps=1e-6)
    want = torch.where((cnt == 0)[:, None], means, nm)
    if cos:
        assert (md.cpu() - want).abs().max().item() <= 1e-6
    else:
        assert torch.equal(md.cpu(), want)


@pytest.mark.parametrize("N,C,D,K,dtype,cos", [(1000, 64, 32, 3, torch.float32, False), (777, 1000, 256, 8, torch.float32, False),
                                               (2048, 512, 128, 4, torch.bfloat16, False), (500, 300, 64, 2, torch.float32, True),
                                               (300, 4096, 512, 5, torch.float32, True), (100, 5, 32, 5, torch.float32, False)])
def test_fused_topk_equals_topk_of_the_dense_scores(dev, N, C, D, K, dtype, cos):
    """vqhip_topk (no N x C tensor) == topk of vqhip_scores (the reference's `dist`, bit-exact vs the oracle), values and order;
    ties are ordered by ascending code (checked with duplicated codes)."""
    from vector_quantize_pytorch_amd import _lib as L
    x, e = _mk(N, C, D, dtype, unit=True, seed=31)
    if C >= 8:
        e[C // 2] = e[1]                                    # an exact duplicate: equal scores, the lower code must come first
    xd, ed = x.to(dev), e.to(dev)
    if cos:
        ed = torch.nn.functional.normalize(ed, dim=-1).contiguous()
    packed = L.pack_codebook(ed)
    idx, val = L.topk(xd, packed, C, K, cosine=cos, want_values=True)
    dist, _, _ = L.scores(xd, packed, ed, cosine=cos)
    # reference order: value descending, then code ascending (stable sort of the dense row)
    order = torch.sort(dist, dim=-1, descending=True, stable=True)
    assert torch.equal(val, order.values[:, :K])
    assert torch.equal(idx, order.indices[:, :K])


@pytest.mark.parametrize("N,C,D,cos", [(300, 64, 32, False), (77, 130, 100, False), (64, 7, 6, False), (200, 64, 64, True), (1, 1, 4, False)])
def test_assign_rowwise_matches_pairwise_distance(dev, N, C, D, cos):
    """vqhip_assign_rowwise (one codebook per row, QINCo): argmin of F.pairwise_distance(x[:, None], codes) -- the reference's
    expression at vqp.py:738 -- or argmax of the cosine einsum (:735); every disagreement must be a near-tie of that expression."""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, D, generator=g).to(dev)
    codes = torch.randn(N, C, D, generator=g).to(dev)
    if cos:
        x, codes = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(codes, dim=-1)
        score = torch.einsum("nd,ncd->nc", x, codes)
    else:
        score = -torch.nn.functional.pairwise_distance(x[:, None, :], codes)
    idx = L.assign_rowwise(x, codes, cosine=cos)
    want = score.argmax(-1)
    bad = (idx != want).nonzero().flatten()
    gap = (score.gather(1, want[:, None]) - score.gather(1, idx[:, None])).abs().flatten()[bad]
    assert bad.numel() <= max(1, N // 100) and (gap <= 1e-5 * score.abs().max()).all()
    # ties: identical codes -> the lowest index
    codes[:, C - 1] = codes[:, 0]
    x2 = codes[:, 0].clone()
    assert (L.assign_rowwise(x2, codes, cosine=cos) == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("negate", [False, True])
def test_pack_unpack_best_equals_the_key_algebra(dev, negate):
    """vqhip_pack_best / vqhip_unpack_best (the K11 helper of the codebook-sharded argmin) against the torch restatement of the key
    (parallel.pack_score_index): bit-identical keys, key order == (score, then lower index), ownership mask and local index."""
    from vector_quantize_pytorch_amd import _lib as L
    from vector_quantize_pytorch_amd.parallel import pack_score_index, unpack_score_index
    g = torch.Generator().manual_seed(5)
    N, lo, hi, off = 100_003, 8192, 16384, 8192
    s = torch.randn(N, generator=g)
    s[:7] = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), 1e-45, -1e-45, 3.0])
    s[100:200] = s[300:400]                                   # equal scores: the lower index must win
    idx = torch.randint(0, hi - lo, (N,), generator=g)
    key = L.pack_best(s.to(dev), idx.to(dev), off, negate=negate)
    ref = pack_score_index(-s if negate else s, idx + off)
    assert torch.equal(key.cpu(), ref)
    other = pack_score_index(-s.roll(1) if negate else s.roll(1), idx.roll(1) + 3 * off)     # "another shard's" keys
    red = torch.maximum(ref, other).to(dev)
    gidx, local, best = L.unpack_best(red, lo, hi, negate=negate, want_best=True)
    s2, i2 = unpack_score_index(red.cpu())
    assert torch.equal(gidx.cpu(), i2)
    assert torch.equal(best.cpu().view(torch.int32), (-s2 if negate else s2).view(torch.int32))
    mine = (i2 >= lo) & (i2 < hi)
    assert torch.equal(local.cpu(), torch.where(mine, i2 - lo, torch.full_like(i2, -1)))


def test_routed_residuals_are_deterministic_beside_concurrent_searches(dev):
    """Round 5 finding (csrc/Makefile, tools/route_concurrency_check.py): built with the SLP vectoriser, the routing kernels' packed
    fp32 operations gave a wrong first element in lanes 48..63 of a wave about once per 270 launches WHEN another kernel's MFMA waves
    ran on the same SIMD (never alone on the chip) -- one near-tie row of a gradient step continuing with the other code.  Here:
    vqhip_route_residual (rotation trick) on three streams, each followed by a screened search as in the stages of a routed residual
    loop (rvq.py:524 with vqp.py:1225-1233), 100 rounds: every output bit-identical to the launch that ran alone."""
    import ctypes
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator(device=dev).manual_seed(0)
    D, C, N, rpc = 64, 256, 3 * 65536 + 300, 65792
    x = torch.randn(N, D, device=dev, generator=g) * 3
    e = torch.randn(C, D, device=dev, generator=g)
    pk = L.pack_codebook(e)
    idx0 = L.assign(x, pk, e, want_q=False)["idx"].clone()
    lib = L.lib()
    nws = lib.vqhip_screen_workspace_bytes(rpc)

    def route(out, r0, n):
        L._check(lib.vqhip_route_residual(ctypes.c_void_p(x.data_ptr() + r0 * D * 4), 0, n, D, D, L._ptr(e), ctypes.c_void_p(idx0.data_ptr() + r0 * 8),
                                          1, 2, ctypes.c_void_p(out.data_ptr() + r0 * D * 4), D, L._stream()), "route")

    def search(out, idx1, ws, r0, n):
        ch = L._Chain(idx_stride=1, prev_idx=None, prev_idx_stride=1, prev_embed=None, x_out=None, ldxo=D, route_mode=0, header_zeroed=0)
        L._check(lib.vqhip_assign_screened_chain(ctypes.c_void_p(out.data_ptr() + r0 * D * 4), 0, n, D, D, L._ptr(pk), L._ptr(e), C, 0,
                                                 ctypes.c_void_p(idx1.data_ptr() + r0 * 8), None, L._ptr(ws), nws, ctypes.byref(ch), L._stream()), "search")

    ref = torch.empty_like(x)
    route(ref, 0, N)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    wss = [torch.zeros((nws + 15) // 16 * 4, dtype=torch.int32, device=dev) for _ in range(3)]
    outs = [torch.empty_like(x) for _ in range(4)]
    idx1 = torch.empty(N, dtype=torch.int64, device=dev)
    wrong = 0
    for _ in range(100):
        for o in outs:
            o.fill_(float("nan"))
        torch.cuda.synchronize()
        for o in outs:
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    route(o, k * rpc, min(rpc, N - k * rpc))
                    search(o, idx1, wss[k], k * rpc, min(rpc, N - k * rpc))
        torch.cuda.synchronize()
        wrong += sum(int(not torch.equal(o, ref)) for o in outs)
    assert wrong == 0, f"{wrong} of 400 routed-residual launches differ from the launch that ran alone"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deterministic_kernels_give_the_same_bits_beside_a_concurrent_mfma_load(dev, dtype):
    """The canary behind round 5's finding (csrc/Makefile): every kernel of the path whose output is a pure function of its inputs
    is run alone, then 40 times on one stream while another stream keeps the chip busy with screened searches (MFMA waves on every
    SIMD): the outputs must be bit-identical.  Covered: routing forward / backward (rotation trick, by tensor and by gathered
    index), the routed residual, the all-stages routing kernel forward / backward, l2norm forward / backward, decode, the exact
    search (indices + winning distances), the screened search (indices), the codebook pack (through a search on it), the EMA fold.  (The statistics' sums go
    through fp32 atomics in any order and are excluded; their integer counts are covered.)"""
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator(device=dev).manual_seed(7)
    N, D, C, Q = 40000, 256, 512, 3
    x = (torch.randn(N, D, device=dev, generator=g) * 2).to(dtype)
    go = torch.randn(N, D, device=dev, generator=g).to(dtype)
    e = torch.randn(C, D, device=dev, generator=g)
    ecodes = e.to(dtype)
    idx = torch.randint(0, C, (N, Q), device=dev, generator=g)
    idx0 = idx[:, 0].contiguous()
    q = ecodes[idx0]
    coef = torch.tensor(0.3, device=dev)
    coefs = torch.tensor([0.3, 0.2, 0.1], device=dev)
    cs = torch.rand(C, device=dev, generator=g) * 5
    cnt = torch.randint(0, 9, (C,), device=dev, generator=g).float()
    esum = torch.randn(C, D, device=dev, generator=g)
    pk = L.pack_codebook(e)

    def ops():
        out = {}
        out["route_fwd"] = L.route_fwd(x, q, 2)
        out["route_bwd"] = L.route_bwd(x, q, go, coef, None, 2)
        out["route_fwd_gather"] = L.route_fwd_gather(x, ecodes, idx0, 2)
        out["route_bwd_gather"] = L.route_bwd_gather(x, ecodes, idx0, go, coef, None, 2)
        out["rvq_route_fwd"] = L.rvq_route(x, e, idx, Q, 2, resid_routed=True)
        out["rvq_route_bwd"] = L.rvq_route(x, e, idx, Q, 2, g_out=go, loss_coef=coefs, backward=True, resid_routed=True)
        out["l2norm"] = L.l2norm_rows(x)
        out["l2norm_bwd"] = L.l2norm_rows_bwd(x, go)
        out["decode"] = L.decode_sum(idx, e, out_dtype=dtype)
        r = L.assign(x[:8192], pk, e, want_q=True, want_best=True)              # want_best: the exact kernel
        out["exact_idx"], out["exact_best"], out["exact_q"] = r["idx"], r["best"], r["q"]
        out["screen_idx"] = L.assign(x, pk, e, want_q=False)["idx"]
        pk2 = L.pack_codebook(e)                                                 # (the packed buffer has uninitialised padding: judged by
        out["screen_idx_fresh_pack"] = L.assign(x[:65536], pk2, e, want_q=False)["idx"]   #  what a search on it returns)
        a = [t.clone() for t in (cs, esum, e)]
        L.ema_finalize(a[0], a[1], a[2], cnt, esum * 0.5, decay=0.8, eps=1e-5)
        out["fold_cs"], out["fold_ea"], out["fold_e"] = a
        out["count"] = L.ema_accumulate(x, idx0, C)[0]
        return out

    ref = ops()
    torch.cuda.synchronize()
    load = torch.cuda.Stream()
    xl = torch.randn(1 << 18, 256, device=dev, generator=g)
    stop_after = 40
    bad = {}
    for it in range(stop_after):
        with torch.cuda.stream(load):                      # ~4 x 0.2 ms of MFMA-dense search per round on the other stream
            for _ in range(4):
                L.assign(xl, pk, e, want_q=False)
        got = ops()
        torch.cuda.synchronize()
        for k, v in got.items():
            if not torch.equal(v, ref[k]):
                bad[k] = bad.get(k, 0) + 1
    assert not bad, f"outputs that differ from the solo run (rounds out of {stop_after}): {bad}"



def test_train_step_fold_inside_the_segmented_sum_matches_the_fold_kernel(dev):
    """VQHIP_STEP_FOLD=1 (round 6, measured no faster, not the default): the wave that adds the last chunk of a code's segmented sum folds
    embed_avg / embed for that code, the last workgroup reduces the loss -- device-scope tickets instead of a launch boundary.  Against
    the default vq_step_fold_kernel over many steps on one reused workspace (tools/fold_stress.py --quick: two child processes, the
    switch is read once per process): identical indices and cluster sizes, embed / embed_avg / loss equal up to the fp32 atomics' order."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fold_stress.py"), "--quick"], cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "fold stress OK" in p.stdout, p.stdout[-3000:]
