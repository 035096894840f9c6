"""Adversarial parity tests of the SCREENED search (csrc/vq_screen.hip).

The screened path certifies an index from an fp16-MFMA score and a model of that score's error.  These tests put the
certificate under load instead of trusting the model:

* a fuzz over magnitudes 1e-6 .. 1e4, DC offsets, heavy tails, mixed-norm codebooks, 2 <= C <= 65536,
  D in {32, 64, 128, 256, 512}, both dtypes and both metrics, > 10^7 rows in total, every row compared bit for bit with the
  exact fp32-MFMA kernel (which the other test files pin to oracle/vq_oracle.c);
* the full BASELINE cfg-2 batch (2^20 rows) at step 1 and over five EMA steps with VQHIP_SCREEN_VERIFY=1;
* a direct measurement of the MFMA unit's accumulation error (fp16 single-pass kernel for bf16 rows, bf16 three-product
  kernel for fp32 rows) on cancellation-heavy vectors against the modelled
  "one truncating rounding (2u) per added term" -- the certificate's one hardware assumption.
"""
import math

import pytest
import torch
import os

# VQHIP_FUZZ_SEED=<int> shifts the seeds of the adversarial cases: other draws of the same distributions (default: the fixed set)
_SEED_SHIFT = 100000 * int(os.environ.get("VQHIP_FUZZ_SEED", "0"))

pytestmark = pytest.mark.gpu

U = 2.0 ** -24


def _rows(kind, N, D, scale, offset, gen, dev):
    x = torch.randn(N, D, generator=gen, device=dev)
    if kind == "heavy":                       # heavy tails: a few coordinates dominate ||x||
        t = torch.randn(N, D, generator=gen, device=dev) / torch.randn(N, D, generator=gen, device=dev).abs().clamp(min=1e-3)
        x = t.clamp(-1e3, 1e3)
    elif kind == "sparse":                    # most coordinates zero
        x = x * (torch.rand(N, D, generator=gen, device=dev) < 0.1)
    elif kind == "rowscale":                  # row norms spread over 6 decades inside one batch
        x = x * torch.exp(torch.randn(N, 1, generator=gen, device=dev) * 3.0)
    return x * scale + offset * scale


def _codes(kind, C, D, scale, gen, dev, x=None):
    e = torch.randn(C, D, generator=gen, device=dev) * scale
    if kind == "kaiming":                     # the reference's default init: tiny codes
        e = (torch.rand(C, D, generator=gen, device=dev) * 2 - 1) * (6.0 / D) ** 0.5 * scale
    elif kind == "onebig":                    # one code 1e3 x the rest: the bound uses max ||c|| for every code
        e[C // 3] *= 1e3
    elif kind == "lognorm":                   # code norms spread over several decades
        e = e * torch.exp(torch.randn(C, 1, generator=gen, device=dev) * 2.0)
    elif kind == "rows":                      # codes drawn from the data (k-means init / dead-code replacement)
        pick = torch.randint(0, x.shape[0], (C,), generator=gen, device=dev)
        e = x[pick].float().clone()
    elif kind == "cluster":                   # tight clusters: many codes within rounding distance of each other
        centers = torch.randn(max(C // 16, 1), D, generator=gen, device=dev) * scale
        e = centers[torch.arange(C, device=dev) % centers.shape[0]] + torch.randn(C, D, generator=gen, device=dev) * scale * 1e-4
    return e.contiguous()


# (rows kind, codes kind, N, C, D, scale, offset)  -- every case runs in bf16 and fp32, Euclidean; a subset in cosine
_CASES = [
    ("randn", "kaiming", 1 << 18, 1024, 256, 1.0, 0.0),
    ("randn", "kaiming", 1 << 18, 1024, 256, 1e-6, 0.0),
    ("randn", "kaiming", 1 << 18, 1024, 256, 1e4, 0.0),
    ("randn", "randn", 1 << 18, 1024, 256, 1.0, 10.0),        # DC offset: X Y >> score gaps
    ("randn", "randn", 1 << 18, 1024, 256, 1.0, 100.0),
    ("randn", "rows", 1 << 18, 1024, 256, 30.0, 100.0),
    ("heavy", "randn", 1 << 18, 1024, 256, 1.0, 0.0),
    ("heavy", "rows", 1 << 18, 512, 128, 1e-3, 0.0),
    ("sparse", "randn", 1 << 18, 1000, 128, 1.0, 0.0),
    ("rowscale", "randn", 1 << 18, 1024, 256, 1.0, 0.0),
    ("rowscale", "lognorm", 1 << 18, 1024, 64, 1.0, 0.0),
    ("randn", "onebig", 1 << 18, 1024, 256, 1.0, 0.0),
    ("randn", "onebig", 1 << 18, 100, 32, 1e-3, 10.0),
    ("randn", "lognorm", 1 << 18, 4096, 128, 1.0, 0.0),
    ("randn", "cluster", 1 << 18, 1024, 256, 1.0, 0.0),
    ("randn", "cluster", 1 << 17, 4096, 64, 1e3, 0.0),
    ("randn", "randn", 1 << 18, 2, 256, 1.0, 0.0),            # smallest codebook
    ("randn", "randn", 1 << 18, 33, 32, 1.0, 0.0),
    ("randn", "randn", 1 << 17, 8192, 32, 1.0, 0.0),
    ("randn", "randn", 1 << 15, 65536, 128, 0.3, 0.0),        # 2048 tiles
    ("randn", "kaiming", 1 << 15, 65536, 256, 1.0, 1.0),
    ("heavy", "lognorm", 1 << 18, 1024, 256, 1e-6, 0.0),
    ("rowscale", "rows", 1 << 18, 1024, 64, 1e4, 0.0),
    ("randn", "randn", 1 << 16, 8192, 512, 1.0, 0.0),         # D = 512: one row block per wave (cfg 4's dimension)
    ("randn", "kaiming", 1 << 16, 1024, 512, 1.0, 0.0),
    ("heavy", "rows", 1 << 15, 4096, 512, 1e-3, 0.0),
    ("rowscale", "cluster", 1 << 16, 1000, 512, 1.0, 10.0),
    ("randn", "randn", 1 << 16, 1024, 256, 1e-30, 0.0),       # ||x||^2 underflows: nothing can be certified, everything must still agree
    ("randn", "randn", 1 << 16, 1024, 256, 1e18, 0.0),        # ||x||^2 ~ 1e38: close to fp32's limit
    ("randn", "kaiming", 1 << 16, 512, 128, 1e20, 0.0),       # ||x||^2 overflows to inf
    ("rowscale", "randn", 1 << 16, 1024, 64, 1e-12, 0.0),
]


def _run_both(L, monkeypatch, xd, ed, cosine):
    packed = L.pack_codebook(ed)
    monkeypatch.setenv("VQHIP_SCREEN", "1")
    r1 = L.assign(xd, packed, ed, cosine=cosine, skip_l2norm=cosine, want_q=True, want_sqerr=True)
    assert r1.get("n_exact") is not None, "the case must take the screened path"
    monkeypatch.setenv("VQHIP_SCREEN", "0")
    r0 = L.assign(xd, packed, ed, cosine=cosine, skip_l2norm=cosine, want_q=True, want_sqerr=True)
    assert r0.get("n_exact") is None
    return r1, r0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_screened_equals_exact_kernel_adversarial_fuzz(dev, monkeypatch, dtype):
    """>= 10^7 rows through the screened path, compared bit for bit with the exact kernel on the same inputs."""
    from vector_quantize_pytorch_amd import _lib as L
    total = flagged = paired = 0
    for ci, (rk, ck, N, C, D, scale, offset) in enumerate(_CASES):
        gen = torch.Generator(device=dev).manual_seed(1000 + ci + _SEED_SHIFT)
        x = _rows(rk, N, D, scale, offset, gen, dev).to(dtype).contiguous()
        e = _codes(ck, C, D, scale, gen, dev, x=x)
        r1, r0 = _run_both(L, monkeypatch, x, e, cosine=False)
        bad = int((r1["idx"] != r0["idx"]).sum())
        assert bad == 0, f"case {ci} {rk}/{ck} N={N} C={C} D={D} scale={scale} offset={offset} {dtype}: {bad} index mismatches"
        assert torch.equal(r1["q"], r0["q"]), f"case {ci}: q differs"
        s1 = float(r1["sqerr_partials"][: r1["nblk"]].sum())
        s0 = float(r0["sqerr_partials"][: r0["nblk"]].sum())
        assert s1 == s0 or abs(s1 - s0) <= 1e-6 * max(abs(s0), 1e-300), f"case {ci}: squared error {s1} vs {s0}"   # (fp32 per-row partial sums in different orders; inf == inf for overflowing rows)
        total += N
        flagged += int(r1["n_exact"])
        paired += int(r1["n_pair"])
    assert total >= 5_000_000
    print(f"[screen fuzz {dtype}] rows {total}, full exact sweep {flagged} ({100.0 * flagged / total:.2f} %), "
          f"decided between two candidates {paired} ({100.0 * paired / total:.2f} %)")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_screened_cosine_equals_exact_kernel_adversarial_fuzz(dev, monkeypatch, dtype):
    from vector_quantize_pytorch_amd import _lib as L
    total = 0
    for ci, (rk, ck, N, C, D, scale, offset) in enumerate(_CASES):
        if ci % 2 or C < 2:
            continue
        gen = torch.Generator(device=dev).manual_seed(2000 + ci + _SEED_SHIFT)
        x = _rows(rk, N, D, scale, offset, gen, dev).to(dtype).contiguous()
        e = _codes(ck, C, D, scale, gen, dev, x=x)
        e = torch.nn.functional.normalize(e, p=2, dim=-1, eps=1e-6).contiguous()    # the cosine codebook is kept unit-norm (vqp.py:388)
        xn = L.l2norm_rows(x)
        r1, r0 = _run_both(L, monkeypatch, xn, e, cosine=True)
        bad = int((r1["idx"] != r0["idx"]).sum())
        assert bad == 0, f"cosine case {ci} {rk}/{ck} N={N} C={C} D={D} {dtype}: {bad} index mismatches"
        assert torch.equal(r1["q"], r0["q"])
        total += N
    assert total >= 1_000_000


def test_cfg2_full_batch_screen_verified_over_ema_steps(dev, monkeypatch):
    """BASELINE cfg 2 at full size (2^20 bf16 rows, C = 1024, D = 256): step 1 on the reference's default init and five more
    EMA steps, each screened search re-done by the exact kernel (VQHIP_SCREEN_VERIFY raises on any disagreement)."""
    from vector_quantize_pytorch_amd import VectorQuantize
    monkeypatch.setenv("VQHIP_SCREEN", "1")
    monkeypatch.setenv("VQHIP_SCREEN_VERIFY", "1")
    torch.manual_seed(0)
    vq = VectorQuantize(dim=256, codebook_size=1024).to(dev).train()
    gen = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for step in range(6):
            x = torch.randn(64, 16384, 256, generator=gen, device=dev).bfloat16()     # a fresh batch every step
            q, idx, loss = vq(x)
            assert idx.shape == (64, 16384) and bool(torch.isfinite(loss))
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------------------------------
# The certificate's hardware assumption, measured.  vq_screen.hip models the bf16 MFMA's internal accumulation as "one
# TRUNCATING rounding (relative error <= 2u, u = 2^-24) per added term".  Below, rows and codes are built from bf16-exact
# values so that every product is exact and the only error in the screen's score is the MFMA accumulation (+ the 4 index
# bits the kernel stores in the mantissa).  t_exact is formed in float64 (exact here: < 2^53 dynamic range is enforced).
# Patterns: one huge product followed by many tiny ones and a cancelling huge one (inside one 16-term MFMA and across
# MFMAs), random exponents over 2^+-12, alternating signs that cancel to ~0.
# ------------------------------------------------------------------------------------------------------------------------
def _bf16_exact(t):
    return t.bfloat16().float()


def _added_start_value(x, e, D):
    """Upper bound [N, C] of what the screening kernels ADD to a code's start value -||c||^2 / 2 before the MFMAs accumulate on top of
    it (round 6, csrc/vq_screen.hip header: Arow ||c|| + kb ||c||^2 -- every tracked score is an upper bound of the code's true score).
    The accumulation model below has to count it among the terms the hardware rounds against.  Arow from the largest row norm of the
    row's wave (64 rows: one operand scale per wave; the families here are per wave), rho <= 2^-12, bf16 / fp16-exact rows (no drop term)."""
    xn = x.double().norm(dim=-1)
    xw = xn.view(-1, 64).max(dim=1, keepdim=True).values.expand(-1, 64).reshape(-1)
    y = e.double().norm(dim=-1)
    arow = (xn * (U * (10.0 + D + 2.002 * (D + 1)) + 2.5e-4) + U * D ** 0.5 * xw / 4096.0) * 1.001
    kb = U * (5.0 + 1.001 * (D + 1) + 0.51) * 1.001
    return arow[:, None] * y[None, :] * 1.001 + kb * (y ** 2)[None, :] * 1.002


def _adversarial_pairs(N, D, gen, big=20, spread=12):
    """rows x [N, D] and ONE code c [D], bf16-exact, products with heavy cancellation; several pattern families by row."""
    x = torch.zeros(N, D)
    c = _bf16_exact((torch.rand(D, generator=gen) + 0.5) * torch.where(torch.rand(D, generator=gen) < 0.5, -1.0, 1.0))
    c[0] = c[3] = c[12] = c[D - 1] = 1.0                                     # the huge products below are exactly +-2^k
    fam = (torch.arange(N) // 64) % 4                                      # one family per wave (64 rows): the fp16 kernel scales x per wave
    mant = _bf16_exact(torch.rand(N, D, generator=gen) + 1.0)                # 8-bit mantissas in [1, 2)
    sign = torch.where(torch.rand(N, D, generator=gen) < 0.5, -1.0, 1.0)
    # family 0: random exponents in [-12, 12]
    e0 = torch.randint(-spread, spread + 1, (N, D), generator=gen).float()
    x0 = sign * mant * torch.exp2(e0)
    # family 1: product +2^20 first, tiny same-sign terms, product -2^20 LAST (different MFMAs)
    x1 = _bf16_exact(mant * torch.sign(c)[None, :])                             # all products positive, ~1
    x1[:, 0] = _bf16_exact(torch.full((N,), 2.0 ** big) / c[0])
    x1[:, D - 1] = _bf16_exact(-torch.full((N,), 2.0 ** big) / c[D - 1])
    # family 2: the cancelling pair sits inside ONE 16-term group (k = 3 and k = 12), tiny terms everywhere else
    x2 = _bf16_exact(mant * torch.sign(c)[None, :] * 2.0 ** -3)
    x2[:, 3] = _bf16_exact(torch.full((N,), 2.0 ** (big - 2)) / c[3])
    x2[:, 12] = _bf16_exact(-torch.full((N,), 2.0 ** (big - 2)) / c[12])
    # family 3: alternating signs of equal magnitude per adjacent pair, magnitudes growing with k
    grow = torch.exp2((torch.arange(D) // 2).float() * (float(big) / (D // 2)))[None, :]
    alt = torch.where(torch.arange(D) % 2 == 0, 1.0, -1.0)[None, :]
    x3 = _bf16_exact(mant * alt * grow * torch.sign(c)[None, :])
    for f, xf in enumerate((x0, x1, x2, x3)):
        x[fam == f] = _bf16_exact(xf)[fam == f]
    return x, c


@pytest.mark.parametrize("D", [32, 64, 128, 256])
@pytest.mark.parametrize("wide", [False, True])
def test_mfma_accumulation_error_within_model(dev, D, wide):
    """bf16 rows -> vq_screen16_kernel (one fp16 MFMA pass).  |screen score - exact score of the SAME fp16 operands| must stay
    below 2u * (number of added terms) * sum|terms| (+ the 4 index bits): the model the certificate charges for the MFMA.
    `wide`: the code has 16 significant bits, so its fp16 image differs from it (the operands the MFMA sees are what
    t_exact is built from -- this isolates the accumulation from the rounding of the codebook)."""
    from oracle import vq_oracle as O
    from vector_quantize_pytorch_amd import _lib as L
    gen = torch.Generator().manual_seed(77 + D + int(wide))
    N = 4096
    x, c = _adversarial_pairs(N, D, gen)
    if wide:
        c = c + _bf16_exact(c * 2.0 ** -9 * (torch.rand(D, generator=gen) * 0.5 + 0.4))
    # second code: far away from every row so that (best, second) = (code 0, code 1) or the reverse, unambiguously
    e = torch.stack([c, torch.zeros(D)])
    e[1, 0] = 2.0 ** -20
    ch = e.half().float()                     # what vq_pack16_kernel stores (its power-of-two scaling commutes with the rounding)
    xd, ed = x.bfloat16().to(dev), e.to(dev)
    assert torch.equal(xd.float().cpu(), x)
    L.screen_debug = True
    try:
        r = L.assign(xd, L.pack_codebook(ed), ed, want_q=False)
    finally:
        L.screen_debug = False
    dbg = r["screen_debug"].double().cpu()
    # (round 6: the kernels rank the codes by an upper bound of their score -- score + the code's own error allowance -- and report the two
    #  best of THAT ranking with the allowance taken off again: with two codes these are the two scores, possibly the other way round)
    dbg[:, :2] = dbg[:, :2].sort(dim=1, descending=True).values
    nh = (-0.5 * O.c_row_sumsq(e)).double()                                       # the accumulator's initial value (fp32)
    prods = x.double()[:, None, :] * ch.double()[None]                            # [N, 2, D], exact
    t_exact = prods.sum(-1) + nh[None, :]
    add = _added_start_value(x, e, D)
    a_sum = prods.abs().sum(-1) + nh.abs()[None, :] + add
    order = t_exact.argsort(dim=1, descending=True)
    t_sorted = t_exact.gather(1, order)
    a_sorted = a_sum.gather(1, order)
    add_sorted = add.gather(1, order)
    n_terms = D + 1
    worst_model = worst_permfma = 0.0
    for k in range(2):
        got = dbg[:, k]
        err = (got - t_sorted[:, k]).abs()
        idx_bits = 16.0 * (got.abs() + add_sorted[:, k]) * 2.0 ** -23              # 4 mantissa bits (of score + allowance) overwritten by the code number
        model = 2.0 * U * n_terms * a_sorted[:, k]
        ok = err <= model + idx_bits
        assert bool(ok.all()), (f"D={D} wide={wide}: MFMA accumulation error exceeds the modelled 2u/term: "
                                f"worst ratio {(err / (model + idx_bits)).max():.3f}")
        worst_model = max(worst_model, float(((err - idx_bits).clamp(min=0) / model).max()))
        per_mfma = 2.0 * U * (D // 16 + 1) * a_sorted[:, k]                        # one rounding per MFMA instead of per term
        worst_permfma = max(worst_permfma, float(((err - idx_bits).clamp(min=0) / per_mfma).max()))
    print(f"[mfma accumulation D={D} wide={wide}] worst error = {worst_model:.4f} of the per-term model, "
          f"{worst_permfma:.4f} of a one-rounding-per-MFMA model")
    assert worst_model <= 0.5, "the per-term model should keep a 2x margin over anything observed"


def test_mfma_accumulation_error_f32_rows(dev):
    """same measurement for fp32 rows (vq_screen16_kernel<.., XF32>: x_h = fp16_rne(x'), one product per k-step).  The rows carry
    at most 11 significant bits and a dynamic range below 2^13 per wave, so x_h = x' exactly and the only error left in the score
    is the MFMA accumulation."""
    from oracle import vq_oracle as O
    from vector_quantize_pytorch_amd import _lib as L
    D, N = 256, 4096
    gen = torch.Generator().manual_seed(5)
    x, c = _adversarial_pairs(N, D, gen, big=8, spread=4)
    assert torch.equal(x.half().float(), x)
    e = torch.stack([c, torch.zeros(D)])
    e[1, 0] = 2.0 ** -20
    xd, ed = x.to(dev), e.to(dev)
    L.screen_debug = True
    try:
        r = L.assign(xd, L.pack_codebook(ed), ed, want_q=False)
    finally:
        L.screen_debug = False
    assert r.get("n_exact") is not None
    dbg = r["screen_debug"].double().cpu()
    dbg[:, :2] = dbg[:, :2].sort(dim=1, descending=True).values                  # (ranked by the upper bound, see above)
    nh = (-0.5 * O.c_row_sumsq(e)).double()
    prods = x.double()[:, None, :] * e.double()[None]
    t_exact = prods.sum(-1) + nh[None, :]
    add = _added_start_value(x, e, D)
    a_sum = prods.abs().sum(-1) + nh.abs()[None, :] + add
    order = t_exact.argsort(dim=1, descending=True)
    t_sorted, a_sorted, add_sorted = t_exact.gather(1, order), a_sum.gather(1, order), add.gather(1, order)
    n_terms = D + 1
    for k in range(2):
        got = dbg[:, k]
        err = (got - t_sorted[:, k]).abs()
        idx_bits = 16.0 * (got.abs() + add_sorted[:, k]) * 2.0 ** -23
        model = 2.0 * U * n_terms * a_sorted[:, k]
        assert bool((err <= model + idx_bits).all()), f"f32-row screen: worst ratio {(err / (model + idx_bits)).max():.3f}"


@pytest.mark.parametrize("persist", ["1", "0"])
def test_both_screening_kernels_equal_exact_kernel(persist):
    """The persistent screening kernel (csrc/vq_screen_c.hip: cyclic tile stream, per-workgroup list segments +
    vq_compact_lists_kernel; the default for bf16 rows at D = 256 from 131 072 rows on) and the 4-wave kernel it replaced there
    (VQHIP_SCREEN_PERSIST=0) against the exact fp32-MFMA kernel, bit for bit (indices, q rows), on the cases of
    tools/persist_check.py: ragged N, padded codebooks, twin codes, wild row norms, both metrics.  A subprocess: the switch is read
    once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQHIP_SCREEN_PERSIST=persist)
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "persist_check.py")], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "ALL OK" in p.stdout, p.stdout[-3000:]
    assert "open 0 pair 0" not in p.stdout.splitlines()[0], "the first case must exercise the lists"
