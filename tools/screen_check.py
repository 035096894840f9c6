"""Dev tool: screened (bf16-MFMA + exact refine) vs exact assignment on the GPU -- bitwise agreement, accuracy of the
screening scores against their certified bound, fraction of rows taking the exact pass, and timing.
    python tools/screen_check.py [--quick]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L  # noqa: E402


def run(x, embed, packed, screened, cosine=False, **kw):
    os.environ["VQHIP_SCREEN"] = "1" if screened else "0"
    if cosine and screened:      # what Codebook.quantize does: normalise once, then the screened search on unit-norm rows
        r = L.assign(L.l2norm_rows(x), packed, embed, cosine=True, skip_l2norm=True, want_q=True, want_sqerr=True, **kw)
    else:                        # exact kernel, normalisation fused
        r = L.assign(x, packed, embed, cosine=cosine, want_q=True, want_sqerr=True, **kw)
    os.environ["VQHIP_SCREEN"] = "1"
    return r


def codebooks(kind, C, D, x, gen):
    if kind == "kaiming":   # the reference's default init (vqp.py:28-31)
        e = torch.empty(C, D, device="cuda")
        torch.nn.init.kaiming_uniform_(e, generator=gen)
        return e
    if kind == "randn":
        return torch.randn(C, D, device="cuda", generator=gen)
    if kind == "rows":      # codes sampled from the data (k-means-like init)
        return x.reshape(-1, D)[torch.randperm(x.shape[0], device="cuda", generator=gen)[:C]].float().contiguous()
    if kind == "dups":      # duplicated codes: every row whose best code is duplicated must take the exact pass
        e = torch.randn(C, D, device="cuda", generator=gen)
        e[C // 2:] = e[: C - C // 2]
        return e
    if kind == "tiny":      # a collapsed codebook (codes ~ 1e-3): gaps near the rounding level
        return torch.randn(C, D, device="cuda", generator=gen) * 1e-3
    raise ValueError(kind)


def check(N, C, D, kind, gen, timing=False, dtype=torch.bfloat16, cosine=False):
    x = torch.randn(N, D, device="cuda", generator=gen).to(dtype)
    embed = codebooks(kind, C, D, x, gen)
    if cosine:
        embed = torch.nn.functional.normalize(embed, dim=-1)
    packed = L.pack_codebook(embed)
    r0 = run(x, embed, packed, False, cosine)
    L.screen_debug = True
    r1 = run(x, embed, packed, True, cosine)
    L.screen_debug = False
    torch.cuda.synchronize()
    assert "n_exact" in r1, "screened path not taken"
    nex = int(r1["n_exact"].item())
    same_idx = bool((r0["idx"] == r1["idx"]).all())
    nbad = int((r0["idx"] != r1["idx"]).sum())
    same_q = bool(torch.equal(r0["q"], r1["q"]))
    s0 = r0["sqerr_partials"][: r0["nblk"]].sum().item()
    s1 = r1["sqerr_partials"][: r1["nblk"]].sum().item()
    # accuracy of the screening scores on a sample: t = x.c - y2/2 in float64 (y2 = the fp32 value the kernels use)
    dbg = r1["screen_debug"]
    ns = min(N, 8192)
    sel = torch.randperm(N, device="cuda", generator=gen)[:ns]
    y2 = L.row_sumsq(embed).double()
    xs = (L.l2norm_rows(x) if cosine else x)[sel].double()
    t = xs @ embed.double().t() - (0.0 if cosine else 0.5) * y2[None, :]
    top = t.topk(2, dim=1).values
    err1 = (dbg[sel, 0].double() - top[:, 0]).abs()
    err2 = (dbg[sel, 1].double() - top[:, 1]).abs()
    thr = dbg[sel, 2].double()
    ratio = float((torch.maximum(err1, err2) / thr).max())
    flagged = dbg[:, 3].sum().item()
    line = (f"{'cos' if cosine else 'l2 '} {str(dtype)[6:]:8s} N={N} C={C} D={D} {kind:8s} idx_equal={same_idx} (bad {nbad}) q_equal={same_q} "
            f"sqerr rel diff={abs(s0 - s1) / max(abs(s0), 1e-30):.2e} exact_rows={nex} ({100.0 * nex / N:.3f}%) "
            f"flagged_dbg={int(flagged)} max|t_err|/thr={ratio:.4f}")
    print(line, flush=True)
    ok = same_idx and same_q and abs(s0 - s1) <= 1e-5 * abs(s0) and nex == int(flagged)
    if timing:
        for screened in (False, True):
            for _ in range(3):
                run(x, embed, packed, screened, cosine)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run(x, embed, packed, screened, cosine)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print(f"   {'screened' if screened else 'exact   '}: {dt * 1e3:.3f} ms/call  {N / dt:.3e} vec/s", flush=True)
    return ok


def main():
    quick = "--quick" in sys.argv
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234)
    ok = True
    cases = [(4096, 1024, 256, "kaiming"), (5000, 1000, 256, "randn"), (300, 37, 128, "randn"), (70000, 512, 64, "rows"),
             (65536, 1024, 256, "rows"), (65536, 1024, 256, "dups"), (65536, 1024, 256, "tiny"), (1000, 2, 64, "randn"),
             (33333, 4096, 128, "kaiming")]
    if "--d32" in sys.argv:      # low-dimensional codebooks: timing + agreement only
        for dtype in (torch.bfloat16, torch.float32):
            ok &= check(1 << 20, 8192, 32, "randn", gen, timing=True, dtype=dtype)
            ok &= check(1 << 20, 1024, 64, "randn", gen, timing=True, dtype=dtype)
        print("ALL OK" if ok else "FAILURES", flush=True)
        sys.exit(0 if ok else 1)
    only_cos = "--cosine" in sys.argv
    for cosine in ((True,) if only_cos else (False, True)):
        for dtype in (torch.bfloat16, torch.float32):
            for c in cases:
                ok &= check(*c, gen, dtype=dtype, cosine=cosine)
            if not quick:
                ok &= check(1 << 20, 1024, 256, "kaiming", gen, timing=True, dtype=dtype, cosine=cosine)
                if not cosine:
                    ok &= check(1 << 20, 1024, 256, "rows", gen, timing=True, dtype=dtype)
    print("ALL OK" if ok else "FAILURES", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
