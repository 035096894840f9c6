"""Dev tool / test helper: the screening kernel that serves bf16 rows at D = 256, N >= 2^17, idx [+ q] outputs -- the persistent one of
csrc/vq_screen_c.hip by default, the 4-wave kernel under VQHIP_SCREEN_PERSIST=0 -- against the exact fp32-MFMA kernel, bit for bit
(idx, q).   python tools/persist_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)

def codebook(kind, C, D, x):
    if kind == "kaiming":
        e = torch.empty(C, D, device=dev); torch.nn.init.kaiming_uniform_(e, generator=g); return e
    if kind == "randn": return torch.randn(C, D, device=dev, generator=g)
    if kind == "rows": return x[torch.randperm(x.shape[0], device=dev, generator=g)[:C]].float().contiguous()
    if kind == "dups":
        e = torch.randn(C, D, device=dev, generator=g); e[C // 2:] = e[: C - C // 2]; return e
    if kind == "tiny": return torch.randn(C, D, device=dev, generator=g) * 1e-3
    # the persistent kernel picks its sweep from the codebook's norm spread (plain scores up to 4 x, upper bounds beyond): both sides of
    # the switch, one huge code, one all-zero code (no plain mode: min ||c|| = 0), twins inside a spread codebook
    if kind in ("ratio3.9", "ratio4.1"):
        e = torch.nn.functional.normalize(torch.randn(C, D, device=dev, generator=g), dim=-1) * 16.0
        s = torch.exp(torch.empty(C, 1, device=dev).uniform_(0.0, 1.0, generator=g) * torch.log(torch.tensor(float(kind[5:]))).item())
        s[0], s[1] = 1.0, float(kind[5:])
        e = e * s
        e[C - 8:] = e[8:16]
        return e
    if kind == "x100":
        e = torch.randn(C, D, device=dev, generator=g); e[7] *= 100.0; return e
    if kind == "zerocode":
        e = torch.randn(C, D, device=dev, generator=g) * 0.1; e[5] = 0.0; return e      # (small codes: the zero code wins only some rows)
    raise ValueError(kind)

def rows(kind, N, D):
    x = torch.randn(N, D, device=dev, generator=g)
    if kind == "wild":      # row norms over 8 decades inside every super-block, some zero rows
        x = x * torch.exp(torch.empty(N, 1, device=dev).uniform_(-9.0, 9.0, generator=g))
        x[::97] = 0
    if kind == "big": x = x * 3e4
    if kind == "small": x = x * 1e-6
    return x.to(torch.bfloat16)

def run(x, e, pk, screened, cosine, want_q):
    os.environ["VQHIP_SCREEN"] = "1" if screened else "0"
    q = torch.empty_like(x) if want_q else None
    if cosine and screened:
        r = L.assign(L.l2norm_rows(x), pk, e, cosine=True, skip_l2norm=True, want_q=want_q, q_out=q)
    else:
        r = L.assign(x, pk, e, cosine=cosine, want_q=want_q, q_out=q)
    os.environ["VQHIP_SCREEN"] = "1"
    return r, q

ok = True
cases = [(1 << 17, 1024, "kaiming", "plain", False, True), (1 << 17, 1024, "kaiming", "plain", False, False),
         ((1 << 17) - 77, 1024, "randn", "plain", False, True), (70001, 2048, "randn", "plain", False, True),
         (1 << 16, 1024, "dups", "plain", False, True), (1 << 16, 1024, "tiny", "plain", False, True),
         (1 << 17, 1024, "rows", "wild", False, True), (1 << 17, 1024, "randn", "big", False, True),
         (1 << 17, 1024, "randn", "small", False, True), (1 << 17, 1000, "randn", "plain", False, True),
         (1 << 17, 1024, "kaiming", "plain", True, True), (99999, 4096, "randn", "wild", True, True),
         (1 << 17, 1024, "ratio3.9", "plain", False, True), (1 << 17, 1024, "ratio4.1", "plain", False, True),
         (1 << 17, 1024, "ratio3.9", "wild", False, True), (1 << 17, 1024, "ratio4.1", "wild", False, False),
         (1 << 17, 1024, "x100", "plain", False, True), (1 << 17, 1000, "zerocode", "plain", False, True),
         (1 << 20, 1024, "kaiming", "plain", False, True)]
for (N, C, ck, rk, cosine, want_q) in cases:
    x = rows(rk, N, 256)
    e = codebook(ck, C, 256, x)
    if cosine: e = torch.nn.functional.normalize(e, dim=-1)
    pk = L.pack_codebook(e)
    r0, q0 = run(x, e, pk, False, cosine, want_q)
    r1, q1 = run(x, e, pk, True, cosine, want_q)
    torch.cuda.synchronize()
    bad = int((r0["idx"] != r1["idx"]).sum())
    qeq = (not want_q) or bool(torch.equal(q0, q1))
    print(f"N={N} C={C} {ck:8s} rows={rk:6s} cos={int(cosine)} q={int(want_q)}: idx bad {bad}, q_equal {qeq}, open {int(r1['n_exact'][0])} pair {int(r1['n_pair'][0])}", flush=True)
    ok &= bad == 0 and qeq
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
