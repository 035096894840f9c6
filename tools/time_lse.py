"""Dev tool: the exact kernel's callers VERDICT r3 #7 named -- vqhip_scores_lse (streaming log-sum-exp) and vqhip_assign with q rows -- at
D = 128 and D = 512, fp32 and bf16 rows.   python tools/time_lse.py   (run it from another checkout to compare builds)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
os.environ["VQHIP_SCREEN"] = "0"
def tm(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator(device=dev).manual_seed(0)
for D, C in ((128, 4096), (512, 1024)):
    for dt in (torch.float32, torch.bfloat16):
        N = 1 << 18
        x = torch.randn(N, D, device=dev, generator=g).to(dt)
        e = torch.randn(C, D, device=dev, generator=g)
        pk = L.pack_codebook(e)
        tgt = torch.randint(0, C, (N,), device=dev, generator=g)
        t_lse = tm(lambda: L.scores_lse(x, pk, e, tgt))
        t_q = tm(lambda: L.assign(x, pk, e, want_q=True, want_sqerr=True))
        print(f"D={D} C={C} {str(dt)[6:]}: scores_lse {t_lse:.0f} us | exact assign + q + sqerr {t_q:.0f} us", flush=True)
