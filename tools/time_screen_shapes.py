"""Dev tool: the screening kernel ALONE (VQHIP_SCREEN_ONLY=1: the listed rows stay undecided) at the shapes of BASELINE configs 2, 3
and 5 (one stage / one group), event-timed, best of 3 x 10 launches.   VQHIP_SO=... python tools/time_screen_shapes.py"""
import os, sys, torch
os.environ["VQHIP_SCREEN_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("cfg2  bf16 N=2^20 C=1024 D=256", 1 << 20, 1024, 256, torch.bfloat16, True),
          ("cfg3  fp32 N=2^18 C=1024 D=256", 1 << 18, 1024, 256, torch.float32, False),
          ("cfg5  fp32 N=2^18 C=4096 D=128", 1 << 18, 4096, 128, torch.float32, False),
          ("      bf16 N=2^20 C=1024 D=128", 1 << 20, 1024, 128, torch.bfloat16, True),
          ("      bf16 N=2^20 C=1024 D=64 ", 1 << 20, 1024, 64, torch.bfloat16, True)]
for name, N, C, D, dt, want_q in shapes:
    x = torch.randn(N, D, device=dev, generator=g).to(dt)
    e = torch.randn(C, D, device=dev, generator=g) * 0.3
    pk = L.pack_codebook(e)
    q = torch.empty_like(x) if want_q else None
    kw = dict(want_q=want_q, q_out=q) if want_q else dict(want_q=False)
    for _ in range(3): L.assign(x, pk, e, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(10): r = L.assign(x, pk, e, **kw)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10 * 1e3)
    tf = 2.0 * N * C * D / (best * 1e-6) / 1e12
    print(f"{name}: {best:7.1f} us  {tf:6.0f} TFLOP/s  uncertified {int(r['n_exact'][0]) + int(r['n_pair'][0])}")
