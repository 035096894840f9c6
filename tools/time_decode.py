"""Dev tool: vqhip_decode_sum at the cfg-3 (Q = 8, shared codebook, D = 256) and cfg-4 (Q = 1, D = 512) shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    return best
for name, N, Q, C, D in (("cfg3", 262144, 8, 1024, 256), ("cfg4", 262144, 1, 8192, 512), ("cfg5 group", 262144, 8, 4096, 128)):
    e = torch.randn(C, D, device=dev, generator=g) if name != "cfg5 group" else torch.randn(Q, C, D, device=dev, generator=g)
    idx = torch.randint(0, C, (N, Q), device=dev, generator=g)
    us = t(lambda: L.decode_sum(idx, e))
    ref = (e[idx].sum(1) if e.ndim == 2 else sum(e[q][idx[:, q]] for q in range(Q)))
    out = L.decode_sum(idx, e)
    print(f"{name}: {us:.0f} us  ({N * D * 4 / us / 1e6:.2f} TB/s of output)  max |err| vs torch {float((out - ref).abs().max()):.2e}")
