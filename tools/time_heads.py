"""Dev tool: multi-head modules (separate codebook per head) with the heads' searches batched into one launch set vs the per-head loop.
    python tools/time_heads.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import VectorQuantize, RandomProjectionQuantizer, _lib as L
import vector_quantize_pytorch_amd.codebook as cbmod
dev = torch.device("cuda:0")

def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

cases = [("RandomProjectionQuantizer(dim=512, codebook_size=1024, codebook_dim=16, num_codebooks=16), x=(8,1024,512)",
          RandomProjectionQuantizer(dim=512, codebook_size=1024, codebook_dim=16, num_codebooks=16).to(dev), torch.randn(8, 1024, 512, device=dev)),
         ("VectorQuantize(dim=512, heads=8, codebook_dim=64, separate_codebook_per_head, C=1024) eval, x=(8,1024,512)",
          VectorQuantize(dim=512, heads=8, codebook_dim=64, codebook_size=1024, separate_codebook_per_head=True).to(dev).eval(), torch.randn(8, 1024, 512, device=dev)),
         ("VectorQuantize(dim=512, heads=8, codebook_dim=64, separate_codebook_per_head, C=1024) train, x=(32,4096,512)",
          VectorQuantize(dim=512, heads=8, codebook_dim=64, codebook_size=1024, separate_codebook_per_head=True).to(dev).train(), torch.randn(32, 4096, 512, device=dev))]
supported = L.assign_batched_supported
for name, mod, x in cases:
    with torch.no_grad():
        cbmod.L.assign_batched_supported = supported
        t1 = tm(lambda: mod(x))
        cbmod.L.assign_batched_supported = lambda *a, **k: False
        t0 = tm(lambda: mod(x))
    print(f"{name}: batched {t1:.0f} us | per-head loop {t0:.0f} us")
