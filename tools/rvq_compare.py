"""Dev tool: ResidualVQ cfg 3 with bf16 input -- fused exact-fp32 residual kernel vs per-stage path (screened assigns)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import ResidualVQ

def bench(mod, x, n=10):
    with torch.no_grad():
        for _ in range(3):
            out = mod(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            out = mod(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out

for dtype in (torch.bfloat16, torch.float32):
    torch.manual_seed(0)
    x = torch.randn(32, 8192, 256, device="cuda").to(dtype)
    res = {}
    for mode in ("fused", "staged"):
        torch.manual_seed(0)
        mod = ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True).cuda().train()
        if mode == "staged":
            mod._fused_eligible = lambda *a, **k: False
        ms, out = bench(mod, x)
        res[mode] = out
        print(f"{dtype} {mode}: {ms:.3f} ms/step", flush=True)
    print("  indices equal:", bool((res["fused"][1] == res["staged"][1]).all()), flush=True)
