"""Dev tool: one training step of VectorQuantize at a dim beyond 512 (csrc/vq_wide.hip), event-timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import VectorQuantize
dev = torch.device("cuda:0")
for dim, C, n in ((1024, 512, 8192), (768, 1024, 65536), (2048, 256, 8192)):
    vq = VectorQuantize(dim=dim, codebook_size=C).to(dev).train()
    x = torch.randn(1, n, dim, device=dev)
    with torch.no_grad():
        for _ in range(3): vq(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): vq(x)
        b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"VectorQuantize(dim={dim}, codebook_size={C}) train step, {n} fp32 rows: {ms:.3f} ms = {2.0 * n * C * dim / ms / 1e9:.1f} TFLOP/s (algorithmic)")
