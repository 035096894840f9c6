"""Dev tool: the search of a wide-dim codebook by itself (vqhip_assign on fp32 rows, index output only), event-timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
for dim, C, n in ((768, 1024, 65536), (1024, 512, 8192), (2048, 1024, 32768)):
    x = torch.randn(n, dim, device=dev); e = torch.randn(C, dim, device=dev)
    pk = L.pack_codebook(e)
    for _ in range(3): L.assign(x, pk, e, want_q=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): L.assign(x, pk, e, want_q=False)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"search dim={dim} C={C} rows={n}: {ms:.3f} ms = {2.0 * n * C * dim / ms / 1e9:.1f} TFLOP/s = {2.0 * n * C * dim / ms / 1e9 / 157.3:.2f} of the fp32 MFMA peak")
