"""Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on the assign kernel's own access pattern (the guide:
"calibrate on a known byte count in your own access pattern"): one 32-code tile, so the kernel is pure
x-read (+ q-write) traffic of known size.  N = 2^21 rows so that x (1 / 2 GiB) exceeds the 256 MiB MALL."""
import sys, torch
sys.path.insert(0, '.')
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
N, D, C = 1 << 21, 256, 32
e = torch.randn(C, D, device=dev)
pk = L.pack_codebook(e)
for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn(N, D, device=dev).to(dtype)
    for _ in range(2):
        L.assign(x, pk, e, want_q=False)              # reads N*D*s bytes, writes 8N
    q = torch.empty_like(x)
    for _ in range(2):
        L.assign(x, pk, e, want_q=True, q_out=q)      # + writes N*D*s
    del x, q
torch.cuda.synchronize()
