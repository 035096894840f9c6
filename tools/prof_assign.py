"""Launches vq_assign_kernel at two codebook sizes (same rows) so that PMC differences isolate the
steady-state tile loop from the per-workgroup fixed costs.  Run under rocprofv3 --pmc ..."""
import sys, torch
sys.path.insert(0, '.')
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
N, D = 1 << 20, 256
x = torch.randn(N, D, device=dev).bfloat16()
for C in (1024, 2048):
    e = torch.randn(C, D, device=dev) * 0.005
    pk = L.pack_codebook(e)
    for _ in range(3):
        L.assign(x, pk, e, want_q=True, want_sqerr=True)
torch.cuda.synchronize()
