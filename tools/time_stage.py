"""Dev tool: one residual-VQ stage (fp32 rows, x - q out) timed against the number of codes: the intercept at C -> 0 is the
memory side of the kernel (load + convert, outputs), the slope the sweep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 262144
x = torch.randn(N, D, device=dev, generator=g)
res = torch.empty_like(x)
def t(C, **kw):
    e = torch.empty(C, D, device=dev); torch.nn.init.kaiming_uniform_(e, generator=g)
    pk = L.pack_codebook(e)
    for _ in range(3): L.assign(x, pk, e, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): L.assign(x, pk, e, **kw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 * 1e3
for C in (32, 256, 1024, 2048, 4096):
    print(f"D={D} C={C}: idx only {t(C, want_q=False):.0f} us | +sqerr {t(C, want_q=False, want_sqerr=True):.0f} | +resid+sqerr {t(C, want_q=False, want_sqerr=True, resid_out=res):.0f}")
