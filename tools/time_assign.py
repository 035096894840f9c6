"""Dev tool: event-timed L.assign at cfg-2 size for output combinations (which part of the output phase costs what)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
N, C, D = 1 << 20, 1024, 256
dt = torch.float32 if "--f32" in sys.argv else torch.bfloat16
x = torch.randn(N, D, device=dev, generator=g).to(dt)
e = torch.empty(C, D, device=dev); torch.nn.init.kaiming_uniform_(e, generator=g)
pk = L.pack_codebook(e)
def t(**kw):
    for _ in range(3): L.assign(x, pk, e, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):            # best of 3 batches of 20 launches
        a.record()
        for _ in range(20): r = L.assign(x, pk, e, **kw)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    global last
    last = r
    return best
q = torch.empty(N, D, device=dev, dtype=dt)
print(f"{dt}: idx only {t(want_q=False):.0f} us | +q {t(want_q=True, q_out=q):.0f} | +q+sqerr {t(want_q=True, q_out=q, want_sqerr=True):.0f} | sqerr only {t(want_q=False, want_sqerr=True):.0f}")
if last.get("n_exact") is not None: print("   open rows", int(last["n_exact"][0]), "pair rows", int(last["n_pair"][0]))
