"""Dev tool: time of one nearest-code search (idx + q + squared error) and of one VectorQuantize train step vs N,
screened vs exact path, bf16 and fp32 rows.  C = 1024, D = 256."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import VectorQuantize, _lib as L
dev = torch.device('cuda:0')

def tm(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

D, C = 256, 1024
torch.manual_seed(0)
vq = VectorQuantize(dim=D, codebook_size=C).to(dev).train()
e = vq.codebook.clone().contiguous()
pk = L.pack_codebook(e)
print("| rows | dtype | search, screened (µs) | search, exact (µs) | train step, screened (µs) | train step, exact (µs) |")
print("|---|---|---|---|---|---|")
for dt in (torch.bfloat16, torch.float32):
    xf = torch.randn(1 << 20, D, device=dev).to(dt)
    for n in (1024, 4096, 16384, 65536, 262144, 1048576):
        x = xf[:n]
        res = []
        for scr in ("1", "0"):
            os.environ["VQHIP_SCREEN"] = scr
            res.append(tm(lambda: L.assign(x, pk, e, want_q=True, want_sqerr=True)))
        for scr in ("1", "0"):
            os.environ["VQHIP_SCREEN"] = scr
            with torch.no_grad():
                res.append(tm(lambda: vq(x[None])))
        print(f"| {n} | {str(dt)[6:]} | {res[0]:.0f} | {res[1]:.0f} | {res[2]:.0f} | {res[3]:.0f} |", flush=True)
