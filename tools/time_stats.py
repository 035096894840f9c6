"""Dev tool: the EMA statistics pass (sort + segmented sum, with and without the loss) at cfg-2 size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
N, C, D = 1 << 20, 1024, 256
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
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(N, D, device=dev, generator=g).to(dt)
    e = torch.randn(C, D, device=dev, generator=g)
    pk = L.pack_codebook(e)
    idx = L.assign(x, pk, e, want_q=False)["idx"]
    cnt = torch.zeros(C, device=dev); es = torch.zeros(C, D, device=dev)
    print(f"{dt}: stats {t(lambda: L.ema_accumulate(x, idx, C, count=cnt, embed_sum=es)):.0f} us | stats + loss {t(lambda: L.ema_accumulate(x, idx, C, count=cnt, embed_sum=es, sqerr_from=(pk, e))):.0f} us")
