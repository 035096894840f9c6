import sys, torch
sys.path.insert(0, '.')
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
def tm(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
D, C = 256, 1024
e = torch.randn(C, D, device=dev) * 0.005
pk = L.pack_codebook(e)
xf = torch.randn(1 << 20, D, device=dev).bfloat16()
for blocks in (256, 512, 768, 1024, 1536, 2048, 4096, 8192):
    x = xf[: blocks * 128]
    t_full = tm(lambda: L.assign(x, pk, e, want_q=True, want_sqerr=True))
    t_idx = tm(lambda: L.assign(x, pk, e, want_q=False))
    print(f"blocks={blocks:5d} rows={blocks*128:8d}  full {t_full*1e3:8.1f} us  idx-only {t_idx*1e3:8.1f} us   ideal-mfma {blocks*128/32*32*128*64/1024/2.4e3:8.1f} us")
