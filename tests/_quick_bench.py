"""scratch timing helper used during bring-up (not part of the contract; bench.py is)."""
import sys, time, torch
sys.path.insert(0, '.')
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')

def tm(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

N, D = 1 << 20, 256
for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn(N, D, device=dev).to(dtype)
    for C in (32, 64, 256, 1024, 2048):
        e = torch.randn(C, D, device=dev) * 0.005
        packed = L.pack_codebook(e)
        t_full = tm(lambda: L.assign(x, packed, e, want_q=True, want_sqerr=True))
        t_idx = tm(lambda: L.assign(x, packed, e, want_q=False, want_sqerr=False))
        print(f"assign {dtype} C={C}: full {t_full:.3f} ms  idx-only {t_idx:.3f} ms   per-tile {(t_full)/(C/32)*1e3:.1f} us")
    e = torch.randn(1024, D, device=dev) * 0.005
    r = L.assign(x, L.pack_codebook(e), e)
    print(f"stats {dtype}: {tm(lambda: L.ema_accumulate(x, r['idx'].reshape(-1), 1024)):.3f} ms")
    idx_same = torch.zeros(N, dtype=torch.int64, device=dev)
    print(f"stats {dtype} (all rows -> code 0, worst-case contention): {tm(lambda: L.ema_accumulate(x, idx_same, 1024)):.3f} ms")
x = torch.randn(1 << 18, 128, device=dev)
e = torch.randn(4096, 128, device=dev)
r = L.assign(x, L.pack_codebook(e), e)
print(f"stats f32 N=2^18 D=128 C=4096: {tm(lambda: L.ema_accumulate(x, r['idx'].reshape(-1), 4096)):.3f} ms; assign {tm(lambda: L.assign(x, L.pack_codebook(e), e)):.3f} ms")

# ---- cfg 3: ResidualVQ Q=8 shared codebook, x = (32, 8192, 256) fp32 ----
from vector_quantize_pytorch_amd import ResidualVQ
torch.manual_seed(0)
rvq = ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True).to(dev).train()
xr = torch.randn(32, 8192, 256, device=dev)
with torch.no_grad():
    t = tm(lambda: rvq(xr), n=3)
    print(f"cfg3 RVQ train step (fused): {t:.3f} ms  -> {32*8192/t*1e3:.3e} vec/s, {32*8192*8/t*1e3:.3e} vec-stage/s, {2*32*8192*8*1024*256/t/1e9:.1f} TF/s")
    rvq.eval()
    t = tm(lambda: rvq(xr), n=3)
    print(f"cfg3 RVQ eval (fused): {t:.3f} ms")
    e = rvq.layers[0]._codebook.embed[0]; pk = L.pack_codebook(e)
    t = tm(lambda: L.rvq_forward(xr, pk, e, 8, want_resid=True, want_sqerr=True), n=3)
    print(f"   vq_rvq_kernel alone (with residual dump): {t:.3f} ms")
    t = tm(lambda: L.rvq_forward(xr, pk, e, 8), n=3)
    print(f"   vq_rvq_kernel alone (indices only): {t:.3f} ms")
