"""scratch timing helper used during bring-up (not part of the contract; bench.py is)."""
import sys, time, torch
sys.path.insert(0, '.')
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
def run(N, C, D, dtype, train=True, iters=5):
    x = torch.randn(N, D, device=dev).to(dtype)
    e = torch.randn(C, D, device=dev) * 0.005
    cs = torch.ones(C, device=dev); ea = e.clone()
    for it in range(iters + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.time()
        packed = L.pack_codebook(e)
        r = L.assign(x, packed, e, want_q=True, want_sqerr=True)
        if train:
            cnt, es = L.ema_accumulate(x, r['idx'], C)
            L.ema_finalize(cs, ea, e, cnt, es, decay=0.8, eps=1e-5)
        loss = L.reduce_partials(r['sqerr_partials'], r['nblk'], 1.0 / (N * D))
    torch.cuda.synchronize(); dt = (time.time() - t0) / iters
    fl = 2.0 * N * C * D
    print(f"N={N} C={C} D={D} {dtype} train={train}: {dt*1e3:.3f} ms/step  {N/dt:.3e} vec/s  {fl/dt/1e12:.1f} TF/s ({fl/dt/157.3e12*100:.1f}% of fp32 MFMA peak) loss={loss.item():.5f}")
    # per-kernel
    for name, fn in [("pack", lambda: L.pack_codebook(e)), ("assign", lambda: L.assign(x, packed, e, want_q=True, want_sqerr=True)),
                     ("stats", lambda: L.ema_accumulate(x, r['idx'], C))]:
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3): fn()
        torch.cuda.synchronize(); print(f"   {name}: {(time.time()-t0)/3*1e3:.3f} ms")
run(1 << 20, 1024, 256, torch.bfloat16)
run(1 << 20, 1024, 256, torch.float32)
run(1 << 18, 1024, 256, torch.float32, train=False)
run(1 << 16, 4096, 128, torch.float32)
run(1 << 15, 8192, 512, torch.float32)
