"""Dev tool: s_memtime stamps inside vq_screen16_kernel (build: tools/build_variants.sh trace="-DVQ_TRACE -DVQ_TRACE_BLOCK0=2048").
Phases per workgroup (load x + scale + convert | sweep | merge / idx / list | output rows) and, inside the sweep, per barrier
interval (SUB tiles): wait at the barrier, MFMA + staging + top-2 of the interval."""
import sys, os, ctypes, torch
os.environ.setdefault("VQHIP_SO", os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libvqhip_trace.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
D, C = 256, 1024
g = torch.Generator(device=dev).manual_seed(0)
e = torch.empty(C, D, device=dev)
torch.nn.init.kaiming_uniform_(e, generator=g)
pk = L.pack_codebook(e)
F32 = len(sys.argv) > 1 and sys.argv[1] == 'f32'      # fp32 rows with a residual output: one stage of the cfg-3 loop
xf = torch.randn(1 << 20, D, device=dev, generator=g)
if not F32: xf = xf.bfloat16()
resid = torch.empty_like(xf) if F32 else None
kw = (lambda x: dict(want_q=False, want_sqerr=True, resid_out=resid[: x.shape[0]])) if F32 else (lambda x: dict(want_q=True, want_sqerr=True))
lib = L.lib()
lib.vqhip_set_trace.argtypes = [ctypes.c_void_p]
NI = 16   # barrier intervals at C = 1024, SUB = 2
for blocks in [int(b) for b in os.environ.get('TRACE_BLOCKS', '256,512,4096').split(',')]:
    x = xf[: blocks * 256]
    L.assign(x, pk, e, **kw(x)); torch.cuda.synchronize()
    tr = torch.zeros(16 * 4 * 64 * 4 + 16 * 4 * 8, dtype=torch.int64, device=dev)
    lib.vqhip_set_trace(ctypes.c_void_p(tr.data_ptr()))
    L.assign(x, pk, e, **kw(x)); torch.cuda.synchronize()
    lib.vqhip_set_trace(ctypes.c_void_p(0))
    ph = tr.cpu()[16 * 4 * 64 * 4:].reshape(16, 4, 8).double()
    d = [(ph[:, :, i + 1] - ph[:, :, i]).mean().item() for i in range(4)]
    print(f"blocks={blocks}: phases (cycles): load x + convert {d[0]:.0f} | sweep {d[1]:.0f} | merge/idx/list {d[2]:.0f} | output rows {d[3]:.0f} | total {sum(d):.0f}")
    t = tr.cpu()[: 16 * 4 * 64 * 4].reshape(16, 4, 64, 4)[:, :, :NI].double()
    bar = t[..., 1] - t[..., 0]
    work = t[..., 2] - t[..., 1]
    period = t[:, :, 1:, 0] - t[:, :, :-1, 0]
    print(f"   per interval: barrier wait {bar[:, :, 1:].mean():.0f} (max {bar[:, :, 1:].max():.0f})  mfma+stage+top2 {work.mean():.0f}  period {period.mean():.0f}  (pure MFMA issue of one wave: {2 * 2 * 16 * 32})")
    print("   block0 per wave, interval 8: barrier", [int(bar[0, w, 8]) for w in range(4)], " work", [int(work[0, w, 8]) for w in range(4)])
