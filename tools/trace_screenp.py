"""Dev tool: s_memtime stamps inside vq_screenp_kernel (build: tools/build_p_variants.sh trace="-DVQP_TRACE"; VQHIP_SO is set here).
Per wave and barrier interval (two tiles, 64 MFMAs): cycles from the interval's start to its barrier (k-step 12 of the second tile),
the wait at the barrier, the rest of the interval, and whatever sits between two intervals (block-level work)."""
import sys, os, ctypes, torch
os.environ.setdefault("VQHIP_SO", os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libvqhip_trace.so"))
os.environ["VQHIP_SCREEN_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VQHIP_SCREEN_PERSIST", "1")
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
D, C = 256, 1024
g = torch.Generator(device=dev).manual_seed(0)
e = torch.empty(C, D, device=dev); torch.nn.init.kaiming_uniform_(e, generator=g)
pk = L.pack_codebook(e)
x = torch.randn(1 << 20, D, device=dev, generator=g).bfloat16()
q = torch.empty_like(x)
lib = L.lib()
lib.vqhip_screenp_set_trace.argtypes = [ctypes.c_void_p]
NIV = 48
want_q = "--noq" not in sys.argv
L.assign(x, pk, e, want_q=want_q, q_out=q if want_q else None); torch.cuda.synchronize()
tr = torch.zeros(8 * 8 * NIV * 8, dtype=torch.int64, device=dev)
lib.vqhip_screenp_set_trace(ctypes.c_void_p(tr.data_ptr()))
L.assign(x, pk, e, want_q=want_q, q_out=q if want_q else None); torch.cuda.synchronize()
lib.vqhip_screenp_set_trace(ctypes.c_void_p(0))
t = tr.cpu().reshape(8, 8, NIV, 8)[:, :4].double()       # [wg, wave, interval, stamp]
nst = C // 64
print("mean over workgroups 0..7, waves 0..3; by interval of the block (block 1 = intervals %d..%d):" % (nst, 2 * nst - 1))
print("  I   start->barrier   barrier wait   barrier->end   to next interval   period")
for i in range(nst, 2 * nst):
    a = (t[:, :, i, 1] - t[:, :, i, 0]).mean(); b = (t[:, :, i, 2] - t[:, :, i, 1]).mean(); c = (t[:, :, i, 3] - t[:, :, i, 2]).mean()
    d = (t[:, :, i + 1, 0] - t[:, :, i, 3]).mean(); p = (t[:, :, i + 1, 0] - t[:, :, i, 0]).mean()
    print(f" {i - nst:2d}   {a:8.0f}         {b:8.0f}       {c:8.0f}        {d:8.0f}        {p:8.0f}")
i = 2 * nst - 1
names = ["interval end -> final fold", "final fold + bookkeeping", "merge, classify, index store", "atomic issue .. its wait", "-> next interval"]
pts = [t[:, :, i, 3], t[:, :, i, 4], t[:, :, i, 5], t[:, :, i, 6], t[:, :, i, 7], t[:, :, i + 1, 0]]
for k, nm in enumerate(names):
    print(f"   block end: {nm:32s} {(pts[k + 1] - pts[k]).mean():8.0f}")
blk = (t[:, :, 2 * nst, 0] - t[:, :, nst, 0]).mean()
print(f"block period {blk:.0f} cycles = {blk / (32 * nst * 2):.1f} per MFMA ({32 * nst * 2 // 32} MFMAs)")
print("wg 0, waves 0..3, barrier waits of block 1:", [[int(t[0, w, i, 2] - t[0, w, i, 1]) for i in range(nst, 2 * nst)] for w in range(4)])
