"""Dev tool: s_memtime stamps inside vq_screenc_kernel (build: tools/build_c_variants.sh ctrace="-DVQC_TRACE"; VQHIP_SO is set here).
Per wave and interval of the kernel: cycles of work (start -> its wait) and cycles in the wait + barrier."""
import sys, os, ctypes, torch
os.environ.setdefault("VQHIP_SO", os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libvqhip_ctrace.so"))
os.environ["VQHIP_SCREEN_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
D, C = 256, 1024
g = torch.Generator(device=dev).manual_seed(0)
e = torch.empty(C, D, device=dev); torch.nn.init.kaiming_uniform_(e, generator=g)
pk = L.pack_codebook(e)
x = torch.randn(1 << 20, D, device=dev, generator=g).bfloat16()
q = torch.empty_like(x)
lib = L.lib()
lib.vqhip_screenc_set_trace.argtypes = [ctypes.c_void_p]
NIV = 64
want_q = "--noq" not in sys.argv
L.assign(x, pk, e, want_q=want_q, q_out=q if want_q else None); torch.cuda.synchronize()
tr = torch.zeros(16 * 4 * NIV * 8 + 512 * 4, dtype=torch.int64, device=dev)
lib.vqhip_screenc_set_trace(ctypes.c_void_p(tr.data_ptr()))
L.assign(x, pk, e, want_q=want_q, q_out=q if want_q else None); torch.cuda.synchronize()
lib.vqhip_screenc_set_trace(ctypes.c_void_p(0))
trc = tr.cpu()
info = trc[16 * 4 * NIV * 8:].reshape(512, 4)
def where(v):
    hw, xcc = int(v[0]), int(v[1])
    return (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)      # xcc, se, sh, cu
cu = {}
for b in range(512):
    cu.setdefault(where(info[b]), []).append(b)
print("distinct CUs:", len(cu), " workgroups per CU:", sorted(set(len(v) for v in cu.values())))
print("CU of workgroup 0:", where(info[0]), "shared with", cu[where(info[0])], " LDS_ALLOC regs:", [hex(int(info[b][2])) for b in cu[where(info[0])]],
      " start offsets (k cycles):", [round((int(info[b][3]) - int(info[0][3])) / 1e3, 1) for b in cu[where(info[0])]])
print("partner of workgroup b (first 10):", [(b, [x for x in cu[where(info[b])] if x != b]) for b in range(10)])
t = trc[:16 * 4 * NIV * 8].reshape(16, 4, NIV, 8).double()       # [wg slot (0..7: workgroups 0..7, 8..15: workgroups 256..263), wave, interval, stamp]
nst = C // 64
T = nst + 3
t0 = t[0, 0, 0, 0]
print("workgroup 0 (wave 0) and workgroup 256 (wave 0): interval, kind, start (k cycles since wg 0's first stamp), work, wait+barrier")
for i in range(0, 3 * T):
    k = "sweep" if (i % T) < nst else "BREAK%d" % ((i % T) - nst)
    a = t[0, 0, i]; b = t[8, 0, i]
    print(f" {i:3d} {k:7s} | {(a[0]-t0)/1e3:8.1f} {a[1]-a[0]:6.0f} {a[2]-a[1]:6.0f} | {(b[0]-t0)/1e3:8.1f} {b[1]-b[0]:6.0f} {b[2]-b[1]:6.0f}")
for lo, nm in ((0, "workgroups 0..7"), (8, "workgroups 256..263")):
    w = t[lo:lo + 8]
    sw = [i for i in range(T, 3 * T) if (i % T) < nst]
    br = [[i for i in range(T, 3 * T) if (i % T) == nst + k] for k in range(3)]
    print(nm, "mean sweep interval: work %.0f wait %.0f" % ((w[:, :, sw, 1] - w[:, :, sw, 0]).mean(), (w[:, :, sw, 2] - w[:, :, sw, 1]).mean()),
          " breaks (work/wait):", [(round(float((w[:, :, b, 1] - w[:, :, b, 0]).mean())), round(float((w[:, :, b, 2] - w[:, :, b, 1]).mean()))) for b in br],
          " period %.0f" % ((w[:, :, 3 * T, 0] - w[:, :, T, 0]).mean() / 2))

names = ["final fold", "merge, classify, index store, list slots", "q rows", "list entries", "row requests", "wait + barrier"]
for per in range(3):
    i = per * T + nst
    w = t[:, :, i]
    pts = [w[..., 0], w[..., 3], w[..., 4], w[..., 5], w[..., 6], w[..., 7], w[..., 2]]
    print(f"BREAK0 of period {per}:", ", ".join(f"{nm} {float((pts[k + 1] - pts[k]).mean()):.0f}" for k, nm in enumerate(names)))
