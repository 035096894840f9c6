"""Dev tool: s_memtime stamps inside vq_screenp_kernel (build: tools/build_p_variants.sh trace="-DVQP_TRACE"; VQHIP_SO is set here).
Per wave and barrier interval: M phase, wait at barrier X (role B), F phase (first staging wait, folds, second staging wait), wait at
barrier Y (role A)."""
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
L.assign(x, pk, e, want_q=True, q_out=q); torch.cuda.synchronize()
tr = torch.zeros(8 * 8 * NIV * 8, dtype=torch.int64, device=dev)
lib.vqhip_screenp_set_trace(ctypes.c_void_p(tr.data_ptr()))
L.assign(x, pk, e, want_q=True, q_out=q); torch.cuda.synchronize()
lib.vqhip_screenp_set_trace(ctypes.c_void_p(0))
t = tr.cpu().reshape(8, 8, NIV, 8).double()       # [wg, wave, interval, stamp]
names = ["M", "barX", "F:to stage wait 0", "F:folds to stage wait 1", "F:stage write 1", "F:rest", "barY"]
segs = [(0, 1), (1, 2), (2, 5), (5, 6), (6, 7), (7, 3), (3, 4)]
for role, ws in (("A (waves 0-3)", slice(0, 4)), ("B (waves 4-7)", slice(4, 8))):
    print("role", role)
    for nm, (a, b) in zip(names, segs):
        d = t[:, ws, 4:, b] - t[:, ws, 4:, a]
        print(f"   {nm:28s} mean {d.mean():7.0f}  min {d.min():7.0f}  max {d.max():7.0f}")
    per = t[:, ws, 5:, 0] - t[:, ws, 4:-1, 0]
    print(f"   period {per.mean():.0f}")
w = 0
print("wg 0 wave 0, intervals 16..23: M / F / barY:", [(int(t[0, w, i, 1] - t[0, w, i, 0]), int(t[0, w, i, 3] - t[0, w, i, 2]), int(t[0, w, i, 4] - t[0, w, i, 3])) for i in range(16, 24)])
w = 4
print("wg 0 wave 4, intervals 16..23: M / barX / F:", [(int(t[0, w, i, 1] - t[0, w, i, 0]), int(t[0, w, i, 2] - t[0, w, i, 1]), int(t[0, w, i, 3] - t[0, w, i, 2])) for i in range(16, 24)])
