"""Dev tool: s_memtime stamps inside vq_screen_kernel (build: -DVQ_TRACE into tools/variants/libvqhip_trace.so)."""
import sys, os, ctypes, torch
os.environ["VQHIP_SO"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libvqhip_trace.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
D, C = 256, 1024
g = torch.Generator(device=dev).manual_seed(0)
e = torch.empty(C, D, device=dev)
torch.nn.init.kaiming_uniform_(e, generator=g)
pk = L.pack_codebook(e)
xf = torch.randn(1 << 20, D, device=dev, generator=g).bfloat16()
lib = L.lib()
lib.vqhip_set_trace.argtypes = [ctypes.c_void_p]
for blocks in (256, 512, 4096):
    x = xf[: blocks * 256]
    L.assign(x, pk, e, want_q=True, want_sqerr=True); torch.cuda.synchronize()
    tr = torch.zeros(16 * 4 * 64 * 4 + 16 * 4 * 8, dtype=torch.int64, device=dev)
    lib.vqhip_set_trace(ctypes.c_void_p(tr.data_ptr()))
    L.assign(x, pk, e, want_q=True, want_sqerr=True); torch.cuda.synchronize()
    lib.vqhip_set_trace(ctypes.c_void_p(0))
    ph = tr.cpu()[16 * 4 * 64 * 4:].reshape(16, 4, 8).double()
    d = [(ph[:, :, i + 1] - ph[:, :, i]).mean().item() for i in range(4)]
    print(f"blocks={blocks}: phases (cycles): load x + eps {d[0]:.0f} | sweep {d[1]:.0f} | merge/idx/list {d[2]:.0f} | q rows + sqerr {d[3]:.0f}")
    t = tr.cpu()[: 16 * 4 * 64 * 4].reshape(16, 4, 64, 4)[:, :, :32].double()
    bar = t[..., 1] - t[..., 0]
    mf = t[..., 2] - t[..., 1]
    ep = t[..., 3] - t[..., 2]
    tile = t[:, :, 1:, 0] - t[:, :, :-1, 0]
    total = t[:, :, 31, 3] - t[:, :, 0, 0]
    print(f"blocks={blocks}: barrier {bar[:, :, 1:].mean():.0f} (max {bar[:, :, 1:].max():.0f})  mfma phase {mf.mean():.0f}  top2 {ep.mean():.0f}  tile period {tile.mean():.0f}  sweep total {total.mean():.0f}")
    print("   block0 wave0 tiles 4..9 [barrier, mfma, top2]:", [(int(bar[0,0,i]), int(mf[0,0,i]), int(ep[0,0,i])) for i in range(4, 10)])
    print("   block0 tile 8 all waves barrier:", [int(bar[0,w,8]) for w in range(4)], " mfma:", [int(mf[0,w,8]) for w in range(4)], " top2:", [int(ep[0,w,8]) for w in range(4)])
