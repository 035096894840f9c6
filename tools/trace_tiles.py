import sys, os, ctypes, torch
os.environ["VQHIP_SO"] = "tools/libvqhip_trace.so"
sys.path.insert(0, '.')
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
D, C = 256, 1024
e = torch.randn(C, D, device=dev) * 0.005
pk = L.pack_codebook(e)
xf = torch.randn(1 << 20, D, device=dev).bfloat16()
lib = L.lib()
for blocks in (256, 8192):
    x = xf[: blocks * 128]
    L.assign(x, pk, e, want_q=False); torch.cuda.synchronize()
    tr = torch.zeros(16 * 4 * 64 * 4, dtype=torch.int64, device=dev)
    lib.vqhip_set_trace.argtypes = [ctypes.c_void_p]
    lib.vqhip_set_trace(ctypes.c_void_p(tr.data_ptr()))
    L.assign(x, pk, e, want_q=False); torch.cuda.synchronize()
    lib.vqhip_set_trace(ctypes.c_void_p(0))
    t = tr.cpu().reshape(16, 4, 64, 4)[:, :, :32].double()
    bar = (t[..., 1] - t[..., 0])           # barrier wait
    mf = (t[..., 2] - t[..., 1])            # mfma phase (incl. staged copy)
    ep = (t[..., 3] - t[..., 2])            # epilogue
    tile = t[:, :, 1:, 0] - t[:, :, :-1, 0]
    print(f"blocks={blocks}: per tile (cycles of the 100 MHz?? counter units as read): barrier {bar[:, :, 1:].mean():.0f} (max {bar[:, :, 1:].max():.0f})  mfma {mf.mean():.0f}  epilogue {ep.mean():.0f}  tile period {tile.mean():.0f}")
    print("   block0 wave0 tiles 4..9 [barrier, mfma, epi]:", [(int(bar[0,0,i]), int(mf[0,0,i]), int(ep[0,0,i])) for i in range(4, 10)])
    print("   block0 all waves tile 8 barrier waits:", [int(bar[0,w,8]) for w in range(4)], " epilogue:", [int(ep[0,w,8]) for w in range(4)])
