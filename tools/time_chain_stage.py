"""Dev tool: the screening kernel of ONE residual-chain stage alone (VQHIP_SCREEN_ONLY=1), fp32 rows: a first stage (plain rows) against
a chained stage (previous input + gathered previous code, result stored / not stored: VQHIP_CHAIN_NOWRITE=1, read once per process),
under VQHIP_SCREEN_STAGGER (read once per process).  Event-timed, best of 3 x 10 launches.
    VQHIP_SCREEN_STAGGER=40 python tools/time_chain_stage.py [D] [C] [log2 N]"""
import ctypes, os, sys, torch
os.environ["VQHIP_SCREEN_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 18)
x = torch.randn(N, D, device=dev, generator=g)
e = torch.randn(C, D, device=dev, generator=g) * 0.3
pk = L.pack_codebook(e)
idx = torch.randint(0, C, (N, 2), device=dev, generator=g)
out = torch.empty_like(x)
nws = L.lib().vqhip_screen_workspace_bytes(N)
ws = torch.zeros((nws + 15) // 16 * 4, dtype=torch.int32, device=dev)


def run(chained):
    ws[:4].zero_()
    ch = L._Chain(idx_stride=2, prev_idx=None, prev_idx_stride=2, prev_embed=None, x_out=None, ldxo=D, route_mode=0, header_zeroed=1)
    if chained:
        ch.prev_idx, ch.prev_embed, ch.x_out = idx.data_ptr(), e.data_ptr(), out.data_ptr()
    L._check(L.lib().vqhip_assign_screened_chain(L._ptr(x), 0, N, D, D, L._ptr(pk), L._ptr(e), C, 0, ctypes.c_void_p(idx.data_ptr() + 8), None,
                                                L._ptr(ws), nws, ctypes.byref(ch), L._stream()), "chain")


def t(chained):
    for _ in range(3): run(chained)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(10): run(chained)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10 * 1e3)
    return best - 3.0          # (the 16-byte header fill beside every launch: ~3 us, the same in both columns)


print(f"D={D} C={C} N={N} stagger={os.environ.get('VQHIP_SCREEN_STAGGER', '0')} nowrite={os.environ.get('VQHIP_CHAIN_NOWRITE', '0')}: "
      f"first stage {t(False):6.1f} us   chained stage {t(True):6.1f} us")
