"""round 6: the searches of a GroupedResidualVQ FIRST forward (k-means init, vqp.py:238-278) as the module issues them:
per L.assign call the open / pair fractions read at once, the screening kernel's margin / threshold quantiles and the codebook's
norm range.  (bench.py's r5 line showed iteration 0 of every codebook at 98 % open; a stand-alone iteration 0 on randn rows shows
0.2 %: tools/kmeans_iter0_debug.py.)

    python tools/kmeans_module_debug.py [calls_to_print]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vector_quantize_pytorch_amd import GroupedResidualVQ, _lib as L
import vector_quantize_pytorch_amd.codebook as cbmod


def main():
    n_print = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    mod = GroupedResidualVQ(dim=512, groups=4, num_quantizers=8, codebook_size=4096, kmeans_init=True).to(dev).train()
    x = torch.randn(32, 8192, 512, device=dev)
    orig = L.assign
    k = [0]
    qs = torch.tensor([0.01, 0.5, 0.99], device=dev)

    def hooked(xx, packed, embed, **kw):
        dbg = k[0] < n_print
        L.screen_debug = dbg
        r = orig(xx, packed, embed, **kw)
        L.screen_debug = False
        if dbg and r.get("n_exact") is not None:
            n = r["idx"].numel()
            d = r["screen_debug"]
            e2 = (embed * embed).sum(-1)
            dup = int((e2 == 0).sum())
            xr = xx.reshape(-1, xx.shape[-1])
            print(f"call {k[0]:3d}: open {int(r['n_exact'][0]) / n:.4f} pair {int(r['n_pair'][0]) / n:.4f}  margin q01/50/99 "
                  f"{[round(float(v), 4) for v in torch.quantile((d[:100000, 0] - d[:100000, 1]), qs)]}  thr q01/50/99 "
                  f"{[round(float(v), 4) for v in torch.quantile(d[:100000, 2], qs)]}  |c|^2 max {float(e2.max()):.4g} min {float(e2.min()):.4g} zero codes {dup}  "
                  f"|x|^2 mean {float((xr[:4096] * xr[:4096]).sum(-1).mean()):.4g} x stride {tuple(xx.stride())} kw {sorted(kw)}", flush=True)
        k[0] += 1
        return r

    cbmod.L.assign = hooked
    with torch.no_grad():
        mod(x)
    torch.cuda.synchronize()
    print("searches through L.assign in the first forward:", k[0])


if __name__ == "__main__":
    main()
