"""round 6: k-means iteration 0 (centroids = sampled data rows, vqp.py:238-256) sends 98 % of the rows of cfg 5 to the exact
sweep.  Dumps the screening kernel's per-row (best, second, threshold, class) for that search and for iteration 1's, with the
codebook-wide scalars the threshold is built from.

    python tools/kmeans_iter0_debug.py [N] [D] [C]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vector_quantize_pytorch_amd import _lib as L


def show(tag, x, m):
    packed = L.pack_codebook(m)
    C, D = m.shape
    off = L.lib().vqhip_packed_bytes(C, D)
    L.screen_debug = True
    r = L.assign(x, packed, m, want_q=False)
    L.screen_debug = False
    d = r["screen_debug"]
    margin = d[:, 0] - d[:, 1]
    thr = d[:, 2]
    cls = d[:, 3]
    qs = torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99], device=x.device)
    print(f"{tag}: open {float((cls == 1).float().mean()):.4f} pair {float((cls == 2).float().mean()):.4f}  "
          f"margin q01/10/50/90/99 {[round(float(v), 4) for v in torch.quantile(margin[:100000], qs)]}  "
          f"thr q01/50/99 {[round(float(v), 4) for v in torch.quantile(thr[:100000], qs[[0, 2, 4]])]}  "
          f"|c|^2 max {float((m * m).sum(-1).max()):.3f} min {float((m * m).sum(-1).min()):.3f}  |x|^2 mean {float((x * x).sum(-1).mean()):.3f}",
          flush=True)
    return r["idx"]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    x = torch.randn(N, D, device=dev)
    m = x[torch.randperm(N, device=dev)[:C]].contiguous()
    for it in range(3):
        idx = show(f"iteration {it}", x, m)
        cnt, esum = L.ema_accumulate(x, idx, C)
        L.kmeans_update(m, esum, cnt)
    # the same rows as a strided feature chunk of a wider tensor (GroupedResidualVQ's groups)
    xw = torch.randn(N, 4 * D, device=dev)
    xs = xw[:, D:2 * D]
    m = xs[torch.randperm(N, device=dev)[:C]].contiguous()
    show("strided chunk, iteration 0", xs, m)


if __name__ == "__main__":
    main()
