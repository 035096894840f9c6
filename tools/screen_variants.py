"""Dev tool: time vq_screen_kernel for every libvqhip_*.so in tools/variants (built with different -DVQS_* switches).
    python tools/screen_variants.py            # spawns one rocprofv3 run per variant, prints the kernel averages
    python tools/screen_variants.py --child    # (internal) the workload
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import torch
    sys.path.insert(0, ROOT)
    from vector_quantize_pytorch_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(0)
    N, C, D = 1 << 20, 1024, 256
    x = torch.randn(N, D, device="cuda", generator=g).to(torch.bfloat16)
    e = torch.empty(C, D, device="cuda")
    torch.nn.init.kaiming_uniform_(e, generator=g)
    packed = L.pack_codebook(e)
    for _ in range(12):
        r = L.assign(x, packed, e, want_q=True, want_sqerr=True)
    torch.cuda.synchronize()
    print("n_exact", int(r["n_exact"].item()), "n_pair", int(r["n_pair"].item()))


def main():
    out = os.path.join(ROOT, "gpurun_out", "variants")
    os.makedirs(out, exist_ok=True)
    for so in sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "libvqhip_*.so"))):
        tag = os.path.basename(so)[len("libvqhip_"):-3]
        d = os.path.join(out, tag)
        env = dict(os.environ, VQHIP_SO=so, TMPDIR="/tmp")
        p = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "v", "--",
                            sys.executable, os.path.abspath(__file__), "--child"], env=env, cwd="/tmp",
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        nex = [l for l in p.stdout.splitlines() if l.startswith("n_exact")]
        f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        line = f"{tag:10s} rc={p.returncode} {nex[0] if nex else ''} "
        tot = 0.0
        if f:
            for row in csv.DictReader(open(f[0])):
                if any(k in row["Name"] for k in ("vq_screen", "vq_assign", "vq_refine", "vq_pair", "vq_finish")):
                    tot += float(row["AverageNs"]) / 1e3
                    line += f"| {row['Name'].split('(')[0].split('vq_')[-1][:18]} avg {float(row['AverageNs']) / 1e3:.1f} min {float(row['MinNs']) / 1e3:.1f} "
            line += f"| SEARCH avg {tot:.1f} us"
        else:
            line += p.stdout[-400:]
        print(line, flush=True)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
