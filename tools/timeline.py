"""Timeline of ONE steady-state step from a rocprofv3 kernel trace: every launch with its stream, start offset, duration and the idle gap
on its stream since the previous kernel -- where the launch gaps and the small kernels of a step sit.
    python tools/timeline.py <tag> [bench.py args...]        (on the GPU box; writes gpurun_out/<tag>/timeline.txt)
A step starts at a `vq_pack_kernel` launch whose predecessor in time is not a pack kernel (the residual modules pack all their
codebooks first); the LAST complete step of the run is printed."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    packs_per_step = None
    if "--packs" in sys.argv:                  # pack launches per step (codebooks of the module): steps are then cut by COUNT from the end
        i = sys.argv.index("--packs")
        packs_per_step = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    bargs = sys.argv[2:] or ["--steps", "4", "--warmup", "2", "--windows", "1", "--no-grad-step", "--no-cpu-baseline", "--no-adversarial",
                             "--no-other-workloads"]
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    d = os.path.join(out, "trace")
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
                        os.path.join(ROOT, "bench.py")] + bargs, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not f:
        open(os.path.join(out, "timeline.txt"), "w").write("no trace\n" + p.stdout[-3000:])
        return
    rows = []
    for r in csv.DictReader(open(f[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    short = lambda n: n.replace("void ", "").split("(")[0][:70]
    # step length from the profiled run's own bench line; a pack kernel that starts more than half a step after the previous pack
    # kernel opens a step (modules with several codebooks pack all of them at the beginning of their forward)
    step_ms = None
    for l in p.stdout.splitlines():
        if l.startswith("{"):
            try:
                step_ms = json.loads(l)["ms_per_step"]
            except Exception:
                pass
    packs = [i for i, r in enumerate(rows) if short(r[2]).startswith("vq_pack_kernel")]
    thr = 0.5 * (step_ms or 0.5) * 1e6
    starts = [i for k, i in enumerate(packs) if k == 0 or rows[i][0] - rows[packs[k - 1]][0] > thr]
    if packs_per_step:                         # (several streams: the packs of a step interleave with the previous step's tail)
        starts = [packs[k] for k in range(len(packs) % packs_per_step, len(packs), packs_per_step)]
    lines = []
    if len(starts) >= 3:
        a, b = starts[-3], starts[-2]          # (the very last step may be followed by the audit kernels)
        t0 = rows[a][0]
        last_end = {}
        lines.append(f"step of {(rows[b][0] - t0) / 1000:.1f} us, {b - a} launches\n")
        lines.append(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  q   kernel\n")
        busy = 0
        for s, e, n, q, st in rows[a:b]:
            gap = (s - last_end[q]) / 1000 if q in last_end else 0.0
            last_end[q] = e
            busy += e - s
            if b - a <= 400:
                lines.append(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:8.1f} {gap:7.1f}  {q:>3} {short(n)}\n")
        lines.append(f"sum of kernel durations {busy / 1000:.1f} us\n\nper kernel in this step:\n")
        per = {}
        for s_, e_, n, q, st in rows[a:b]:
            k = short(n)
            per.setdefault(k, [0, 0])
            per[k][0] += 1; per[k][1] += e_ - s_
        for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{t / 1000:10.1f} us {100.0 * t / busy:5.1f} %  x{c:<4d} {k}\n")
    else:
        lines.append(f"only {len(starts)} step starts found among {len(rows)} launches\n")
    open(os.path.join(out, "timeline.txt"), "w").writelines(lines)
    for g in glob.glob(os.path.join(d, "**", "*"), recursive=True):
        if os.path.isfile(g):
            os.remove(g)


if __name__ == "__main__":
    main()
