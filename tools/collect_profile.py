"""Collects the rocprofv3 evidence for one bench.py configuration into gpurun_out/<tag>/ (copy it to profiles/<tag>/):
    kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
    pmc_summary.json   mean counter value per launch and kernel, one rocprofv3 --pmc pass per counter group
    bench.json         the bench.py line of an un-profiled run on the same box
Run on the GPU box:  python tools/collect_profile.py <tag> [bench.py args...]
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
              ["SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"],
              ["SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_F32"],
              ["SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"]]


def run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, **kw)


def main():
    tag = sys.argv[1]
    bargs = sys.argv[2:] or ["--steps", "5", "--warmup", "2", "--windows", "1", "--no-grad-step", "--no-cpu-baseline", "--no-adversarial",
                             "--no-other-workloads"]
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    bench = [sys.executable, os.path.join(ROOT, "bench.py")] + bargs
    env = dict(os.environ, TMPDIR="/tmp")

    wl = ["--workload", bargs[bargs.index("--workload") + 1]] if "--workload" in bargs else []
    p = run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-workloads", *wl], env=env, cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    open(os.path.join(out, "bench.json"), "w").write((line[-1] if line else p.stdout[-2000:]) + "\n")

    d = os.path.join(out, "trace")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "t", "--"] + bench, env=env, cwd="/tmp")
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(out, "kernel_stats.csv"))

    summary = {}
    for i, ctrs in enumerate([] if os.environ.get("COLLECT_NO_PMC") == "1" else PMC_PASSES):     # (COLLECT_NO_PMC=1: stats + step traffic only)
        d = os.path.join(out, f"pmc{i}")
        p = run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + bench,
                env=env, cwd="/tmp")
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not f:
            summary[f"pass{i}_error"] = p.stdout[-500:]
            continue
        acc = {}
        for row in csv.DictReader(open(f[0])):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if not name.startswith("vq_"):
                continue
            k = (name, row["Counter_Name"])
            s = acc.setdefault(k, [0.0, {}])
            # one row per (dispatch, counter [, dimension]); sum the dimensions of a dispatch
            s[1][row["Dispatch_Id"]] = s[1].get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        for (name, ctr), (_, per) in acc.items():
            vals = list(per.values())
            summary.setdefault(name, {})[ctr] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
    json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    # HBM / fabric bytes of ONE STEP, all vq_* kernels together, by DIFFERENCE of two runs that differ only in the number of timed
    # steps (so first-forward work -- k-means, packing -- cancels).  FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is
    # doubled (gfx950 counts the 128-byte requests of wide coalesced reads as 64 bytes: MI355X_MICROARCH.md "HBM", calibration in
    # profiles/traffic.json).
    def with_steps(n):
        b = [a for a in bargs]
        for flag, val in (("--steps", str(n)), ("--windows", "1")):
            if flag in b:
                b[b.index(flag) + 1] = val
            else:
                b += [flag, val]
        return b

    def total_kib(ctr, n):
        d = os.path.join(out, f"traffic_{ctr}_{n}")
        run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--",
             sys.executable, os.path.join(ROOT, "bench.py")] + with_steps(n), env=env, cwd="/tmp")
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        tot = 0.0
        for row in csv.DictReader(open(f[0])) if f else []:
            if row["Kernel_Name"].replace("void ", "").startswith("vq_") and row["Counter_Name"] == ctr:
                tot += float(row["Counter_Value"])
        shutil.rmtree(d, ignore_errors=True)
        return tot

    s_lo, s_hi = 4, 14
    fetch = (total_kib("FETCH_SIZE", s_hi) - total_kib("FETCH_SIZE", s_lo)) / (s_hi - s_lo)
    write = (total_kib("WRITE_SIZE", s_hi) - total_kib("WRITE_SIZE", s_lo)) / (s_hi - s_lo)
    entry = {"fetch_bytes_per_step": 2.0 * 1024.0 * fetch, "write_bytes_per_step": 1024.0 * write,
             "bytes_per_step": 2.0 * 1024.0 * fetch + 1024.0 * write, "bench_args": bargs,
             "method": f"all vq_* kernels; (run with {s_hi} timed steps - run with {s_lo}) / {s_hi - s_lo}; rocprofv3 --kernel-trace --pmc "
                       "FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; FETCH x 2"}
    json.dump(entry, open(os.path.join(out, "traffic_step.json"), "w"), indent=1)
    print(json.dumps(entry))
    for sub in ["trace"] + [f"pmc{i}" for i in range(len(PMC_PASSES))]:
        shutil.rmtree(os.path.join(out, sub), ignore_errors=True)
    print(open(os.path.join(out, "bench.json")).read())
    if os.path.exists(os.path.join(out, "kernel_stats.csv")):
        print("".join(open(os.path.join(out, "kernel_stats.csv")).readlines()[:10]))
    print(json.dumps({k: v for k, v in summary.items() if "screen" in k or "assign" in k or "error" in k}, indent=1)[:3000])


if __name__ == "__main__":
    main()
