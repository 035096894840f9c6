"""Folds the traffic measurements of one round's profile directories into profiles/traffic.json (what bench.py reads for
`roofline.traffic`): the per-step HBM bytes of the three workloads (traffic_step.json of tools/collect_profile.py) and the bytes per
launch of the cfg-2 search kernels (FETCH_SIZE x 2 + WRITE_SIZE of pmc_summary.json, KiB -> bytes).
    python tools/update_traffic.py r5 <commit>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, commit = sys.argv[1], sys.argv[2]
tp = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tp))
for wl, d in (("vq_cfg2", f"{R}_final"), ("rvq_cfg3", f"{R}_rvq_cfg3"), ("grvq_cfg5", f"{R}_grvq_cfg5"), ("vq_cfg4_shard", f"{R}_vq_cfg4_shard")):
    fp = os.path.join(ROOT, "profiles", d, "traffic_step.json")
    if not os.path.exists(fp):
        continue
    t = json.load(open(fp))
    tj["step_traffic"][wl] = {"bytes_per_step": t["bytes_per_step"], "fetch_bytes_per_step": t["fetch_bytes_per_step"],
                              "write_bytes_per_step": t["write_bytes_per_step"], "measured_at_commit": commit,
                              "source": f"profiles/{d}/traffic_step.json (tools/collect_profile.py: all vq_* kernels of one step, difference of two rocprofv3 "
                                        f"--pmc FETCH_SIZE / WRITE_SIZE runs, round {R[1:]}; not re-measured in this run)"}
pm = json.load(open(os.path.join(ROOT, "profiles", f"{R}_final", "pmc_summary.json")))
kern = [k for k in pm if isinstance(pm[k], dict) and any(k.startswith(p) for p in ("vq_screenc_kernel", "vq_screen16_kernel", "vq_compact_lists_kernel", "vq_refine_kernel",
                                                                                   "vq_pair_kernel", "vq_finish_listed_kernel"))]
fetch = sum(pm[k]["FETCH_SIZE"]["mean"] for k in kern) * 1024.0
write = sum(pm[k]["WRITE_SIZE"]["mean"] for k in kern) * 1024.0
tj["assign_screened_cfg2_bytes_per_launch"] = 2.0 * fetch + write
tj["assign_screened_cfg2"] = {"kernels": kern, "fetch_size_reported_bytes": fetch, "write_size_reported_bytes": write, "algorithmic_bytes": 1083179008,
                              "measured_at_commit": commit, "source": f"profiles/{R}_final/pmc_summary.json",
                              "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python bench.py --steps 5 --warmup 2 ...` "
                                        f"(tools/collect_profile.py {R}_final), mean per launch summed over the kernels of one screened search; KiB -> bytes; "
                                        "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md 'HBM'; calibration in this file)"}
json.dump(tj, open(tp, "w"), indent=1)
print("search bytes per launch", tj["assign_screened_cfg2_bytes_per_launch"], {k: v["bytes_per_step"] for k, v in tj["step_traffic"].items()})
