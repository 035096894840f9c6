"""Turns the same-box A/B runs of a round (gpurun_out/<tag><letter>/: bench.py JSON lines, one per variant and repeat; rocprofv3
databases) into the committed record profiles/<tag>_ab/summary.md: one table per run directory -- variant, ms per step, search ms, and,
where a database is present, the per-kernel averages.   python tools/ab_summary.py r6"""
import glob, json, os, re, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
NOTES = {}
notes_file = os.path.join(ROOT, "profiles", f"{tag}_ab", "notes.json")       # {"r6n": "what the run compared", ...}
if os.path.exists(notes_file):
    NOTES = json.load(open(notes_file))
out = [f"# Same-box A/B runs of round {tag[1:]} (`tools/ab_summary.py {tag}`)\n",
       "Every table is ONE `gpurun` call (one box, variants interleaved); boxes differ by up to 5 %, so only rows of one table compare.\n"
       "`now` = the tree at the time of the run, other names = a library built from a variant (`VQHIP_SO`) or an environment switch.\n"]
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "[a-z]*"))):
    name = os.path.basename(d)
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "*.json"))):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        if "windows_ms" in j:        # tools/cfg5_diag.py: per-window ms of the grouped module in one process context
            w = j["windows_ms"]
            diag = f"{j.get('mode', '')} context, warm-up {j.get('warmup')}: windows {min(w):.2f} .. {max(w):.2f} ms (median {sorted(w)[len(w) // 2]:.2f}), first forward {j.get('first_forward_ms')} ms"
            rows.append((os.path.basename(f)[:-5], sorted(w)[len(w) // 2], None, diag, None, None, None))
            continue
        if "ms_per_step" not in j:
            continue
        r = j.get("roofline") or {}
        rows.append((os.path.basename(f)[:-5], j["ms_per_step"], r.get("kernel_ms"), (j.get("config") or {}).get("workload", "")[:48],
                     r.get("rows_exact_pass_frac"), r.get("rows_pair_pass_frac"), j.get("screen_stress")))
    dbs = sorted(glob.glob(os.path.join(d, "*", "*.db")) + glob.glob(os.path.join(d, "*.db")))
    txts = [t for t in sorted(glob.glob(os.path.join(d, "*.txt"))) if os.path.getsize(t) < 6000]
    if not rows and not dbs and not txts:
        continue
    out.append(f"\n## {name}" + (f" — {NOTES[name]}" if name in NOTES else "") + "\n")
    if rows:
        out.append("| run | ms / step | search ms | open rows | pair rows | workload |\n|---|---|---|---|---|---|")
        for n, ms, k, w, fo, fp, _ in rows:
            fmt = lambda v, p: "" if v is None else (p % v)
            out.append(f"| {n} | {ms:.4f} | {fmt(k, '%.4f')} | {fmt(fo, '%.5f')} | {fmt(fp, '%.5f')} | {w} |")
        for n, *_, st in rows:
            if st:
                out.append(f"\n`{n}` stress legs: " + "; ".join(
                    f"{k} {v['ms_per_step']} ms (open {v.get('open_frac')}, pair {v.get('pair_frac')}" +
                    (f", vs control {v['vs_control']}" if v.get("vs_control") else "") + ")" for k, v in st.items()))
    for db in dbs:
        con = sqlite3.connect(db)
        out.append(f"\n`{os.path.relpath(db, d)}` (rocprofv3 --kernel-trace, average µs per launch, vq kernels):\n")
        out.append("| kernel | launches | avg µs |\n|---|---|---|")
        q = ("select name, count(*), avg(end-start)/1e3 from kernels where name like '%vq_%' and name not like '%at::%' group by name order by sum(end-start) desc limit 16")
        for nm, n, avg in con.execute(q):
            out.append(f"| `{re.sub(r'[(].*', '', nm)[:70]}` | {n} | {avg:.2f} |")
    for t in txts:
        body = open(t, errors="replace").read().strip().splitlines()
        body = [l for l in body if "amdgpu.ids" not in l][-12:]
        if body:
            out.append(f"\n`{os.path.basename(t)}`:\n```\n" + "\n".join(l[:220] for l in body) + "\n```")
open(os.path.join(ROOT, "profiles", f"{tag}_ab", "summary.md"), "w").write("\n".join(out) + "\n")
print("wrote", len(out), "lines")
