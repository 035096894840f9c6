"""Dev tool (GPU box): PMC counters of the screening kernel alone (VQHIP_SCREEN_ONLY=1), cfg-2 shape, one rocprofv3 pass per group.
    python tools/pmc_screen.py <out_dir> [env assignments, e.g. VQHIP_SCREEN_PERSIST=0]"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
          ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VMEM_RD"],
          ["SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_MEM_VIOLATIONS", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_FLAT"]]
BODY = """
import os, sys, torch
sys.path.insert(0, %r)
from vector_quantize_pytorch_amd import _lib as L
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(1 << 20, 256, device=dev, generator=g).bfloat16()
e = torch.empty(1024, 256, device=dev); torch.nn.init.kaiming_uniform_(e, generator=g)
pk = L.pack_codebook(e); q = torch.empty_like(x)
for _ in range(4): L.assign(x, pk, e, want_q=True, q_out=q)
torch.cuda.synchronize()
""" % ROOT
out = os.path.abspath(sys.argv[1])
env = dict(os.environ, TMPDIR="/tmp", VQHIP_SCREEN_ONLY="1")
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1); env[k] = v
os.makedirs(out, exist_ok=True)
open("/tmp/_pmc_body.py", "w").write(BODY)
summary = {}
for i, ctrs in enumerate(PASSES):
    d = os.path.join(out, f"pmc{i}")
    p = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, "/tmp/_pmc_body.py"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env, cwd="/tmp")
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        summary[f"pass{i}_error"] = p.stdout[-400:]; continue
    acc = {}
    for row in csv.DictReader(open(f[0])):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "screen" not in name: continue
        s = acc.setdefault((name, row["Counter_Name"]), {})
        s[row["Dispatch_Id"]] = s.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    for (name, ctr), per in acc.items():
        vals = list(per.values())
        summary.setdefault(name, {})[ctr] = sum(vals) / len(vals)
    shutil.rmtree(d, ignore_errors=True)
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
