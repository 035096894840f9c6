"""Per-kernel register / scratch use of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kres.py vector_quantize_pytorch_amd/csrc/vq_screen.hip [filter] [-- extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-c", "-o", "/dev/null", src,
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, r in rows.items():
    if flt in name:
        g = lambda k: str(r.get(k, "?"))
        print(f"{name[:110]:110s} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>4} spill {g('VGPRs Spill'):>4} scratch {g('ScratchSize'):>5} occ {g('Occupancy')}")
