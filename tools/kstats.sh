#!/bin/bash
# usage: tools/kstats.sh <tag> [VQHIP_SO|""] [bench.py args...]   -- rocprofv3 kernel stats of a short bench.py run, top kernels printed
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
[ -n "$2" ] && export VQHIP_SO=$2
TAG=$1; shift; shift
set -- "$TAG" "$@"
EXTRA="${@:2}"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$1 -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > $R/gpurun_out/ks_$1.json 2>/dev/null
cd $R
echo "== $1: $(python -c "import json,sys; d=json.loads(open('gpurun_out/ks_$1.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4))")"
f=$(find gpurun_out/ks_$1 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for i,row in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 14: print("   %-60s avg %8.1f us  min %8.1f" % (row["Name"].split("(")[0][-60:], float(row["AverageNs"])/1e3, float(row["MinNs"])/1e3))
PY
