#!/bin/bash
out=gpurun_out/r5c; mkdir -p $out
pick() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); g=d.get('grad_step') or {}
        print(round(d['ms_per_step'],4), d.get('windows_ms_per_step'), 'grad', g.get('ms_per_step'))
PY
}
Q="--no-cpu-baseline --no-other-workloads --no-adversarial"
for b in 0 1 2; do for k in 1 3; do
  VQHIP_RVQ_BATCH_STATS=$b VQHIP_RVQ_CHUNKS=$k python bench.py $Q --workload rvq_cfg3 --steps 10 > $out/cfg3_b${b}_k$k.json 2>$out/cfg3_b${b}_k$k.err; echo "cfg3 batch=$b chunks=$k: $(pick $out/cfg3_b${b}_k$k.json) $(tail -1 $out/cfg3_b${b}_k$k.err)"
done; done
for b in 0 1 2; do
  VQHIP_RVQ_BATCH_STATS=$b python bench.py $Q --workload grvq_cfg5 --steps 5 > $out/cfg5_b$b.json 2>$out/cfg5_b$b.err; echo "cfg5 batch=$b: $(pick $out/cfg5_b$b.json) $(tail -1 $out/cfg5_b$b.err)"
done
VQHIP_RVQ_CHUNKS=1 python tools/timeline.py r5c/tl_cfg3_b2 --workload rvq_cfg3 --steps 4 --warmup 2 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads
