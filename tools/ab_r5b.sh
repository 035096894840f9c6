#!/bin/bash
out=gpurun_out/r5b; mkdir -p $out
pick() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); g=d.get('grad_step') or {}
        print(round(d['ms_per_step'],4), d.get('windows_ms_per_step'), 'grad', g.get('ms_per_step'), 'search_ms', (d.get('roofline') or {}).get('kernel_ms'))
PY
}
Q="--no-cpu-baseline --no-other-workloads --no-adversarial"
for k in 1 2 3; do
  VQHIP_SCREEN_PERSIST=0 VQHIP_STEP_CHUNKS=$k python bench.py $Q --no-grad-step > $out/cfg2_np_k$k.json 2>$out/cfg2_np_k$k.err; echo "cfg2 PERSIST=0 step chunks=$k: $(pick $out/cfg2_np_k$k.json)"
done
for k in 1 2 3; do
  VQHIP_RVQ_CHUNKS=$k python tools/timeline.py r5b/tl_cfg3_k$k --workload rvq_cfg3 --steps 4 --warmup 2 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads
  head -3 gpurun_out/r5b/tl_cfg3_k$k/timeline.txt
done
VQHIP_STEP_CHUNKS=2 python tools/timeline.py r5b/tl_cfg2_k2 --steps 4 --warmup 2 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads
cat gpurun_out/r5b/tl_cfg2_k2/timeline.txt | head -50
