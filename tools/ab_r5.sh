#!/bin/bash
# round-5 A/B matrix: row chunks of the residual chain (cfg 3 / cfg 5) and the row pipeline of the fused step (cfg 2)
out=gpurun_out/r5a; mkdir -p $out
pick() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); g=d.get('grad_step') or {}
        print(round(d['ms_per_step'],4), d.get('windows_ms_per_step'), 'grad', g.get('ms_per_step'), 'search_ms', (d.get('roofline') or {}).get('kernel_ms'))
PY
}
Q="--no-cpu-baseline --no-other-workloads --no-adversarial"
for k in 1 2 3 4; do
  VQHIP_STEP_CHUNKS=$k python bench.py $Q --no-grad-step > $out/cfg2_k$k.json 2>$out/cfg2_k$k.err; echo "cfg2 step chunks=$k: $(pick $out/cfg2_k$k.json)"
done
for k in 1 2 3 4; do
  VQHIP_RVQ_CHUNKS=$k python bench.py $Q --workload rvq_cfg3 --steps 10 > $out/cfg3_k$k.json 2>$out/cfg3_k$k.err; echo "cfg3 rvq chunks=$k: $(pick $out/cfg3_k$k.json)"
done
for k in 0 1; do
  VQHIP_GRVQ_CHUNK_ROWS=$k python bench.py $Q --workload grvq_cfg5 --steps 5 --no-grad-step > $out/cfg5_c$k.json 2>$out/cfg5_c$k.err; echo "cfg5 chunk rows=$k: $(pick $out/cfg5_c$k.json)"
done
