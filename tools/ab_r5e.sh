#!/bin/bash
out=gpurun_out/r5e; mkdir -p $out
pick() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); g=d.get('grad_step') or {}
        print(round(d['ms_per_step'],4), d.get('windows_ms_per_step'), 'grad', g.get('ms_per_step'))
PY
}
Q="--no-cpu-baseline --no-other-workloads --no-adversarial --no-grad-step"
for s in 0 20 40 60 80 120; do for k in 1 3; do
  VQHIP_SCREEN_STAGGER=$s VQHIP_RVQ_CHUNKS=$k python bench.py $Q --workload rvq_cfg3 --steps 10 > $out/cfg3_s${s}_k$k.json 2>$out/err; echo "cfg3 stagger=$s chunks=$k: $(pick $out/cfg3_s${s}_k$k.json)"
done; done
for s in 0 40; do
  VQHIP_CHAIN_NOWRITE=1 VQHIP_SCREEN_STAGGER=$s VQHIP_RVQ_CHUNKS=1 python bench.py $Q --workload rvq_cfg3 --steps 10 > $out/cfg3_nw_s${s}.json 2>$out/err; echo "cfg3 NOWRITE stagger=$s chunks=1: $(pick $out/cfg3_nw_s${s}.json)"
done
for s in 0 40 80; do
  VQHIP_SCREEN_STAGGER=$s python bench.py $Q --workload grvq_cfg5 --steps 5 > $out/cfg5_s$s.json 2>$out/err; echo "cfg5 stagger=$s: $(pick $out/cfg5_s$s.json)"
done
