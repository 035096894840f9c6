#!/bin/bash
# A/B matrix of the residual workloads' host-side switches (round 5; results: profiles/r5_rvq_cfg3/chunks.txt).
#   VQHIP_RVQ_CHUNKS=k        interleaved row chunks of the residual chain (default: 3 chunks of ~87k rows at cfg 3)
#   VQHIP_RVQ_BATCH_STATS=b   0 per-stage statistics beside the loop (default) | 1 stage 0 beside, rest batched | 2 all behind the loop
#   VQHIP_STEP_CHUNKS=k       row pipeline inside vqhip_vq_train_step (default 1: it loses, profiles/r5_step_pipeline)
# usage (on the GPU box): bash tools/ab_rvq.sh
out=gpurun_out/ab_rvq; mkdir -p $out
pick() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); g=d.get('grad_step') or {}
        print(round(d['ms_per_step'],4), d.get('windows_ms_per_step'), 'grad', g.get('ms_per_step'))
PY
}
Q="--no-cpu-baseline --no-other-workloads --no-adversarial"
for b in 0 2; do for k in 1 2 3; do
  VQHIP_RVQ_BATCH_STATS=$b VQHIP_RVQ_CHUNKS=$k python bench.py $Q --workload rvq_cfg3 --steps 10 > $out/cfg3_b${b}_k$k.json 2>$out/err; echo "cfg3 batch=$b chunks=$k: $(pick $out/cfg3_b${b}_k$k.json)"
done; done
for k in 1 2; do
  VQHIP_STEP_CHUNKS=$k python bench.py $Q --no-grad-step > $out/cfg2_k$k.json 2>$out/err; echo "cfg2 step chunks=$k: $(pick $out/cfg2_k$k.json)"
done
