#!/bin/bash
# Everything the round's profiles/ directories hold, collected at ONE commit (on the GPU box): bash tools/collect_all.sh <round tag, e.g. r5>
R=${1:-r6}
Q="--steps 5 --warmup 2 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads"
python tools/collect_profile.py ${R}_final > gpurun_out/${R}_final.log 2>&1
COLLECT_NO_PMC=1 python tools/collect_profile.py ${R}_rvq_cfg3 --workload rvq_cfg3 $Q > gpurun_out/${R}_rvq_cfg3.log 2>&1
COLLECT_NO_PMC=1 python tools/collect_profile.py ${R}_grvq_cfg5 --workload grvq_cfg5 $Q > gpurun_out/${R}_grvq_cfg5.log 2>&1
COLLECT_NO_PMC=1 python tools/collect_profile.py ${R}_vq_cfg4_shard --workload vq_cfg4_shard $Q > gpurun_out/${R}_vq_cfg4_shard.log 2>&1
python tools/timeline.py ${R}_rvq_cfg3 --packs 1 --workload rvq_cfg3 --steps 6 --warmup 30 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads
python tools/timeline.py ${R}_grvq_cfg5 --workload grvq_cfg5 --steps 4 --warmup 40 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads
python tools/timeline.py ${R}_final --steps 6 --warmup 10 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads
for c in 0 1; do echo "VQHIP_SCRATCH_CACHE=$c"; VQHIP_SCRATCH_CACHE=$c python tools/scale_n.py 2>&1 | grep -v amdgpu; done > gpurun_out/${R}_final/scale_n_scratch_cache.txt
python bench.py > gpurun_out/${R}_final/bench_default.json 2> gpurun_out/${R}_final/bench_default.err
python tools/time_wide.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_final/time_wide.txt
git rev-parse HEAD > gpurun_out/${R}_final/commit.txt 2>/dev/null || true
tail -3 gpurun_out/${R}_final.log; tail -2 gpurun_out/${R}_rvq_cfg3.log | cut -c1-400; tail -2 gpurun_out/${R}_grvq_cfg5.log | cut -c1-400
