#!/bin/bash
# round-4 third GPU call: step timelines (launch gaps) at cfg 2 / 3 / 5, the adversarial fuzz with the cyclic persistent kernel on
set -x
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/timeline.py r4c/tl2
COMMON="--steps 3 --warmup 2 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads"
timeout 200 python tools/timeline.py r4c/tl3 --workload rvq_cfg3 $COMMON
timeout 300 python tools/timeline.py r4c/tl5 --workload grvq_cfg5 $COMMON
VQHIP_SCREEN_PERSIST=2 timeout 400 python -m pytest tests/test_gpu_screen_fuzz.py -q -x -s 2>&1 | tail -12 > $O/fuzz_persist2.log
VQHIP_SCREEN_PERSIST=2 timeout 200 python bench.py --no-cpu-baseline --no-other-workloads > $O/bench_persist2.json 2> $O/bench_persist2.err
cat $O/tl2/timeline.txt; tail -3 $O/fuzz_persist2.log
