#!/bin/bash
# round-4 GPU call: the fused train step (vqhip_vq_train_step) -- module tests, bench lines with and without it, timeline
set -x
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -25 > $O/test.log
timeout 200 python bench.py --no-cpu-baseline --no-other-workloads > $O/bench_fused.json 2> $O/bench_fused.err
VQHIP_FUSED_STEP=0 timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --no-grad-step --no-adversarial > $O/bench_unfused.json 2> $O/bench_unfused.err
VQHIP_SCREEN_PERSIST=2 timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --no-grad-step --no-adversarial > $O/bench_fused_p2.json 2> $O/bench_fused_p2.err
timeout 200 python tools/timeline.py r4d/tl2
tail -5 $O/test.log; cat $O/tl2/timeline.txt
