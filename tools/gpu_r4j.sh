#!/bin/bash
set -x
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/test.log
timeout 300 python bench.py --no-cpu-baseline --no-grad-step --no-adversarial --no-other-workloads > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --workload rvq_cfg3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
tail -12 $O/test.log
