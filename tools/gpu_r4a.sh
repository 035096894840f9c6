#!/bin/bash
# round-4 first GPU call: full GPU test suite, the atomics micro-benchmark, baseline bench lines
set -x
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $O/test.log
hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o /tmp/atomic_bench tools/atomic_bench.hip > $O/atomic_build.log 2>&1
timeout 120 /tmp/atomic_bench > $O/atomic_bench.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --workload rvq_cfg3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --workload grvq_cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
tail -5 $O/test.log
