#!/bin/bash
set -x
O=gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/test.log
timeout 300 python bench.py --no-cpu-baseline --no-adversarial --no-other-workloads > $O/bench.json 2> $O/bench.err
timeout 200 python tools/time_heads.py > $O/time_heads.txt 2>&1
tail -6 $O/test.log; cat $O/time_heads.txt
