#!/bin/bash
# round-4 GPU call: full GPU suite after the spill fixes / routed-residual kernel / persistent kernel as default / score-row slices; default bench line
set -x
O=gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/test.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python tools/scale_n.py > $O/scale_n.txt 2>&1
VQHIP_FUSED_STEP=0 timeout 200 python tools/scale_n.py > $O/scale_n_unfused.txt 2>&1
tail -5 $O/test.log
