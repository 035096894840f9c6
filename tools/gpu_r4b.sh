#!/bin/bash
# round-4 second GPU call: per-kernel profile of the cfg-3 gradient step (regressed 3.6 -> 5.3 ms with the routed chain prologue),
# and the default bench line with its new other_workloads object
set -x
O=$PWD/gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_grad3 -o g3 -- python $GRAFT_REPO_ROOT/tools/grad_step.py rvq_cfg3 5 > $O/prof_grad3.log 2>&1)
find $O/prof_grad3 -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c "cut -c1-160 {} | head -25" > $O/grad3_kernel_stats.txt
find $O/prof_grad3 -name '*.db' -delete; find $O/prof_grad3 -name '*kernel_trace.csv' -delete
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
cat $O/grad3_kernel_stats.txt
