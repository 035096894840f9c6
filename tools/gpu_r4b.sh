#!/bin/bash
# per-kernel profile of the cfg-3 / cfg-5 gradient step (cfg 3 regressed 3.6 -> 5.3 ms with the routed chain prologue) and of the cfg-5 forward
set -x
O=$PWD/gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "pack_unpack" 2>&1 | tail -3 > $O/test_pack.log
prof() {  # tag, command...
  tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o t -- "$@" > $O/prof_$tag.log 2>&1)
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cut -c1-150 "$f" | head -30 > $O/${tag}_kernel_stats.txt
  rm -rf $O/prof_$tag
}
prof grad3 python $GRAFT_REPO_ROOT/tools/grad_step.py rvq_cfg3 5
prof grad5 python $GRAFT_REPO_ROOT/tools/grad_step.py grvq_cfg5 3
cat $O/test_pack.log; cat $O/grad3_kernel_stats.txt
