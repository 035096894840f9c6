cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
timeout 1200 python -m pytest tests -m gpu -x -q -k "route or grad or golden" 2>&1 | tail -3
for wl in rvq_cfg3 vq_cfg2; do
  rm -rf gpurun_out/r3b/prof_$wl
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/prof_$wl -o t -- python $GRAFT_REPO_ROOT/tools/grad_step.py $wl 5 > /dev/null 2>&1)
  f=$(find gpurun_out/r3b/prof_$wl -name "*kernel_stats.csv" | head -1)
  echo "== $wl"; head -6 $f | cut -c1-130
done
