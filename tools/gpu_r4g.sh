#!/bin/bash
set -x
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/test.log
tail -8 $O/test.log
