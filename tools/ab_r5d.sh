#!/bin/bash
A="--workload rvq_cfg3 --steps 6 --warmup 30 --windows 1 --no-grad-step --no-cpu-baseline --no-adversarial --no-other-workloads"
VQHIP_RVQ_BATCH_STATS=2 VQHIP_RVQ_CHUNKS=1 python tools/timeline.py r5d/tl_cfg3_b2 $A
VQHIP_RVQ_BATCH_STATS=0 VQHIP_RVQ_CHUNKS=1 python tools/timeline.py r5d/tl_cfg3_b0 $A
VQHIP_RVQ_BATCH_STATS=0 VQHIP_RVQ_CHUNKS=3 python tools/timeline.py r5d/tl_cfg3_b0k3 $A
