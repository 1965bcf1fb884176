#!/bin/bash
# experiment: 3 / 4 RAB weight gradients per flat-kernel launch (fewer split-K partials) against pairs
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6h; mkdir -p $E; cd $R
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2; do
  timeout 300 $B 2>&1 | tail -1 | cut -c1-140
  SRHIP_PP_GROUP=3 SRHIP_WGRAD_MAX_AGE=14 timeout 300 $B 2>&1 | tail -1 | cut -c1-140
  SRHIP_PP_GROUP=4 SRHIP_WGRAD_MAX_AGE=18 timeout 300 $B 2>&1 | tail -1 | cut -c1-140
done
