#!/bin/bash
# three RAB weight gradients per flat launch as the default: full GPU suite, A/B against pairs, timeline
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6i; mkdir -p $E; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; tail -3 $E/pytest_gpu.log
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2 3; do
  timeout 300 $B 2>&1 | tail -1 > $E/g3_$i.json; cut -c1-140 $E/g3_$i.json
  SRHIP_PP_GROUP=2 SRHIP_WGRAD_MAX_AGE=8 timeout 300 $B 2>&1 | tail -1 > $E/g2_$i.json; cut -c1-140 $E/g2_$i.json
done
timeout 300 python tools/step_timeline.py 2>&1 | tail -12
