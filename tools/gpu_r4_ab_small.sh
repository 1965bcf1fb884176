#!/bin/bash
# round 4: same-box A/B, alternating runs: block target of the row-tap weight gradient inside the step (srhip_debug_set(1, target >= 100))
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-fp32-line --no-sustained --steps 12 --warmup 4"
( for r in 1 2; do
    for T in 0 512 640 1024; do
      echo "wgrad block target $T: $(SRHIP_DEBUG=1:$T timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    done
  done ) > gpurun_out/r4_ab_wgrad_target.txt 2>&1
cat gpurun_out/r4_ab_wgrad_target.txt
