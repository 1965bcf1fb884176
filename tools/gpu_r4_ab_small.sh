#!/bin/bash
# round 4: same-box A/B, alternating runs: persistent walk for launches with fewer tiles than block slots (key 11) against the one-tile kernels
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-fp32-line --no-sustained --steps 12 --warmup 4"
I="python bench.py --workload infer"
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -6
  for r in 1 2 3; do
    echo "train default (small grids persistent): $(timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "train SRHIP_DEBUG=11:0 (one-tile/K-split): $(SRHIP_DEBUG=11:0 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "infer default                          : $(timeout 600 $I 2>&1 | tail -1 | cut -c90-160)"
    echo "infer SRHIP_DEBUG=11:0                 : $(SRHIP_DEBUG=11:0 timeout 600 $I 2>&1 | tail -1 | cut -c90-160)"
  done ) > gpurun_out/r4_ab_small_grids.txt 2>&1
cat gpurun_out/r4_ab_small_grids.txt
