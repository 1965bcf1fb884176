#!/bin/bash
# round 4: same-box A/B of small step changes, alternating runs
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-fp32-line --no-sustained --steps 12 --warmup 4"
I="python bench.py --workload infer"
( for r in 1 2 3; do
    echo "train default (K-split 64-wide tile): $(timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "train SRHIP_DEBUG=10:0 (2x2 form)    : $(SRHIP_DEBUG=10:0 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "infer default                         : $(timeout 600 $I 2>&1 | tail -1 | cut -c90-160)"
    echo "infer SRHIP_DEBUG=10:0                : $(SRHIP_DEBUG=10:0 timeout 600 $I 2>&1 | tail -1 | cut -c90-160)"
  done ) > gpurun_out/r4_ab_ks.txt 2>&1
cat gpurun_out/r4_ab_ks.txt
