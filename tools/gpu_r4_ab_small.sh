#!/bin/bash
# round 4: same-box A/B of small step changes, alternating runs
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-fp32-line --no-sustained --steps 12 --warmup 4"
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parity_configs_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
  for r in 1 2 3; do
    echo "default (conv7 bwd one launch): $(timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "SRHIP_DEBUG=7:16 (two launches): $(SRHIP_DEBUG=7:16 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "SRHIP_RUN_AHEAD=1              : $(SRHIP_RUN_AHEAD=1 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "SRHIP_RUN_AHEAD=3              : $(SRHIP_RUN_AHEAD=3 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
  done ) > gpurun_out/r4_ab_small2.txt 2>&1
cat gpurun_out/r4_ab_small2.txt
