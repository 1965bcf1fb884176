#!/bin/bash
# round 4: same-box A/B of the small step changes, alternating runs: bus sum in one pass; gradient hold off / aliased only (default) / all
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-fp32-line --no-sustained --steps 12 --warmup 4"
( for r in 1 2 3; do
    echo "default (hold aliased) : $(timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "SRHIP_BUS_SUM=0        : $(SRHIP_BUS_SUM=0 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "SRHIP_HOLD=0 (racy)    : $(SRHIP_HOLD=0 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
    echo "SRHIP_HOLD=2 (hold all): $(SRHIP_HOLD=2 timeout 600 $B 2>&1 | tail -1 | cut -c58-130)"
  done
  echo "--- race checks with the default hold"
  timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "lagging or first_step" 2>&1 | grep -E "passed|failed"
  timeout 600 python tools/dbg_first.py 2>&1 | grep -v amdgpu.ids | tail -3
) > gpurun_out/r4_ab_small.txt 2>&1
cat gpurun_out/r4_ab_small.txt
