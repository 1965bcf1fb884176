#!/bin/bash
# full GPU suite + default bench line (round 4 checkpoints) + the ATen kernels the step still launches
mkdir -p gpurun_out
( T0=$(date +%s); timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8; echo "suite wall: $(( $(date +%s) - T0 )) s"
  timeout 900 python bench.py --no-cpu-baseline --no-fp32-line 2>&1 | tail -1 | cut -c1-400
  timeout 600 python tools/prof_adds.py 2>&1 | grep -v amdgpu.ids | tail -50 ) > gpurun_out/r4_suite.log 2>&1
cat gpurun_out/r4_suite.log | cut -c1-220
