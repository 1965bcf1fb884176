#!/bin/bash
# round 4: BatchNorm gradients accumulated by the kernel (ABI 7) + unpool outputs allocated channels-last: tests, bench A/B is the suite run before
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
  for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 | cut -c1-200; done
  timeout 600 python tools/prof_adds.py 2>&1 | grep -v amdgpu.ids | grep "kernel time per step" ) > gpurun_out/r4_check2.log 2>&1
cat gpurun_out/r4_check2.log
