#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python tools/sweep_wgrad_addr.py 2>&1 | grep -v amdgpu.ids
  echo "--- wgrad tests"
  timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -x -q 2>&1 | tail -5
  echo "--- regression test of the pass-through race WITHOUT the hold (must fail)"
  SRHIP_HOLD=0 timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -k lagging 2>&1 | tail -6
) > gpurun_out/r4_wgaddr.log 2>&1
cat gpurun_out/r4_wgaddr.log
