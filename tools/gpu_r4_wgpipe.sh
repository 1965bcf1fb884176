#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python tools/sweep_wgrad_pipe.py 2>&1 | grep -v amdgpu.ids
  echo "--- passthrough debug (HOLD=0)"
  SRHIP_HOLD=0 timeout 300 python tools/dbg_passthrough.py 2>&1 | grep -v amdgpu.ids | tail -5
) > gpurun_out/r4_wgpipe.log 2>&1
cat gpurun_out/r4_wgpipe.log
