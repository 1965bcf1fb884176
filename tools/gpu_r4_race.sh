#!/bin/bash
# round 4: the gradient pass-through race (ops._hold_for_side): regression test, the first-step test after two pool streams
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "lagging or first_step" 2>&1 | tail -15
  echo "--- dbg_first PRE=1"
  timeout 600 python tools/dbg_first.py 2>&1 | tail -8
  echo "--- conv + model suites in one process"
  timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -8
) > gpurun_out/r4_race.log 2>&1
cat gpurun_out/r4_race.log
