#!/bin/bash
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4i
mkdir -p $E
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_half_mode_gpu.py -x -q -m gpu > $E/tests.log 2>&1; tail -3 $E/tests.log
VARS=-1,0 timeout 600 python tools/sweep_pers.py > $E/sweep.log 2>&1; grep median $E/sweep.log
MODE=half VARS=-1,0 timeout 600 python tools/sweep_pers.py > $E/sweep_half.log 2>&1; grep median $E/sweep_half.log
VARS=0,64,-1,0,64,-1 timeout 600 python tools/ablate_pers.py > $E/ablate.log 2>&1; cat $E/ablate.log
bash tools/gpu_roofline2.sh > $E/roof2.log 2>&1; tail -12 $E/roof2.log
