#!/bin/bash
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4c
mkdir -p $E
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > $E/tests.log 2>&1; tail -3 $E/tests.log
VARS=-1,0 timeout 600 python tools/sweep_pers.py > $E/sweep.log 2>&1; grep median $E/sweep.log
VARS=0,-1,16,2,32,0 timeout 600 python tools/ablate_pers.py > $E/ablate.log 2>&1; cat $E/ablate.log
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line 2>&1 | tail -1 > $E/bench_new.json; cut -c1-200 $E/bench_new.json
