#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -12 $O/t1.log
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-fp32-line > $O/bench_dist.log 2>&1; tail -1 $O/bench_dist.log | cut -c1-200
