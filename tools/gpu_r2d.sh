#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_half_mode_gpu.py tests/test_conv_gpu.py -x -q -m gpu -s > $O/t1.log 2>&1; echo "t1 rc=$?"; grep -E "half mode|passed|failed|Error" $O/t1.log | tail -20
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("bf16x3", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"])'
timeout 600 python bench.py --no-cpu-baseline --conv-math half > $O/bench_half.log 2>&1; tail -1 $O/bench_half.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("half", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], d["last_losses"])'
timeout 900 python bench.py --workload chain --steps 4 --warmup 2 > $O/chain.log 2>&1; tail -1 $O/chain.log | cut -c1-1500
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_half_mode_gpu.py --deselect tests/test_conv_gpu.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
