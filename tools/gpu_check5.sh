#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "train_parity|passed|failed|Error|assert" gpurun_out/pytest_gpu.log | tail -30
python tools/debug_graph.py 32 8 > gpurun_out/debug_graph32.log 2>&1; tail -17 gpurun_out/debug_graph32.log
timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; echo "rc=$?" >> gpurun_out/bench_graph.log; tail -3 gpurun_out/bench_graph.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
