#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "train_parity\[|passed|failed|Error|assert" gpurun_out/pytest_gpu.log | tail -12
python tools/sweep_wgrad.py > gpurun_out/sweep_wgrad.log 2>&1; tail -30 gpurun_out/sweep_wgrad.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o p4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof4.log 2>&1
