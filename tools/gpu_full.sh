#!/bin/bash
# Full GPU validation: parity tests, bench (eager default), rocprof kernel stats of the same command
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error|assert" gpurun_out/pytest_gpu.log | tail -8
timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/bench.log 2>&1; echo "rc=$?" >> gpurun_out/bench.log; tail -2 gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_full -o pf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_full.log 2>&1
