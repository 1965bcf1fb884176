#!/bin/bash
mkdir -p gpurun_out
python tools/debug_wdiff.py > gpurun_out/debug_wdiff.log 2>&1; tail -40 gpurun_out/debug_wdiff.log
timeout 600 python -X faulthandler bench.py --steps 2 --warmup 3 --batch 4 --no-cpu-baseline > gpurun_out/bench_graph_b4.log 2>&1; echo "rc=$?" >> gpurun_out/bench_graph_b4.log; tail -30 gpurun_out/bench_graph_b4.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_eager -o eager -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_eager.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_eager | head; tail -3 gpurun_out/prof_eager.log
