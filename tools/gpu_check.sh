#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, conv micro-bench, bench.py eager+graph.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python tools/bench_conv.py 32 > gpurun_out/bench_conv.log 2>&1; tail -12 gpurun_out/bench_conv.log
timeout 900 python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/bench_eager.log 2>&1; tail -3 gpurun_out/bench_eager.log
timeout 900 python bench.py --steps 5 --warmup 3 > gpurun_out/bench_graph.log 2>&1; tail -3 gpurun_out/bench_graph.log
