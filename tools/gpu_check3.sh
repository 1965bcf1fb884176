#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
python tools/bench_conv.py 32 > gpurun_out/bench_conv.log 2>&1; tail -12 gpurun_out/bench_conv.log
timeout 900 python -X faulthandler bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; echo "rc=$?" >> gpurun_out/bench_graph.log; tail -30 gpurun_out/bench_graph.log
