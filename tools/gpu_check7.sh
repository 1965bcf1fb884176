#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "train_parity\[|passed|failed|Error|assert" gpurun_out/pytest_gpu.log | tail -20
python tools/sweep_conv.py > gpurun_out/sweep_conv.log 2>&1; cat gpurun_out/sweep_conv.log | tail -50
timeout 900 python bench.py --steps 6 --warmup 3 > gpurun_out/bench_graph.log 2>&1; echo "rc=$?" >> gpurun_out/bench_graph.log; tail -3 gpurun_out/bench_graph.log
