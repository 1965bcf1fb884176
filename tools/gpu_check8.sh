#!/bin/bash
mkdir -p gpurun_out
python tools/debug_nan.py a > gpurun_out/nan_a.log 2>&1; tail -3 gpurun_out/nan_a.log
python tools/debug_nan.py b > gpurun_out/nan_b.log 2>&1; tail -3 gpurun_out/nan_b.log
python tools/debug_nan.py c > gpurun_out/nan_c.log 2>&1; tail -3 gpurun_out/nan_c.log
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "train_parity\[|passed|failed|Error|assert" gpurun_out/pytest_gpu.log | tail -20
