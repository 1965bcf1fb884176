#!/bin/bash
mkdir -p gpurun_out
python tools/debug_dvfs.py > gpurun_out/dvfs.log 2>&1; cat gpurun_out/dvfs.log | tail -8
python tools/debug_graph.py 8 10 > gpurun_out/debug_graph.log 2>&1; tail -22 gpurun_out/debug_graph.log
python -m pytest tests/test_model_gpu.py -m gpu -q -k "train_two or adam" > gpurun_out/pytest_train.log 2>&1; tail -5 gpurun_out/pytest_train.log
