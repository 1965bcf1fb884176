#!/bin/bash
mkdir -p gpurun_out
python tools/sweep_dma.py 2>&1 | tail -30
python -m pytest tests/test_conv_gpu.py -m gpu -q 2>&1 | tail -5
