#!/bin/bash
python tools/sweep_wgrad.py 2>&1 | tail -36
python -m pytest tests/test_conv_gpu.py -m gpu -q 2>&1 | tail -5
