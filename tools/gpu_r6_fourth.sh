#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6d; mkdir -p $E; cd $R
timeout 1500 python -m pytest tests/test_parity_configs_gpu.py -x -q -s > $E/pytest_configs.log 2>&1; grep "share of elements\|passed\|failed" $E/pytest_configs.log
