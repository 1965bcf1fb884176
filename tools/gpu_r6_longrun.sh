#!/bin/bash
# the final binary of round 6: two 150-iteration runs bit-identical? + the same trajectory through the forced single-rank RCCL exchange; allocator footprint over 400 steps
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6l; mkdir -p $E; cd $R
RCCL=1 timeout 900 python tools/long_run_check.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $E/long_run.txt; cat $E/long_run.txt
timeout 900 python tools/mem_growth.py 2>&1 | grep -v amdgpu.ids > $E/mem_growth.txt; cat $E/mem_growth.txt
