#!/bin/bash
# same-box A/B of two builds of the library (SRHIP_LIB): one-tile patch kernels with the prologue DMAs issued before the address tables
mkdir -p gpurun_out
P=$GRAFT_REPO_ROOT/sradsgan_amd/lib/libsradsgan_hip_prev.so
( for r in 1 2; do
    echo "== new build"; timeout 600 python tools/sweep_ks.py 2>&1 | grep "K-split (default)"
    B=16 VARS=-1 ROUNDS=3 timeout 600 python tools/sweep_pers.py 2>&1 | grep "median" | head -2
    echo "== previous build"; SRHIP_LIB=$P timeout 600 python tools/sweep_ks.py 2>&1 | grep "K-split (default)"
    SRHIP_LIB=$P B=16 VARS=-1 ROUNDS=3 timeout 600 python tools/sweep_pers.py 2>&1 | grep "median" | head -2
  done
  for r in 1 2; do
    echo "infer new : $(timeout 600 python bench.py --workload infer 2>&1 | tail -1 | cut -c90-130)"
    echo "infer prev: $(SRHIP_LIB=$P timeout 600 python bench.py --workload infer 2>&1 | tail -1 | cut -c90-130)"
  done ) > gpurun_out/r4_ab_lib.txt 2>&1
cat gpurun_out/r4_ab_lib.txt
