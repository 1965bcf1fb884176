#!/bin/bash
# conv2's CLAM pooling epilogue at B = 32 as the default: full GPU suite, A/B against the stand-alone pooling pass (srhip_debug_set(19, 0))
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6o; mkdir -p $E; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; tail -3 $E/pytest_gpu.log
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2 3; do
  timeout 300 $B 2>&1 | tail -1 > $E/epi_$i.json; cut -c1-140 $E/epi_$i.json
  SRHIP_POOL_EPI_ANY=0 timeout 300 $B 2>&1 | tail -1 > $E/pass_$i.json; cut -c1-140 $E/pass_$i.json
done
