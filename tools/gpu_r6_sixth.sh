#!/bin/bash
# communicator created inside the first step's exchange (default) against created when the TrainStep is built (SRHIP_DP_EAGER_INIT=1)
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6f; mkdir -p $E; cd $R
B="python bench.py --steps 20 --warmup 5 --step-only"
for i in 1 2; do
  BENCH_FORCE_DIST=1 timeout 300 $B 2>&1 | tail -1 > $E/lazy_$i.json; cut -c1-150 $E/lazy_$i.json
  BENCH_FORCE_DIST=1 SRHIP_DP_EAGER_INIT=1 timeout 300 $B 2>&1 | tail -1 > $E/eager_$i.json; cut -c1-150 $E/eager_$i.json
done
timeout 300 $B 2>&1 | tail -1 > $E/plain.json; cut -c1-150 $E/plain.json
BENCH_FORCE_DIST=1 timeout 300 python tools/step_timeline.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" > $E/step_timeline_forced_single_rank_rccl.txt; cat $E/step_timeline_forced_single_rank_rccl.txt
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $E/bench_n1_rccl_single_rank.json; cut -c1-200 $E/bench_n1_rccl_single_rank.json
timeout 600 python -m pytest tests/test_parity_configs_gpu.py -x -q -k "forced_single_rank" 2>&1 | tail -3
