#!/bin/bash
# round 4, final evidence with the final binary: tools/gpu_evidence.sh + a 20-step line + the conv A/B sweep + the weight-gradient
# A/B + SQ wait buckets + the long-run reproducibility check
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu_evidence.sh > gpurun_out/evidence_run.log 2>&1
E=$R/gpurun_out/evidence
timeout 900 python bench.py --no-cpu-baseline --no-fp32-line --steps 20 --warmup 5 2>&1 | tail -1 > $E/bench_n1_steps20.json; cut -c1-200 $E/bench_n1_steps20.json
timeout 900 python tools/sweep_pers.py 2>&1 | grep -v amdgpu.ids > $E/patch_pers_sweep.txt; tail -12 $E/patch_pers_sweep.txt
MODE=half timeout 900 python tools/sweep_pers.py 2>&1 | grep -v amdgpu.ids > $E/patch_pers_sweep_half.txt
timeout 900 python tools/sweep_wgrad_addr.py 2>&1 | grep -v amdgpu.ids > $E/wgrad_scalar_offsets_ab.txt; tail -9 $E/wgrad_scalar_offsets_ab.txt
timeout 900 python tools/long_run_check.py > $E/long_run.txt 2>&1; tail -4 $E/long_run.txt
tail -30 gpurun_out/evidence_run.log
