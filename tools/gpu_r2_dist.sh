#!/bin/bash
# Why does the single-rank RCCL line cost +12 %?  One knob at a time.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-fp32-line --steps 10 > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
run plain A=1
run dist BENCH_FORCE_DIST=1
run dist_fake BENCH_FORCE_DIST=1 SRHIP_DP_MODE=fake
run plain_mainstream BENCH_MAIN_STREAM=1
run dist_mainstream BENCH_FORCE_DIST=1 BENCH_MAIN_STREAM=1
run dist_prio BENCH_FORCE_DIST=1 SRHIP_DP_PRIO=1
run dist_q16 BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=16
run dist_q4 BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4
run dist_mainstream_q16 BENCH_FORCE_DIST=1 BENCH_MAIN_STREAM=1 GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_parity_bench_gpu.py tests/test_trainer_gpu.py -q -m gpu -s > $O/parity.log 2>&1; echo "parity rc=$?"; grep -E "b12|x[2389] @|passed|failed|within 2e-5" $O/parity.log | tail -30
