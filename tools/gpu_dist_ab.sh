#!/bin/bash
# single-rank RCCL line (BENCH_FORCE_DIST=1) under environment variants, against the plain line
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("rccl_ranks"))' $name; }
for rep in 1 2; do
run plain A=1
run dist_late BENCH_FORCE_DIST=1
run dist_dfirst BENCH_FORCE_DIST=1 SRHIP_D_FWD_FIRST=1
run dist_late_q8 BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=8
done
