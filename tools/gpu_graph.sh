#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-fp32-line --steps 6 --warmup 3 --spinup-steps 3 --trace-losses > $O/$name.log 2>&1; echo "== $name: $(grep 'losses per step' $O/$name.log | cut -c1-300)"; tail -1 $O/$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["losses_finite"], d["last_losses"])' 2>&1 | tail -1; }
run eager A=1
run graph A=1 BENCH_GRAPH=1
run graph_nogp BENCH_GRAPH=1 BENCH_NO_GP=1
run graph_b8 BENCH_GRAPH=1 BENCH_BATCH=8
