#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-fp32-line --steps 10 > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["last_losses"]["loss_G"])' 2>&1 | tail -1)"; }
run late1 SRHIP_LATE_JOIN=1
run early1 SRHIP_LATE_JOIN=0
run late2 SRHIP_LATE_JOIN=1
run early2 SRHIP_LATE_JOIN=0
