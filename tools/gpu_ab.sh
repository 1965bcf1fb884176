#!/bin/bash
# A/B of the whole step under environment variants: usage gpu_ab.sh name1 "ENV=.." name2 "ENV=.." ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-fp32-line --steps 20 --warmup 5 > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["last_losses"]["loss_G"], d["last_losses"]["loss_D"], d["power"]["sclk_mhz_mean"], d["power"]["watts_mean"])' 2>&1 | tail -1)"; }
for rep in 1 2 3; do
  i=1
  while [ $i -le $# ]; do
    n=${!i}; j=$((i+1)); e=${!j}
    run ${n}_$rep $e
    i=$((i+2))
  done
done
