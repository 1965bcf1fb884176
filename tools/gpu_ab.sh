#!/bin/bash
# A/B of the whole step under SRHIP_DEBUG variants: usage gpu_ab.sh "name1=ENV..." ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-fp32-line --steps 10 > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["last_losses"]["loss_G"], d["roofline_wgrad"]["avg_launch_ms"])' 2>&1 | tail -1)"; }
for rep in 1 2; do
run base$rep A=1
run rowtap2_$rep SRHIP_DEBUG=1:8
done
