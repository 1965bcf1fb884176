#!/bin/bash
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4b
mkdir -p $E
cd $R
timeout 600 python tools/ablate_pers.py > $E/ablate.log 2>&1; cat $E/ablate.log
SRHIP_DEBUG=5:-1 timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_old.json; cut -c1-200 $E/bench_old.json
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_new.json; cut -c1-200 $E/bench_new.json
SRHIP_DEBUG=5:-1 timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_old2.json; cut -c1-200 $E/bench_old2.json
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_new2.json; cut -c1-200 $E/bench_new2.json
