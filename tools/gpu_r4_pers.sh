#!/bin/bash
# Round 4, call 1: persistent patch kernel -- bit-identity tests, same-box A/B sweep, step time with and without it.
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4a
mkdir -p $E
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "persistent or bit_identical" > $E/tests.log 2>&1; tail -5 $E/tests.log
timeout 600 python tools/sweep_pers.py > $E/sweep.log 2>&1; cat $E/sweep.log
SRHIP_DEBUG=5:-1 timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_old.json; cut -c1-200 $E/bench_old.json
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_new.json; cut -c1-200 $E/bench_new.json
SRHIP_DEBUG=5:-1 timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_old2.json; cut -c1-200 $E/bench_old2.json
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line --no-sustained 2>&1 | tail -1 > $E/bench_new2.json; cut -c1-200 $E/bench_new2.json
