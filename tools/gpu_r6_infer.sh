#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6k; mkdir -p $E
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $E/kt -o kt --output-format csv -- python3 $R/bench.py --workload infer --no-graph --step-only --steps 20 --spinup-steps 5 > $E/kt.log 2>&1
cd $R; python tools/kstats.py $E/kt 30 | tee $E/infer_kernels.txt; rm -rf $E/kt
