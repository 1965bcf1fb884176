#!/bin/bash
# per-kernel durations of six plain steps (tools/run_steps.py), three streams sharing the chip
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5kt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/tools/run_steps.py 6 > $O/log.txt 2>&1
cd $R; python tools/kstats.py $O ${1:-60} | tee $O/table.txt
