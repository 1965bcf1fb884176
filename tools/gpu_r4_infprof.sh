#!/bin/bash
# round 4: per-kernel table of the B = 16 inference workload (eager launches so that every kernel is visible)
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/infprof
mkdir -p $E
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $E/kt -o kt --output-format csv -- python3 $R/bench.py --workload infer --steps 20 --warmup 5 > $E/run.log 2>&1
cd $R
tail -1 $E/run.log | cut -c1-200
python tools/kstats.py $E/kt 40 > $E/infer_kernel_stats.txt 2>&1; head -44 $E/infer_kernel_stats.txt | cut -c1-150
