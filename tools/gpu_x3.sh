#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x3
mkdir -p $O
cd $R
timeout 600 python bench.py --workload chain --scales 3 --conv-math bf16x3 --steps 4 --warmup 2 --spinup-steps 2 2>&1 | tail -1 | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/p -o st -- python3 $R/bench.py --workload chain --scales 3 --conv-math bf16x3 --steps 3 --warmup 1 --spinup-steps 0 > $O/p.log 2>&1
cd $R
f=$(find $O/p -name "*.db" | head -1); python tools/rocpd_stats.py $f 14
rm -rf $O/p
