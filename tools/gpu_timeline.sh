#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o st -- python3 $R/bench.py --no-cpu-baseline --no-fp32-line --no-sustained --spinup-steps 0 --steps 6 --warmup 3 > /dev/null 2>&1
cd $R; f=$(find gpurun_out/prof_tl -name "*.db" | head -1); python tools/rocpd_timeline.py $f -2; python tools/rocpd_timeline.py $f -3 | head -8
rm -rf gpurun_out/prof_tl
