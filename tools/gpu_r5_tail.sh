#!/bin/bash
# kernel trace of six plain steps -> the kernels of the last 2.5 ms of a step, per stream (what the step's end waits for)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5tail; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o kt -- python3 $R/tools/run_steps.py 8 > $O/log.txt 2>&1
cd $R; f=$(find $O -name "*.db" | head -1); python tools/step_tail.py $f 2.5 | tee $O/tail.txt | tail -70
