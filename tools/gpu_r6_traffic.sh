#!/bin/bash
# HBM bytes AND in-step duration of every kernel of the training step: counter passes (FETCH_SIZE, WRITE_SIZE; dispatches serialised by the
# profiler) + one kernel-trace pass over plain steps (tools/run_steps.py), then tools/traffic_table.py (MB per call, us in the step, TB/s).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/tools/run_steps.py 3 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/tools/run_steps.py 3 > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/run_steps.py 3 > $O/kt.log 2>&1
cd $R
python tools/traffic_table.py $O 3 70 | tee $O/table.txt | head -75
rm -rf $O/fetch $O/write
