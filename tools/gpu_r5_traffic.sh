#!/bin/bash
# HBM bytes of every kernel of the training step: two counter passes (FETCH_SIZE, WRITE_SIZE; dispatches are serialised by the
# profiler) over three plain steps (tools/run_steps.py), then tools/traffic_table.py.  Finds kernels that move more than their tensors.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/tools/run_steps.py 3 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/tools/run_steps.py 3 > $O/write.log 2>&1
cd $R
python tools/traffic_table.py $O 3 | tee $O/table.txt | head -70
