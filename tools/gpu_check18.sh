#!/bin/bash
R=$GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8 NO_DEVICE_ID=1 SYNC_EACH=1 OVERLAP=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tiny -o t -- python3 $R/tools/debug_dist_overhead.py tiny > $R/gpurun_out/prof_tiny.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_plain -o t -- python3 $R/tools/debug_dist_overhead.py plain > $R/gpurun_out/prof_plain.log 2>&1
grep wall $R/gpurun_out/prof_tiny.log $R/gpurun_out/prof_plain.log | cut -c1-160
