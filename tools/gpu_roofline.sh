#!/bin/bash
mkdir -p gpurun_out/roof
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/roof -o roof --output-format csv -- python3 $R/bench.py --roofline-only > $R/gpurun_out/roof/roofline_only.log 2>&1
cd $R; tail -1 gpurun_out/roof/roofline_only.log; grep fast_conv_dma gpurun_out/roof/*kernel_stats.csv | cut -c1-60,150-260
