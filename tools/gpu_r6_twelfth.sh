#!/bin/bash
# slam_pool_mlp_kernel with its pooling operands fetched before the MLP; conv2's pooling epilogue at B = 32 (SRHIP_POOL_EPI_ANY=1)
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6n; mkdir -p $E; cd $R
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "tail or clam or slam or rab or resgroup or generator" 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2 3; do
  timeout 300 $B 2>&1 | tail -1 > $E/new_$i.json; cut -c1-140 $E/new_$i.json
  SRHIP_TAIL_DBG=64 timeout 300 $B 2>&1 | tail -1 > $E/old_$i.json; cut -c1-140 $E/old_$i.json
  SRHIP_POOL_EPI_ANY=1 timeout 300 $B 2>&1 | tail -1 > $E/epi_$i.json; cut -c1-140 $E/epi_$i.json
done
timeout 300 python tools/step_timeline.py 2>&1 | grep "G fwd done\|step to step"
SRHIP_TAIL_DBG=64 timeout 300 python tools/step_timeline.py 2>&1 | grep "G fwd done\|step to step"
SRHIP_POOL_EPI_ANY=1 timeout 300 python tools/step_timeline.py 2>&1 | grep "G fwd done\|step to step"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $E/tt_new -o t --output-format csv -- python3 $R/tools/time_tail_train.py > $E/tt_new.log 2>&1
cd $R; python tools/kstats.py $E/tt_new 14 | grep "slam_pool\|clam"; rm -rf $E/tt_new
