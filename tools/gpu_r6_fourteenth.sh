#!/bin/bash
# srhip_cat_channels in the multi-scale block against torch.cat (SRHIP_CAT=0): tests, same-box A/B
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6p; mkdir -p $E; cd $R
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "cat_channels or msb or generator or train_two" 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2 3; do
  timeout 300 $B 2>&1 | tail -1 > $E/cat1_$i.json; cut -c1-140 $E/cat1_$i.json
  SRHIP_CAT=0 timeout 300 $B 2>&1 | tail -1 > $E/cat0_$i.json; cut -c1-140 $E/cat0_$i.json
done
