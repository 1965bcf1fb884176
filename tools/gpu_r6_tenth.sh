#!/bin/bash
# three RABs per flat launch: 28 splits per convolution (252 blocks; 4 of the 28 splits are not XCD-aligned) against 24 (216 blocks, all aligned)
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6j; mkdir -p $E; cd $R
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2; do
  timeout 300 $B 2>&1 | tail -1 | cut -c1-140
  SRHIP_FLAT_BLOCKS=216 timeout 300 $B 2>&1 | tail -1 | cut -c1-140
done
timeout 300 python bench.py --roofline-only --no-sustained 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('28 splits', d['roofline_wgrad']['avg_launch_ms'], d['roofline_wgrad']['frac'])"
SRHIP_FLAT_BLOCKS=216 timeout 300 python bench.py --roofline-only --no-sustained 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('24 splits', d['roofline_wgrad']['avg_launch_ms'], d['roofline_wgrad']['frac'])"
