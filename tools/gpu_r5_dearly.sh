#!/bin/bash
# D(real) beside the generator's forward (SRHIP_D_REAL_EARLY 0 / 1 / 2), same box, alternating
O=gpurun_out/r5dearly; mkdir -p $O
B="python bench.py --steps 20 --no-cpu-baseline --no-fp32-line --no-sustained"
for i in 1 2; do for m in 0 1 2; do SRHIP_D_REAL_EARLY=$m $B 2>$O/err_$m$i.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early=$m', d['value'], d['ms_per_step'], d['last_losses'])"; done; done | tee $O/summary.txt
SRHIP_D_REAL_EARLY=2 python tools/step_timeline.py 2>&1 | grep -v amdgpu | head -16 | tee $O/timeline_early2.txt
