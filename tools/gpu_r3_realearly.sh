#!/bin/bash
# round 3: D(real) forward + backward on the D stream beside the generator's forward (SRHIP_D_REAL_EARLY): tests + same-box A/B
O=gpurun_out/r3i; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_parity_bench_gpu.py -x -q --durations=12 -k "train or determin or graph or first_step or b12 or post_step or discriminator or penalty or sgam" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -22 $O/tests.log | cut -c1-200
B="python bench.py --steps 20 --warmup 5 --no-fp32-line --no-cpu-baseline --no-sustained"
for v in "X=1" "SRHIP_D_REAL_EARLY=0" "X=1" "SRHIP_D_REAL_EARLY=0"; do env $v timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['last_losses'])"; done
timeout 200 python tools/step_timeline.py > $O/timeline.txt 2>&1; tail -14 $O/timeline.txt
for v in "X=1" "SRHIP_D_REAL_EARLY=0"; do env $v timeout 400 python bench.py --workload chain --scales 2,3,8,9 --conv-math bf16x3 --steps 10 --warmup 3 --spinup-steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k:(v['ms_per_step']) for k,v in d['per_scale'].items()})"; done
