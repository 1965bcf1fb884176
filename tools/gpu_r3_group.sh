#!/bin/bash
# round 3: grouped row-tap weight gradients (srhip_conv2d_wgrad_multi, SRHIP_WGRAD_GROUP): tests, op-level roofline, same-box step A/B
O=gpurun_out/r3m; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_parity_bench_gpu.py -x -q -k "grouped or rowtap or stream_fork or train_two_iterations_small or train_two_iterations_full or determin or graph or first_step or b12 or post_step" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log | cut -c1-200
for g in 2 1 3 4; do SRHIP_WGRAD_GROUP=$g timeout 200 python bench.py --roofline-only --no-sustained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline_wgrad']; print('group $g', d['avg_launch_ms'], 'ms per launch', d['achieved'], 'TF/s frac', d['frac'], d['kernel'][:70])"; done
B="python bench.py --steps 20 --warmup 5 --no-fp32-line --no-cpu-baseline --no-sustained"
for g in 2 1 2 1 4; do SRHIP_WGRAD_GROUP=$g timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $g', d['value'], 'img/s', d['ms_per_step'], 'ms', d['last_losses'])"; done
