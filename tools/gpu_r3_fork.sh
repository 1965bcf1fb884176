#!/bin/bash
# round 3: side-stream fork through the C helper (srhip_stream_fork) vs torch events: tests + host time + step time, same box
O=gpurun_out/r3k; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_conv_gpu.py -x -q -k "train_two_iterations_small or determin or graph or first_step or conv_fwd_bwd or rowtap or checkpoint" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
for v in "X=1" "SRHIP_FORK_C=0" "X=1" "SRHIP_FORK_C=0"; do
  env $v timeout 200 python tools/host_profile.py 2>&1 | grep "host enqueue" | sed "s/^/$v /"
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-fp32-line --no-cpu-baseline --no-sustained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms')"
done
for v in "X=1" "SRHIP_FORK_C=0"; do env $v timeout 400 python bench.py --workload chain --scales 8,9 --conv-math bf16x3 --steps 20 --warmup 5 --spinup-steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k:(v['ms_per_step']) for k,v in d['per_scale'].items()})"; done
