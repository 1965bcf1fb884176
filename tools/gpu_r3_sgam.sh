#!/bin/bash
# round 3: SGAM products in split-bf16: parity (fp64) + time against the exact-fp32 kernels, op level and step level
O=gpurun_out/r3g; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_model_gpu.py -x -q -s -k "sgam or gab_up or generator_small or train_two" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; grep -E "^sgam|passed|failed|Error" $O/tests.log | tail -30
timeout 300 python tools/time_sgam.py 2>&1 | grep -v amdgpu.ids | tee $O/time_sgam.txt
B="python bench.py --steps 20 --warmup 5 --no-fp32-line --no-cpu-baseline --no-sustained"
for v in "X=1" "SRHIP_DEBUG=4:1" "X=1" "SRHIP_DEBUG=4:1"; do env $v timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['last_losses'])"; done
for v in "X=1" "SRHIP_DEBUG=4:1"; do env $v timeout 400 python bench.py --workload chain --scales 2,3 --conv-math bf16x3 --steps 10 --warmup 3 --spinup-steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k:(v['ms_per_step']) for k,v in d['per_scale'].items()})"; done
