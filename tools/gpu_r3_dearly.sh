#!/bin/bash
# round 3: discriminator backward of the real + penalty terms enqueued on the D stream before the generator's backward (SRHIP_D_EARLY)
O=gpurun_out/r3d; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_parity_bench_gpu.py -x -q -k "train or determin or graph or first_step or b12 or post_step or discriminator or penalty" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
B="python bench.py --steps 20 --warmup 5 --no-fp32-line --no-cpu-baseline --no-sustained"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['launch'], (d.get('power') or {}).get('watts_mean'), d['last_losses'])
except Exception as e: print('$name FAILED', e, open('$O/$name.err').read()[-800:])
PY
}
run early_a SRHIP_D_EARLY=1
run late_a SRHIP_D_EARLY=0
run early_b SRHIP_D_EARLY=1
run late_b SRHIP_D_EARLY=0
SRHIP_D_EARLY=1 timeout 200 python tools/step_timeline.py > $O/timeline_early.txt 2>&1; cat $O/timeline_early.txt | tail -14
SRHIP_D_EARLY=0 timeout 200 python tools/step_timeline.py > $O/timeline_late.txt 2>&1; cat $O/timeline_late.txt | tail -13
for v in "SRHIP_D_EARLY=1" "SRHIP_D_EARLY=0"; do
  env $v timeout 300 python bench.py --workload chain --scales 2,3,8,9 --conv-math bf16x3 --steps 12 --warmup 4 --spinup-steps 10 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k:(v['ms_per_step']) for k,v in d['per_scale'].items()})"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/loop_shape.hip -o /tmp/loop_shape 2>/dev/null && timeout 120 /tmp/loop_shape | tee $O/loop_shape.txt
