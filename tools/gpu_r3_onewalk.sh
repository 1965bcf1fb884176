#!/bin/bash
# round 3: one walk of D(gen) for both losses (SRHIP_D_ONEWALK) and the fused tail backward (SRHIP_TAIL_FUSED): tests + same-box A/B
O=gpurun_out/r3e; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_parity_bench_gpu.py tests/test_sragan.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
B="python bench.py --steps 20 --warmup 5 --no-fp32-line --no-cpu-baseline --no-sustained"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], 'img/s', d['ms_per_step'], 'ms', (d.get('power') or {}).get('watts_mean'), d['last_losses'])
except Exception as e: print('$name FAILED', e, open('$O/$name.err').read()[-1200:])
PY
}
run new_a X=1
run twowalk_a SRHIP_D_ONEWALK=0
run tail3_a SRHIP_TAIL_FUSED=0
run new_b X=1
run twowalk_b SRHIP_D_ONEWALK=0
run tail3_b SRHIP_TAIL_FUSED=0
timeout 200 python tools/step_timeline.py > $O/timeline.txt 2>&1; tail -14 $O/timeline.txt
timeout 200 python tools/host_profile.py > $O/host.log 2>&1; head -3 $O/host.log | tail -2
