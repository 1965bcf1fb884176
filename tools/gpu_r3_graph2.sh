#!/bin/bash
# round 3: why is the three-stream hipGraph replay slower on the GPU than the eager step?  stream-count and runtime-knob variants
O=gpurun_out/r3c; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_parity_configs_gpu.py -q -s -k "8-27 or 9-24" > $O/parity89.log 2>&1; echo "parity rc=$?" | tee -a $O/parity89.log
grep -E "vs fp64|vs fp32|passed|failed" $O/parity89.log | cut -c1-400
B="python bench.py --steps 12 --warmup 4 --spinup-steps 10 --no-fp32-line --no-cpu-baseline --no-sustained"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['launch'], (d.get('power') or {}).get('watts_mean'))
except Exception as e: print('$name FAILED', e)
PY
}
run eager3 X=1
run eager1 SRHIP_OVERLAP_WGRAD=0
run graph3 BENCH_GRAPH=1
run graph2 BENCH_GRAPH=1 SRHIP_OVERLAP_D=0
run graph1 BENCH_GRAPH=1 SRHIP_OVERLAP_WGRAD=0
run graph3_q4 BENCH_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run graph3_q8 BENCH_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run graph3_nocap BENCH_GRAPH=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run graph3_cap1 BENCH_GRAPH=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph3_dyn BENCH_GRAPH=1 DEBUG_HIP_DYNAMIC_QUEUES=1
timeout 300 python tools/host_profile.py > $O/host_eager.log 2>&1; head -3 $O/host_eager.log | tail -2
