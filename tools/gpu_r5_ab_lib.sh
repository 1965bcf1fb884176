#!/bin/bash
# Same-box A/B of two builds of the library: sradsgan_amd/lib/libsradsgan_hip_prev.so (the previous commit's sources) against the current one.
O=gpurun_out/r5ablib; mkdir -p $O
B="python bench.py --steps 20 --no-cpu-baseline --no-fp32-line --no-sustained"
P=$PWD/sradsgan_amd/lib/libsradsgan_hip_prev.so
for i in 1 2; do
  SRHIP_LIB=$P $B > $O/prev$i.json 2>$O/prev$i.err
  $B > $O/cur$i.json 2>$O/cur$i.err
done
for f in prev1 cur1 prev2 cur2; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], 'fprop in-step', d['roofline'].get('in_step_avg_launch_ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done | tee $O/summary.txt
