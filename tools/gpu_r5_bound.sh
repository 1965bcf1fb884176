#!/bin/bash
# Is the step bound by the host, by the main stream's chain, or by the GPU's total throughput?  One box, one call.
mkdir -p gpurun_out/r5bound; O=gpurun_out/r5bound
B="python bench.py --steps 20 --no-cpu-baseline --no-fp32-line --no-sustained"
python tools/host_profile.py > $O/host.log 2>&1
$B > $O/default.json 2>$O/default.err
SRHIP_OVERLAP_WGRAD=0 SRHIP_OVERLAP_D=0 $B > $O/one_stream.json 2>$O/one.err
SRHIP_OVERLAP_D=0 $B > $O/no_d_stream.json 2>$O/nod.err
SRHIP_RUN_AHEAD=0 $B > $O/runahead0.json 2>$O/ra.err
$B > $O/default2.json 2>$O/default2.err
python tools/step_timeline.py > $O/timeline.txt 2>&1
for f in default one_stream no_d_stream runahead0 default2; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done | tee $O/summary.txt
head -3 $O/host.log; head -40 $O/timeline.txt
