#!/bin/bash
# Round 4: does holding G's weight gradients until the end of the backward (and launching them in groups there) hurt the step?
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4h
mkdir -p $E
cd $R
B="python bench.py --no-cpu-baseline --no-fp32-line --no-sustained --steps 12"
for rep in 1 2; do
  timeout 600 $B 2>&1 | tail -1 > $E/base_$rep.json; echo "base           $(python -c "import json;d=json.load(open('$E/base_$rep.json'));print(d['ms_per_step'], d['value'])")"
  SRHIP_WGRAD_DEFER=1 timeout 600 $B 2>&1 | tail -1 > $E/defer2_$rep.json; echo "defer, pairs   $(python -c "import json;d=json.load(open('$E/defer2_$rep.json'));print(d['ms_per_step'], d['value'])")"
  SRHIP_WGRAD_DEFER=1 SRHIP_WGRAD_FLUSH_GROUP=4 SRHIP_WGRAD_GROUP=4 timeout 600 $B 2>&1 | tail -1 > $E/defer4_$rep.json; echo "defer, fours   $(python -c "import json;d=json.load(open('$E/defer4_$rep.json'));print(d['ms_per_step'], d['value'])")"
  SRHIP_WGRAD_GROUP=4 timeout 600 $B 2>&1 | tail -1 > $E/group4_$rep.json; echo "group 4        $(python -c "import json;d=json.load(open('$E/group4_$rep.json'));print(d['ms_per_step'], d['value'])")"
done
