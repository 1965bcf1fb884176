#!/bin/bash
# complete RAB weight-gradient pairs start in slots behind conv1's data gradient (SRHIP_WGRAD_SLOTS=1, new) against the round-5 request order (=0)
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6g; mkdir -p $E; cd $R
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "compact_record or rab or resgroup or generator_small or train_two" 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2 3; do
  SRHIP_WGRAD_SLOTS=1 timeout 300 $B 2>&1 | tail -1 > $E/slots1_$i.json; cut -c1-140 $E/slots1_$i.json
  SRHIP_WGRAD_SLOTS=0 timeout 300 $B 2>&1 | tail -1 > $E/slots0_$i.json; cut -c1-140 $E/slots0_$i.json
done
SRHIP_WGRAD_SLOTS=1 timeout 300 python tools/step_timeline.py 2>&1 | tail -12
SRHIP_WGRAD_SLOTS=0 timeout 300 python tools/step_timeline.py 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
export SRHIP_WGRAD_SLOTS=1
rocprofv3 --kernel-trace --stats -d $E/kt1 -o kt --output-format csv -- python3 $R/tools/run_steps.py 8 > $E/kt1.log 2>&1
export SRHIP_WGRAD_SLOTS=0
rocprofv3 --kernel-trace --stats -d $E/kt0 -o kt --output-format csv -- python3 $R/tools/run_steps.py 8 > $E/kt0.log 2>&1
cd $R
echo "slots=1"; python tools/kstats.py $E/kt1 12; echo "slots=0"; python tools/kstats.py $E/kt0 12
rm -rf $E/kt1 $E/kt0
