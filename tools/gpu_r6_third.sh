#!/bin/bash
# Round 6, third call: the tail backward with the 7x7 data gradient inside the main pass -- test, per-kernel table, same-box A/B in the step.
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6c; mkdir -p $E; cd $R
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "tail or clam or slam or rab or resgroup or near_tie or compact" > $E/pytest_tail.log 2>&1; tail -4 $E/pytest_tail.log
B="python bench.py --steps 30 --warmup 5 --step-only"
for i in 1 2 3; do
  timeout 300 $B 2>&1 | tail -1 > $E/step_new_$i.json; cut -c1-140 $E/step_new_$i.json
  SRHIP_TAIL_DBG=32 timeout 300 $B 2>&1 | tail -1 > $E/step_old_$i.json; cut -c1-140 $E/step_old_$i.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $E/tt_new -o t --output-format csv -- python3 $R/tools/time_tail_train.py > $E/tt_new.log 2>&1
export SRHIP_TAIL_DBG=32
rocprofv3 --kernel-trace --stats -d $E/tt_old -o t --output-format csv -- python3 $R/tools/time_tail_train.py > $E/tt_old.log 2>&1
unset SRHIP_TAIL_DBG
cd $R
python tools/kstats.py $E/tt_new 16 > $E/tail_train_new.txt; cat $E/tail_train_new.txt
python tools/kstats.py $E/tt_old 16 > $E/tail_train_old.txt; cat $E/tail_train_old.txt
rm -rf $E/tt_new $E/tt_old
timeout 300 python bench.py --workload infer --batch 32 --step-only 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --workload infer --batch 32 --step-only --no-graph 2>&1 | tail -1 | cut -c1-300
