#!/bin/bash
# Round 6, first GPU call: the GPU suite, the step on this box, the exchange forms A/B (single-rank RCCL), the timeline with parts.
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6a; mkdir -p $E; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $E/pytest.log 2>&1; tail -5 $E/pytest.log
B="python bench.py --steps 20 --warmup 5 --step-only"
for i in 1 2; do
  timeout 300 $B 2>&1 | tail -1 > $E/step_plain_$i.json; cut -c1-160 $E/step_plain_$i.json
  BENCH_FORCE_DIST=1 SRHIP_DP_PARTS=0 SRHIP_DP_THREAD=0 timeout 300 $B 2>&1 | tail -1 > $E/step_rccl_whole_caller_$i.json; cut -c1-160 $E/step_rccl_whole_caller_$i.json
  BENCH_FORCE_DIST=1 SRHIP_DP_PARTS=3 SRHIP_DP_THREAD=0 timeout 300 $B 2>&1 | tail -1 > $E/step_rccl_parts_caller_$i.json; cut -c1-160 $E/step_rccl_parts_caller_$i.json
  BENCH_FORCE_DIST=1 SRHIP_DP_PARTS=3 SRHIP_DP_THREAD=1 timeout 300 $B 2>&1 | tail -1 > $E/step_rccl_parts_thread_$i.json; cut -c1-160 $E/step_rccl_parts_thread_$i.json
  BENCH_FORCE_DIST=1 SRHIP_DP_PARTS=0 SRHIP_DP_THREAD=1 timeout 300 $B 2>&1 | tail -1 > $E/step_rccl_whole_thread_$i.json; cut -c1-160 $E/step_rccl_whole_thread_$i.json
done
BENCH_FORCE_DIST=1 timeout 300 python tools/step_timeline.py > $E/timeline_parts_thread.txt 2>&1; cat $E/timeline_parts_thread.txt
BENCH_FORCE_DIST=1 SRHIP_DP_THREAD=0 timeout 300 python tools/step_timeline.py > $E/timeline_parts_caller.txt 2>&1; cat $E/timeline_parts_caller.txt
timeout 300 python tools/step_timeline.py > $E/timeline_plain.txt 2>&1; tail -12 $E/timeline_plain.txt
timeout 300 python tools/time_d_convs.py > $E/d_convs.txt 2>&1; tail -12 $E/d_convs.txt
