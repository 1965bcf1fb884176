#!/bin/bash
# Round evidence in one call: default bench line (with cpu_baseline), fp32-mode line, half-mode line, inference line, chain
# line, single-rank RCCL line, whole-step kernel stats (rocprofv3), roofline kernel-trace + PMC passes per conv arithmetic mode.
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/evidence
mkdir -p $E
cd $R
# the roofline passes first: the bench lines below read the traffic figures of THIS binary (bench.py refuses a file of another source hash)
bash tools/gpu_roofline2.sh > $E/roofline2.log 2>&1; tail -12 $E/roofline2.log
cp $R/gpurun_out/roof2/summary.txt $E/roofline_pmc_summary.txt; cp $R/gpurun_out/roof2/roofline_traffic.json $E/roofline_traffic.json
for M in fp32 bf16x3 half; do f=$(find $R/gpurun_out/roof2/$M/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $E/roofline_only_${M}_kernel_stats.csv; done
cp $E/roofline_traffic.json $R/profiles/roofline_traffic.json
cd $R
timeout 900 python bench.py > $E/bench_n1.log 2>&1; tail -1 $E/bench_n1.log > $E/bench_n1.json; cut -c1-260 $E/bench_n1.json
timeout 600 python bench.py --no-cpu-baseline --conv-math fp32 2>&1 | tail -1 > $E/bench_n1_fp32.json; cut -c1-200 $E/bench_n1_fp32.json
timeout 600 python bench.py --no-cpu-baseline --conv-math half 2>&1 | tail -1 > $E/bench_n1_half.json; cut -c1-200 $E/bench_n1_half.json
timeout 600 python bench.py --workload infer 2>&1 | tail -1 > $E/bench_infer.json; cut -c1-260 $E/bench_infer.json
for W in srgan sragan edsr; do timeout 600 python bench.py --workload $W --steps 20 --warmup 5 2>&1 | tail -1 > $E/bench_$W.json; cut -c1-200 $E/bench_$W.json; done
timeout 900 python bench.py --workload chain 2>&1 | tail -1 > $E/bench_chain_half.json; cut -c1-300 $E/bench_chain_half.json
timeout 900 python bench.py --workload chain --conv-math bf16x3 2>&1 | tail -1 > $E/bench_chain_bf16x3.json; cut -c1-300 $E/bench_chain_bf16x3.json
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $E/bench_n1_rccl_single_rank.json; cut -c1-200 $E/bench_n1_rccl_single_rank.json
timeout 600 python bench.py --no-cpu-baseline --no-fp32-line 2>&1 | tail -1 > $E/bench_n1_again.json; cut -c1-200 $E/bench_n1_again.json
timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-fp32-line 2>&1 | tail -1 > $E/bench_n1_steps20.json; cut -c1-200 $E/bench_n1_steps20.json
cd /tmp && export TMPDIR=/tmp
# the step and NOTHING else (round 6: --step-only drops the roofline section and the probe steps, whose isolated launches used to sit in this table)
rocprofv3 --kernel-trace --stats -d $E/step -o st -- python3 $R/bench.py --step-only --spinup-steps 0 --steps 6 --warmup 2 > $E/step.log 2>&1
cd $R
f=$(find $E/step -name "*.db" | head -1); python tools/rocpd_stats.py $f 70 > $E/step_kernel_stats.txt; head -14 $E/step_kernel_stats.txt
rm -rf $E/step/*.db
BENCH_FORCE_DIST=1 timeout 300 python tools/step_timeline.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" > $E/step_timeline_forced_single_rank_rccl.txt; tail -22 $E/step_timeline_forced_single_rank_rccl.txt
timeout 300 python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > $E/step_timeline.txt
timeout 300 python tools/time_d_convs.py 2>&1 | grep -v amdgpu.ids > $E/d_convs.txt
bash tools/gpu_r6_traffic.sh > $E/traffic.log 2>&1; cp $R/gpurun_out/r6traffic/table.txt $E/step_traffic.txt; head -5 $E/step_traffic.txt
