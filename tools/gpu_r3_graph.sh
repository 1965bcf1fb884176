#!/bin/bash
# round 3: parity at the real tiles / B=32 / RCCL single rank / B=16 inference, then three-stream hipGraph capture vs eager
O=gpurun_out/r3b; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests/test_parity_configs_gpu.py -q -s > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/parity.log
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -s > $O/graph_test.log 2>&1; echo "graph test rc=$?" | tee -a $O/graph_test.log
tail -15 $O/graph_test.log
timeout 300 python tools/host_profile.py > $O/host_eager.log 2>&1; head -3 $O/host_eager.log
GRAPH=1 timeout 300 python tools/host_profile.py > $O/host_graph.log 2>&1; head -3 $O/host_graph.log
for mode in "" "--graph"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-fp32-line --no-cpu-baseline --no-sustained $mode > $O/bench_x4$mode.json 2> $O/bench_x4$mode.err; tail -c 600 $O/bench_x4$mode.json | head -c 300; echo
  timeout 600 python bench.py --workload chain --scales 8,9 --conv-math bf16x3 --steps 20 --warmup 5 $mode > $O/chain89$mode.json 2> $O/chain89$mode.err; cat $O/chain89$mode.json | head -c 900; echo
done
