#!/bin/bash
# Round-2 first GPU pass: new kernels' tests, then the whole GPU suite, then bench lines (plain and single-rank RCCL).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_attention_gpu.py -x -q -m gpu -s > $O/attn.log 2>&1; echo "attn rc=$?" | tee -a $O/attn.log; tail -15 $O/attn.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_attention_gpu.py -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-fp32-line > $O/bench_dist.log 2>&1; tail -1 $O/bench_dist.log | cut -c1-300
