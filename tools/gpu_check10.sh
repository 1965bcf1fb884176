#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --trace-losses 2>&1 | grep -E "losses per step" ; }
run BENCH_X=1
run BENCH_SYNC_EACH=1
run BENCH_NO_WARM_BARRIER=1
run BENCH_NO_GP=1
python -m pytest tests/test_conv_gpu.py -m gpu -q -k "batch_norm" 2>&1 | tail -5
