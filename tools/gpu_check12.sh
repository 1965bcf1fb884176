#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --trace-losses 2>&1 | grep -E "losses per step|ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]*).*/\1/' | cut -c1-200; }
run AMD_SERIALIZE_KERNEL=3
run AMD_SERIALIZE_COPY=3
run AMD_SERIALIZE_KERNEL=1
run AMD_SERIALIZE_KERNEL=2
run BENCH_X=1
echo "== eager"; timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-graph 2>&1 | grep -E "ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]*).*/\1/'
