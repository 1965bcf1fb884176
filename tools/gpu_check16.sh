#!/bin/bash
for b in 32 16 8 4; do
  timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --batch $b 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"value": ([0-9.]*).*("ms_per_step": [0-9.]*).*("per-GPU batch [0-9]*").*/eager \1 img\/s \2 \3/'
done
AMD_SERIALIZE_KERNEL=2 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --batch 8 --graph 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"value": ([0-9.]*).*("ms_per_step": [0-9.]*).*/graph(serialize=2) B=8 \1 img\/s \2/'
AMD_SERIALIZE_KERNEL=2 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --batch 32 --graph 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"value": ([0-9.]*).*("ms_per_step": [0-9.]*).*/graph(serialize=2) B=32 \1 img\/s \2/'
