#!/bin/bash
mkdir -p gpurun_out
for v in "" "--trace-losses" "--no-graph" ; do
  timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline $v 2>&1 | grep -E "losses per step|last_losses" | sed -E 's/.*("last_losses": \{[^}]*\}).*/\1/' 
done
python -m pytest tests/test_conv_gpu.py -m gpu -q -k "batch_norm" 2>&1 | tail -30
