#!/bin/bash
# how many cores does the GPU box give us, and what does the oracle's speed do with the thread count?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, 'absent')
import torch; print('torch threads default', torch.get_num_threads())
PY
for T in 0 4 8 16; do
  SRHIP_TEST_THREADS=$T timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "two_iterations_small" --durations=3 2>&1 | grep -E "passed|failed|call" | head -3
done
