"""Two ranks on two GPUs: TrainStep + dp.GradSync over a real RCCL communicator (BASELINE configs[3]'s exchange step at
world 2).  Skipped when fewer than two GPUs are visible (the gpurun boxes have one); the single-rank RCCL path is pinned by
tests/test_parity_configs_gpu.py and the ordering logic by tests/test_dp_gloo.py (world 2, gloo, CPU)."""
import os
import socket
import subprocess
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_train_step_gradient_exchange_world2():
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_world2_worker.py')], env=env))
    deadline = time.monotonic() + 600
    codes = [None, None]
    try:
        while any(c is None for c in codes) and time.monotonic() < deadline:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes):
                break                                           # one rank failed: do not wait for the other to hang
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0, 0], codes
