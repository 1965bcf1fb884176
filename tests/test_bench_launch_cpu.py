"""bench.self_launch -- the code `python bench.py --gpus N` (the driver's multi-GPU run) goes through first -- driven on CPU with
stub rank scripts: exit codes, the rank-0 line, a dying rank, a hung rank, a too-small node, a WORLD_SIZE that disagrees
with --gpus, and that the launcher parent counts GPUs without importing torch (it must never open the GPU driver)."""
import os
import subprocess
import sys
import time
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / 'rank_stub.py'
    p.write_text('import os, sys, time\nrank = int(os.environ["RANK"])\nworld = int(os.environ["WORLD_SIZE"])\n' + body)
    return str(p)


def _args(n):
    return types.SimpleNamespace(gpus=n)


def test_all_ranks_succeed_and_rank0_line_passes_through(tmp_path, capfd):
    script = _stub(tmp_path, 'assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0\n'
                             'assert os.environ["LOCAL_RANK"] == str(rank) and world == 4\n'
                             'print("{\\"rank\\": %d, \\"argv\\": \\"%s\\"}" % (rank, " ".join(sys.argv[1:])), flush=True)\n')
    rc = bench.self_launch(_args(4), script=script, argv=['--gpus', '4', '--steps', '3'], visible=4)
    out = capfd.readouterr().out
    assert rc == 0
    assert out.strip().splitlines() == ['{"rank": 0, "argv": "--gpus 4 --steps 3"}']     # only rank 0's stdout reaches ours


def test_one_failing_rank_stops_the_others_quickly(tmp_path, capfd):
    script = _stub(tmp_path, 'if rank == 2:\n    time.sleep(0.5)\n    sys.exit(3)\ntime.sleep(120)\n')
    t0 = time.monotonic()
    rc = bench.self_launch(_args(4), script=script, argv=[], visible=8)
    assert rc == 3
    assert time.monotonic() - t0 < 20.0
    assert 'rank 2 exited with code 3' in capfd.readouterr().err


def test_hung_rank_hits_the_deadline(tmp_path, monkeypatch, capfd):
    script = _stub(tmp_path, 'if rank == 1:\n    time.sleep(120)\n')
    monkeypatch.setenv('BENCH_LAUNCH_DEADLINE_S', '2')
    t0 = time.monotonic()
    rc = bench.self_launch(_args(2), script=script, argv=[], visible=2)
    assert rc == 124
    assert time.monotonic() - t0 < 20.0
    assert 'launch deadline' in capfd.readouterr().err


def test_fewer_gpus_than_ranks_is_refused(tmp_path, capfd):
    script = _stub(tmp_path, 'open(os.path.join(os.path.dirname(__file__), "ran_%d" % rank), "w").close()\n')
    rc = bench.self_launch(_args(8), script=script, argv=[], visible=1)
    assert rc == 2
    assert not list(tmp_path.glob('ran_*'))                                   # nothing was started
    assert 'only 1 GPU' in capfd.readouterr().err


def test_gpu_count_comes_from_sysfs_and_visible_devices(monkeypatch):
    """No GPU in the build container: zero KFD GPU agents, whatever the environment says; a *_VISIBLE_DEVICES list can only
    narrow the count; and the module has not pulled torch in (self_launch's parent never opens the GPU driver)."""
    base = bench.visible_gpu_count()
    assert base >= 0
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    assert bench.visible_gpu_count() == min(base, 3)
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '')
    assert bench.visible_gpu_count() == 0
    import ast
    tree = ast.parse(open(os.path.join(ROOT, 'bench.py')).read())
    for fn in (n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('self_launch', 'visible_gpu_count')):
        for node in ast.walk(fn):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                names = [a.name for a in node.names] + [getattr(node, 'module', None) or '']
                assert not any(n.split('.')[0] == 'torch' for n in names), 'the launcher parent must not import torch'
            assert not (isinstance(node, ast.Attribute) and node.attr == 'device_count')


def test_world_size_that_disagrees_with_gpus_is_an_error():
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert 'WORLD_SIZE=1 but --gpus 2' in out.stderr
    assert out.stdout.strip() == ''                                           # no JSON line for a job that is not the one asked for


# ---- the per-rank supervisor (bench.supervise_rank): fail-soft first N > 1 run ---------------------------------------------- #
_WORKER = '''import os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
attempt = int(os.environ["BENCH_ATTEMPT"])
assert os.environ["BENCH_WORKER"] == "1"
hb = os.environ["BENCH_HB_FILE"]
def beat():
    open(hb, "a").close(); os.utime(hb, None)
beat()
'''


def _run_supervisors(tmp_path, body, world, extra_env=None, timeout=60):
    """Starts `world` supervisors the way torch.distributed.run would start ranks (same parent, same MASTER_PORT), each around
    the stub worker; returns [(returncode, stdout, stderr)] by rank."""
    worker = tmp_path / 'worker_stub.py'
    worker.write_text(_WORKER + body)
    drv = tmp_path / 'sup_driver.py'
    drv.write_text('import sys, types\nsys.path.insert(0, %r)\nimport bench\n'
                   'sys.exit(bench.supervise_rank(types.SimpleNamespace(gpus=%d), worker_cmd=[sys.executable, %r]))\n'
                   % (ROOT, world, str(worker)))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT='29641',
                   BENCH_COORD_DIR=str(tmp_path / 'coord'), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(drv)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        out, err = p.communicate(timeout=timeout)
        res.append((p.returncode, out, err))
    return res


def test_supervisor_passes_a_clean_first_attempt_through(tmp_path):
    res = _run_supervisors(tmp_path, 'assert attempt == 1 and "SRHIP_DP_HOST_SYNC" not in os.environ\n'
                                     'print("{\\"rank\\": %d, \\"attempt\\": %d}" % (rank, attempt), flush=True)\n', 2)
    assert [r[0] for r in res] == [0, 0]
    assert res[0][1].strip() == '{"rank": 0, "attempt": 1}' and res[1][1].strip() == ''     # only rank 0's line comes out


def test_supervisor_retries_every_rank_host_synchronised_after_a_rank_dies(tmp_path):
    body = ('if attempt == 1:\n'
            '    if rank == 0:\n        print("{\\"stale\\": 1}", flush=True)\n'      # a line of the failed attempt must never surface
            '    if rank == 1:\n        sys.exit(7)\n'
            '    time.sleep(120)\n'                                                     # the peers sit in a collective for ever
            'assert attempt == 2 and os.environ["SRHIP_DP_HOST_SYNC"] == "1"\n'
            'assert os.environ["MASTER_PORT"] != "29641" and os.environ["TORCHELASTIC_USE_AGENT_STORE"] == "False"\n'
            'print("{\\"rank\\": %d, \\"attempt\\": %d, \\"port\\": %s}" % (rank, attempt, os.environ["MASTER_PORT"]), flush=True)\n')
    t0 = time.monotonic()
    res = _run_supervisors(tmp_path, body, 3)
    assert time.monotonic() - t0 < 40.0
    assert [r[0] for r in res] == [0, 0, 0]
    lines = res[0][1].strip().splitlines()
    assert len(lines) == 1 and '"attempt": 2' in lines[0] and 'stale' not in res[0][1]
    assert 'worker exited with code 7' in res[1][2] and 'retrying with SRHIP_DP_HOST_SYNC=1' in res[1][2]
    assert 'a peer reported failure' in res[0][2]


def test_supervisor_treats_a_silent_rank_as_hung_and_gives_up_after_the_retry(tmp_path):
    body = ('while True:\n    time.sleep(0.2)\n    if rank == 0:\n        beat()\n')      # rank 1 never beats again; rank 0 is alive but stuck
    t0 = time.monotonic()
    res = _run_supervisors(tmp_path, body, 2, extra_env={'BENCH_HEARTBEAT_STALE_S': '2', 'BENCH_FIRST_HEARTBEAT_S': '20'})
    assert time.monotonic() - t0 < 50.0
    assert all(r[0] != 0 for r in res)
    assert res[0][1] == ''                                                                # no line from a job that never finished
    assert 'no heartbeat from the worker' in res[1][2]
