"""Samples the GPU's shader clock and power (rocm-smi) while one kernel selection runs in a loop: does the clock under the full
conv kernel differ from the clock under its MFMA-only / memory-only ablations?"""
import os, sys, subprocess, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
x = torch.randn(32, 64, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05)
b = torch.randn(256, device=dev) * 0.01
fn = lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
            sclk = re.findall(r'sclk clock level: \d+: \((\d+)Mhz\)', r)
            pw = re.findall(r'Power \(W\): ([\d.]+)', r)
            out.append((sclk[:1], pw[:1]))
        except Exception as e:
            out.append(('err', str(e)[:60]))
        time.sleep(0.2)


for cfg in [int(v) for v in sys.argv[1].split(',')]:
    lib.srhip_debug_set(0, cfg)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        for _ in range(200): fn()
        torch.cuda.synchronize(); n += 200
    dt = time.time() - t0
    stop.set(); th.join()
    print('cfg %3d: %.1f us per launch; samples (sclk MHz, W): %s' % (cfg, dt / n * 1e6, out[1:8]), flush=True)
lib.srhip_debug_set(0, 0)
