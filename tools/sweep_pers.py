"""Round 4: persistent patch kernel (conv_patch_pers_kernel) against the one-tile-per-block kernel of rounds 1-3 on the RAB
convs at the bench shape: interleaved timing rounds in one process (median of per-round means, back-to-back launches on
one stream) and bit-exactness.  srhip_debug_set(5, v): -1 = old kernel, 0 = persistent (3 blocks / CU), n = grid n.
  B=32 VARS=-1,0,512,256 python tools/sweep_pers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
B = int(os.environ.get('B', '32'))
H = int(os.environ.get('H', '54'))
variants = [int(v) for v in os.environ.get('VARS', '-1,0,512,256').split(',')]
ROUNDS = int(os.environ.get('ROUNDS', '7'))
NIT = int(os.environ.get('NIT', '200'))
torch.manual_seed(0)
x64 = torch.randn(B, 64, H, H, device=dev).contiguous(memory_format=torch.channels_last)
t256 = torch.randn(B, 256, H, H, device=dev).contiguous(memory_format=torch.channels_last)
w1 = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05)     # conv1: 64 -> 256
w2 = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.05)     # conv2: 256 -> 64
b1 = torch.randn(256, device=dev) * 0.01
fl = 2.0 * B * H * H * 256 * 64 * 9
ops_ = {
    'conv1 fprop 64->256 bias+lrelu': lambda: ops.conv2d_fwd_raw(x64, w1, b1, 1, 1, 0.2),
    'conv2 dgrad 64->256 actmask': lambda: ops.conv2d_dgrad_raw(x64, w2, tuple(t256.shape), 1, 1, None, t256, 0.2),
    'conv2 fprop 256->64': lambda: ops.conv2d_fwd_raw(t256, w2, None, 1, 1, None),
    'conv1 dgrad 256->64 +skip': lambda: ops.conv2d_dgrad_raw(t256, w1, tuple(x64.shape), 1, 1, x64),
}
if os.environ.get('OPS'):
    ops_ = {k: v for k, v in ops_.items() if any(t in k for t in os.environ['OPS'].split(','))}
with ops.conv_math(os.environ.get('MODE', 'bf16x3')):
    for name, fn in ops_.items():
        lib.srhip_debug_set(5, -1)
        ref = fn().clone()
        for v in variants:
            lib.srhip_debug_set(5, v)
            y = fn()
            print('%-32s var %4d max |diff| vs one-tile kernel %.3e' % (name, v, float((y - ref).abs().max())), flush=True)
        times = {v: [] for v in variants}
        for rnd in range(ROUNDS):
            for v in variants:
                lib.srhip_debug_set(5, v)
                for _ in range(20): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(True), torch.cuda.Event(True)
                s.record()
                for _ in range(NIT): fn()
                e.record(); torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e) / NIT)
        for v in variants:
            t = sorted(times[v])
            med = t[len(t) // 2]
            print('%-32s var %4d median %.1f us  min %.1f us  %.0f TFLOP/s-eq  frac %.3f' % (name, v, med * 1e3, t[0] * 1e3, fl / med / 1e9, fl / med / 1e9 / 833.3), flush=True)
lib.srhip_debug_set(5, 0)
