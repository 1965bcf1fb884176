"""Round 4: K-split form of the 64-wide one-tile patch kernel (conv_patch_ks_kernel, default) against the 2 x 2 wave grid of
rounds 1-3 (srhip_debug_set(10, 0)) on the RAB convs with 64 destination channels: conv2's fprop (256 -> 64, bias) and conv1's
dgrad (256 -> 64, + skip gradient), at the training and the inference batch; agreement and interleaved timing.
  python tools/sweep_ks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
CL = torch.channels_last
ROUNDS, NIT = int(os.environ.get('ROUNDS', '5')), int(os.environ.get('NIT', '200'))
torch.manual_seed(0)
mk = lambda n, c, h, w: torch.randn(n, c, h, w, device=dev).contiguous(memory_format=CL)
with ops.conv_math('bf16x3'):
    for B in (32, 16):
        t256, x64, g64 = mk(B, 256, 54, 54), mk(B, 64, 54, 54), mk(B, 64, 54, 54)
        w2 = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.05)
        w1 = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05)
        b2 = torch.randn(64, device=dev) * 0.01
        fl = 2.0 * B * 54 * 54 * 256 * 64 * 9
        fns = {'conv2 fprop 256->64 +bias': lambda: ops.conv2d_fwd_raw(t256, w2, b2, 1, 1, None),
               'conv1 dgrad 256->64 +skip': lambda: ops.conv2d_dgrad_raw(t256, w1, tuple(x64.shape), 1, 1, g64)}
        for name, fn in fns.items():
            lib.srhip_debug_set(10, 0); ref = fn().clone()
            lib.srhip_debug_set(10, 1); got = fn().clone()
            print('B=%d %-28s max |diff| %.3e  (max |ref| %.3e)' % (B, name, float((ref - got).abs().max()), float(ref.abs().max())), flush=True)
            t = {0: [], 1: []}
            for _ in range(ROUNDS):
                for v in (0, 1):
                    lib.srhip_debug_set(10, v)
                    for _ in range(20): fn()
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
                    s.record()
                    for _ in range(NIT): fn()
                    e.record(); torch.cuda.synchronize()
                    t[v].append(s.elapsed_time(e) / NIT)
            for v in (0, 1):
                tt = sorted(t[v]); med = tt[len(tt) // 2]
                print('B=%d %-28s %-22s median %.1f us  min %.1f  frac %.3f' % (B, name, 'K-split (default)' if v else '2x2 waves (key 10 = 0)', med * 1e3, tt[0] * 1e3, fl / med / 1e9 / 833.3), flush=True)
lib.srhip_debug_set(10, 1)
