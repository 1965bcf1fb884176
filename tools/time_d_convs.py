"""The discriminator's eight 3x3 convs at the bench shape (B = 32, 216^2 input): forward, data gradient and weight gradient of each,
isolated, back-to-back launches under HIP events; TFLOP/s of the direct convolution.  PHASES=0: the stride-2 data gradients as four
launches (rounds 1-4) instead of one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
if os.environ.get('PHASES') == '0':
    lib.srhip_debug_set(17, 0)
B = 32
layers = [(3, 64, 1, 216), (64, 64, 2, 216), (64, 128, 1, 108), (128, 128, 2, 108), (128, 256, 1, 54), (256, 256, 2, 54), (256, 512, 1, 27), (512, 512, 2, 27)]


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
for cin, cout, st, h in layers:
    ho = (h + 2 - 3) // st + 1
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, ho, ho, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    b = torch.zeros(cout, device=dev)
    gf = 2.0 * B * ho * ho * cin * cout * 9 / 1e9
    tf = timed(lambda: ops.conv2d_fwd_raw(x, w, b, st, 1))
    td = timed(lambda: ops.conv2d_dgrad_raw(dy, w, (B, cin, h, h), st, 1)) if cin >= 16 else float('nan')
    tw = timed(lambda: ops.conv2d_wgrad_raw(x, dy, (cout, cin, 3, 3), st, 1, True))
    print('%3d -> %3d s%d @%3d: fwd %6.1f us (%5.1f TF/s)  dgrad %6.1f us (%5.1f)  wgrad %6.1f us (%5.1f)   %.1f GF' % (
        cin, cout, st, h, tf, gf / tf * 1e3, td, gf / td * 1e3, tw, gf / tw * 1e3, gf), flush=True)
    tot['fwd'] += tf; tot['wgrad'] += tw
    if td == td: tot['dgrad'] += td
print('sum over the layers: fwd %.0f us, dgrad %.0f us (without the 3-channel layer), wgrad %.0f us' % (tot['fwd'], tot['dgrad'], tot['wgrad']))
