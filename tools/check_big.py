"""Default kernels vs the forced kernel families vs exact-fp32 mode at bench-like sizes (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
B = int(os.environ.get('B', '12'))
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
shapes = [(64, 256, 54, 1), (256, 64, 54, 1), (64, 64, 54, 1), (64, 256, 108, 1), (64, 64, 216, 2), (128, 256, 54, 2), (256, 512, 27, 1)]
for cin, cout, hw, stride in shapes:
    torch.manual_seed(1)
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    b = torch.randn(cout, device=dev) * 0.1
    ho = (hw + 2 - 3) // stride + 1
    dy = torch.randn(B, cout, ho, ho, device=dev).contiguous(memory_format=torch.channels_last)
    def run():
        y = ops.conv2d_fwd_raw(x, w, b, stride, 1, 0.2)
        dx = ops.conv2d_dgrad_raw(dy, w, tuple(x.shape), stride, 1)
        dw, db = ops.conv2d_wgrad_raw(x, dy, tuple(w.shape), stride, 1, True)
        return [t.clone() for t in (y, dx, dw, db)]
    with ops.conv_math('fp32'):
        ref = run()
    base = run()
    out = {}
    for name, dbg in (('cfg21', ((0, 21),)), ('cfg-1', ((0, -1),)), ('wcfg7', ((1, 7),))):
        for k, v in dbg: lib.srhip_debug_set(k, v)
        out[name] = run()
        for k, v in dbg: lib.srhip_debug_set(k, 0)
    line = '%3d->%3d @%3d s%d  default-vs-fp32: %s' % (cin, cout, hw, stride, ' '.join('%.1e' % rel(a, r) for a, r in zip(base, ref)))
    for name in out:
        line += '  | %s-vs-default: %s' % (name, ' '.join('%.1e' % rel(a, r) for a, r in zip(out[name], base)))
    print(line, flush=True)
    for name in out:
        for t in out[name]:
            assert torch.isfinite(t).all(), name
