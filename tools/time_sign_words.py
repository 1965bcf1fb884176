"""RAB conv1 forward (64 -> 256, bias + LeakyReLU onto planes) and conv2's masked data gradient (64 -> 256 onto planes) at B = 32, 54 x 54:
the LeakyReLU mask as sign words (srhip_conv2d_fwd_pp_signs / _dgrad_pp_signs) against the hi plane of t.  Three operand sets in rotation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
dev = torch.device('cuda:0')
n, h, w = 32, 54, 54
g = torch.Generator().manual_seed(3)
w1 = (torch.randn(256, 64, 3, 3, generator=g) * 0.05).to(dev); b1 = (torch.randn(256, generator=g) * 0.1).to(dev)
w2 = (torch.randn(64, 256, 3, 3, generator=g) * 0.05).to(dev)
sets = []
for k in range(3):
    x = ops.pp_from_f32(torch.randn(n, 64, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last))
    du = ops.pp_from_f32(torch.randn(n, 64, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last))
    sets.append((x, du, ops.pp_empty(n, 256, h, w, dev), ops.pp_empty(n, 256, h, w, dev), ops.pp_sign_words(n, h, w, 256, dev)))
def timed(fn, reps=60):
    for i in range(9): fn(sets[i % 3])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(sets[i % 3])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
with ops.conv_math('bf16x3'):
    for rnd in range(3):
        a = timed(lambda s: ops.conv2d_fwd_pp_raw(s[0], w1, b1, 0.2, out_pp=s[2]))
        b = timed(lambda s: ops.conv2d_fwd_pp_raw(s[0], w1, b1, 0.2, out_pp=s[2], signs=s[4]))
        c = timed(lambda s: ops.conv2d_dgrad_pp_raw(s[1], w2, actmask=s[2], slope=0.2, out_pp=s[3]))
        d = timed(lambda s: ops.conv2d_dgrad_pp_raw(s[1], w2, slope=0.2, out_pp=s[3], signs=s[4]))
        print('conv1 fwd -> planes: %.1f us, + sign words %.1f us   conv2 dgrad: mask from t hi plane %.1f us, from sign words %.1f us' % (a, b, c, d))
