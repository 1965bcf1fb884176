import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sradsgan_ref as O
from sradsgan_amd import model as M, ops
DEV = torch.device('cuda:0')
tag = 'gen_small_x9'
x = O.det_fill('x3', (2, 3, 10, 12), 0.5, 0.5)[:1]
def run_oracle(dtype):
    om = O.GeneratorResNet(O.ResGroup, n_residual_blocks=2, n_basic_blocks=1, upscale_factor=9)
    O.det_init_(om, prefix=tag + '.')
    om = om.to(dtype)
    xo = x.clone().to(dtype).requires_grad_(True)
    yo = om(xo)
    dy = O.det_fill(tag + '.dy', tuple(yo.shape), 1.0).to(dtype)
    yo.backward(dy)
    return om, {k: p.grad.double() for k, p in om.named_parameters()}, dy
om32, g32, dy = run_oracle(torch.float32)
om64, g64, _ = run_oracle(torch.float64)
for mode in ('fp32', 'bf16x3'):
    ops.set_conv_math(mode)
    hm = M.GeneratorResNet(M.ResGroup, n_residual_blocks=2, n_basic_blocks=1, upscale_factor=9)
    hm.load_state_dict(om32.state_dict(), strict=True)
    hm.to(DEV)
    xh = x.clone().to(DEV).requires_grad_(True)
    yh = hm(xh)
    yh.backward(dy.float().to(DEV))
    print('== mode', mode)
    rows = []
    for k, p in hm.named_parameters():
        if p.grad is None: continue
        ref = g64[k]; sc = float(ref.abs().max()) + 1e-30
        e = (p.grad.cpu().double() - ref).abs()
        e32 = (g32[k] - ref).abs()
        rows.append((float(e.max()) / sc, float(e32.max()) / sc, k, int((e > 1e-4 * sc).sum()), e.numel()))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print('%.3e (oracle fp32 %.3e)  %-40s  elems>1e-4: %d / %d' % r)
