import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sradsgan_amd import ops, _hip
DEV = torch.device('cuda:0'); lib = _hip.lib()
n, h, w = 2, 23, 37
g = torch.Generator().manual_seed(5)
cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
x = cl(torch.randn(n, 64, h, w, generator=g))
w1 = torch.nn.Parameter((torch.randn(256, 64, 3, 3, generator=g) * 0.05).to(DEV))
b1 = (torch.randn(256, generator=g) * 0.1).to(DEV)
with ops.conv_math('bf16x3'):
    t = ops.conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2)
    t_pp = ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV))
    t_ref = ops.pp_from_f32(t)
    torch.cuda.synchronize()
    guard = lib.srhip_pp_guard(w)
    for pl in (0, 1):
        a = t_pp.buf[pl].float(); b = t_ref.buf[pl].float()
        d = (a != b)
        print('plane', pl, 'mismatch', int(d.sum()), 'of', d.numel(), 'nan', int(torch.isnan(a).sum()))
        idx = d.nonzero()
        if len(idx):
            rows = idx[:, 0] - guard
            print(' rows min/max', int(rows.min()), int(rows.max()), 'chan min/max', int(idx[:, 1].min()), int(idx[:, 1].max()))
            for r, c in idx[:8].tolist():
                rr = r - guard
                print('  row', rr, '(n,h,w)=', rr // ((h + 1) * (w + 1)), (rr // (w + 1)) % (h + 1), rr % (w + 1), 'ch', c, 'got', float(a[r, c]), 'want', float(b[r, c]))
    back = ops.pp_to_f32(t_pp)
    print('max |hi+lo - t|', float((back - t).abs().max()), 'max|t|', float(t.abs().max()))
