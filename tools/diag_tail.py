"""debug: srhip_attn_tail_bwd_g twice on the same inputs (NaN-poisoned buffers): which intermediate differs between runs?"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
from sradsgan_amd.ops import _p, _stream
DEV = torch.device('cuda:0')
n, h, w, c, hid = 32, 54, 54, 64, 4
gen = torch.Generator().manual_seed(5)
cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
u, skip, g = (cl(torch.randn(n, 64, h, w, generator=gen)) for _ in range(3))
fc1, fc2, w7, wc, bc = [t.to(DEV) for t in ((torch.randn(4, 64, 1, 1, generator=gen) * 0.2), (torch.randn(64, 4, 1, 1, generator=gen) * 0.2),
                                          (torch.randn(1, 2, 7, 7, generator=gen) * 0.1), (torch.randn(64, 64, 1, 1, generator=gen) * 0.1), (torch.randn(64, generator=gen) * 0.1))]
lib = _hip.lib()
with torch.no_grad():
    out, saved = ops._tail_forward(u, skip, fc1, fc2, w7, wc, bc)
avg, mx, arg, s, pooled, argc, m = saved
npix = n * h * w
def run():
    f32 = dict(device=DEV, dtype=torch.float32)
    du = torch.full((n, h, w, c), float('nan'), **f32)
    ws = torch.full((lib.srhip_attn_tail_bwd_fused_workspace(n, h, w, hid) // 4,), float('nan'), **f32)
    dupp = ops.pp_empty(n, c, h, w, DEV)
    dw7, dfc1, dfc2 = torch.empty(98, **f32), torch.empty(hid * 64, **f32), torch.empty(hid * 64, **f32)
    _hip.check(lib.srhip_attn_tail_bwd_g(_p(g), _p(wc.contiguous()), _p(u), _p(s), _p(m), _p(pooled), _p(argc), _p(avg), _p(mx), _p(arg),
                                         _p(w7.contiguous()), _p(fc1.contiguous()), _p(fc2.contiguous()), _p(du), _p(dupp.buf), _p(dw7), 0, _p(dfc1), _p(dfc2), 0,
                                         _p(ws), ws.numel() * 4, n, h, w, c, hid, _stream()), 'bwd_g')
    torch.cuda.synchronize()
    return dict(du=du, da=ws[:npix].clone(), dpooled=ws[npix:3 * npix].clone(), dsp=ws[3 * npix:3 * npix + n * 48 * 64].clone(), dw7=dw7, dfc1=dfc1, dfc2=dfc2)
ref = run()
print('nan in ref:', {k: int(torch.isnan(v).sum()) for k, v in ref.items()})
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    r = run()
    bad = {k: (int((r[k] != ref[k]).sum()), float((r[k] - ref[k]).abs().max())) for k in ref if not torch.equal(r[k], ref[k])}
    if 'dsp' in bad:
        d = (r['dsp'] != ref['dsp']).view(n, 48, 64)
        idx = d.nonzero()[:6].tolist()
        bad['dsp_where'] = idx
    if 'du' in bad:
        d = (r['du'] != ref['du']).view(n, h * w, c)
        bad['du_where'] = d.nonzero()[:6].tolist()
    if bad: print('trial', t, bad)
    if 'du_where' in bad:
        b_, p_, c0 = bad['du_where'][0]
        comp = c0 % 4
        ch = torch.arange(comp, 64, 4, device=DEV)
        gz_scale = (s[b_, ch] * m[b_ * h * w + p_])
        dgz = ((r['du'].view(n, h * w, c)[b_, p_, ch] - ref['du'].view(n, h * w, c)[b_, p_, ch]) / gz_scale).double().cpu()
        grow = g.permute(0, 2, 3, 1).reshape(n, h * w, c)[b_, p_].float().cpu()
        W = wc.view(64, 64).float().cpu()                      # [k][c]
        gh = grow.bfloat16().float(); gl = (grow - gh).bfloat16().float()
        Wh = W.bfloat16().float(); Wl = (W - Wh).bfloat16().float()
        chc = ch.cpu()
        print('   pixel', b_, p_, 'component', comp, 'delta gz:', [round(float(x), 6) for x in dgz[:8]])
        dpl = ref['dpooled'].view(-1, 2)[b_ * h * w + p_]
        mm_ = float(m[b_ * h * w + p_])
        print('   dp.x/64/mm %.6f  dp.y/mm %.6f  argc %d  mm %.6f; all 16 deltas equal: %s' % (float(dpl[0]) / 64 / mm_, float(dpl[1]) / mm_, int(argc[b_ * h * w + p_]), mm_, bool((dgz - dgz[0]).abs().max() < 1e-6)))
        print('   delta du (bad - ref) raw:', [float(x) for x in (r['du'].view(n, h * w, c)[b_, p_, ch] - ref['du'].view(n, h * w, c)[b_, p_, ch])[:6]])
        break
        for ks in range(4):
            k = slice(ks * 16, ks * 16 + 16)
            for name, a_, b2 in (('al*bh', gl, Wh), ('ah*bl', gh, Wl), ('ah*bh', gh, Wh)):
                part = (a_[k, None].double() * b2[k][:, chc].double()).sum(0)
                print('      ks %d %s' % (ks, name), [round(float(x), 6) for x in part[:8]])
            for half in range(2):
                kk = slice(ks * 16 + half * 8, ks * 16 + half * 8 + 8)
                part = (gh[kk, None].double() * Wh[kk][:, chc].double()).sum(0)
                print('      ks %d ah*bh k-half %d' % (ks, half), [round(float(x), 6) for x in part[:8]])
        break
