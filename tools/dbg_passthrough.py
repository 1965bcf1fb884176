"""Debug: does autograd add into the attention tail's passed-through gradient in place?  (SRHIP_HOLD=0 python tools/dbg_passthrough.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
DEV = torch.device('cuda:0')
n, h, w = 4, 54, 54
g = torch.Generator().manual_seed(41)
cl = lambda *s: torch.randn(*s, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
x0, u0, c1, c2 = cl(n, 64, h, w), cl(n, 64, h, w), cl(n, 64, h, w), cl(n, 64, h, w)
init = [(torch.randn(*s, generator=g) * 0.1).to(DEV) for s in [(4, 64, 1, 1), (64, 4, 1, 1), (1, 2, 7, 7), (64, 64, 1, 1), (64,)]]
seen = {}
orig = ops._AttentionTail.backward
def spy(ctx, gr):
    seen['ptr'] = gr.data_ptr(); seen['val'] = gr.clone(); seen['t'] = None
    out = orig(ctx, gr)
    seen['same_obj'] = out[1] is gr
    seen['out_ptr'] = out[1].data_ptr()
    return out
ops._AttentionTail.backward = staticmethod(spy)
big = torch.randn(8192, 8192, device=DEV)
def run(side):
    ps = [torch.nn.Parameter(t.clone()) for t in init]
    for p in ps: p.grad = torch.zeros_like(p)
    x, u = x0.clone().requires_grad_(True), u0.clone().requires_grad_(True)
    skip = x * 1.0
    y2 = skip * 2.0
    y1 = ops.attention_tail(u, skip, *ps)
    loss = (y1 * c1).sum() + (y2 * c2).sum()
    torch.cuda.synchronize()
    if side is not None:
        with torch.cuda.stream(side):
            for _ in range(6): big @ big
    with ops.direct_param_grads(side):
        loss.backward()
    if side is not None: torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print('side', side is not None, 'returned same object:', seen['same_obj'], 'x.grad shares memory with g:', x.grad.data_ptr() == seen['ptr'],
          'is g channels_last:', seen['val'].is_contiguous(memory_format=torch.channels_last))
    return [p.grad.clone() for p in ps]
a = run(None); b = run(torch.cuda.Stream())
print([bool(torch.equal(p, q)) for p, q in zip(a, b)])
