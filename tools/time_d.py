import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import model as M, ops
dev = torch.device('cuda:0')
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
x = torch.rand(32, 3, 216, 216, device=dev)
for att in (True, False):
    D = M.Discriminator(attention=att).to(dev)
    def fwd_bwd():
        xx = x.clone().requires_grad_(True)
        D(xx).mean().backward()
    def gp():
        xx = x.clone().requires_grad_(True)
        d = D(xx)
        with ops.no_param_grads():
            (g,) = torch.autograd.grad(d, xx, torch.ones_like(d), create_graph=True)
        ops.gp_penalty(g).backward()
    print('attention=%s  fwd+bwd %.2f ms   gp (fwd + double bwd) %.2f ms' % (att, timeit(fwd_bwd), timeit(gp)), flush=True)
