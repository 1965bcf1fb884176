import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import model as M, ops
dev = torch.device('cuda:0')
x = torch.rand(32, 3, 216, 216, device=dev)
D = M.Discriminator().to(dev)
for _ in range(3):
    xx = x.clone().requires_grad_(True)
    d = D(xx)
    with ops.no_param_grads():
        (g,) = torch.autograd.grad(d, xx, torch.ones_like(d), create_graph=True)
    ops.gp_penalty(g).backward()
torch.cuda.synchronize()
