import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sradsgan_amd import ops
dev = torch.device('cuda:0'); B = 32
cl = lambda x: x.contiguous(memory_format=torch.channels_last)
tt = cl(torch.randn(B, 256, 54, 54, device=dev).clamp_min(-0.2))
w2 = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.05); b2 = torch.randn(64, device=dev) * 0.1
t_pp = ops.pp_from_f32(tt)
for _ in range(30):
    ops.conv2d_fwd_raw(tt, w2, b2, 1, 1)
    ops.conv2d_fwd_pp_raw(t_pp, w2, b2)
torch.cuda.synchronize()
