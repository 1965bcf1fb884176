"""Order in which the BatchNorm nodes of one training step run their backward (forward node vs second-order node, by tensor): does
the second-order node of a layer always run before the forward node of the same layer?  (debug aid for ops.bn second-order folding)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd import ops
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
gen = torch.Generator().manual_seed(1)
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
step(lr, hr, al)
log = []
f0, b0 = ops._BNTrainFwd.backward, ops._BNTrainBwd.backward
def fwd_bwd(ctx, dy):
    x = ctx.saved_tensors[0]
    log.append(('fwd-node', tuple(x.shape), x.data_ptr(), torch.is_grad_enabled(), torch.cuda.current_stream().cuda_stream))
    return f0(ctx, dy)
def bwd_bwd(ctx, ddx, ddg, ddb):
    x = ctx.saved_tensors[1]
    log.append(('2nd-node', tuple(x.shape), x.data_ptr(), ddx is not None, torch.cuda.current_stream().cuda_stream))
    return b0(ctx, ddx, ddg, ddb)
ops._BNTrainFwd.backward = staticmethod(fwd_bwd)
ops._BNTrainBwd.backward = staticmethod(bwd_bwd)
step(lr, hr, al)
torch.cuda.synchronize()
for i, r in enumerate(log):
    print(i, r)
