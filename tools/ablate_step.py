"""Upper bounds for kernel work on the training step: the step timed with whole FAMILIES of launches removed (their C-ABI entry points
return at once: outputs stay uninitialised, so results are garbage -- timing only).  What the step gains when a family costs nothing
bounds what any optimisation of that family can gain.  Usage: python tools/ablate_step.py [family ...]   (default: all)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench
from sradsgan_amd import _hip
from sradsgan_amd.train_step import TrainStep

real = _hip.lib()
active = set()


def _conv_fwd(a): return a[14] == 2
def _conv_dgrad(a): return a[13] == 2
def _conv_wgrad(a): return a[16] == 2
def _conv_wgrad_multi(a): return a[15] == 2


RULES = {
    's2_fwd': {'srhip_conv2d_fwd': lambda a: a[14] == 2},
    's2_dgrad': {'srhip_conv2d_dgrad': lambda a: a[13] == 2},
    's2_wgrad': {'srhip_conv2d_wgrad': lambda a: a[16] == 2, 'srhip_conv2d_wgrad_multi': lambda a: a[15] == 2},
    'c3': {'srhip_conv2d_fwd': lambda a: a[10] == 3 or a[11] == 3, 'srhip_conv2d_dgrad': lambda a: a[9] == 3 or a[10] == 3,
           'srhip_conv2d_wgrad': lambda a: a[12] == 3 or a[13] == 3},
    'bn_fwd': {'srhip_bn_train_fwd': lambda a: True},
    'bn_bwd': {'srhip_bn_train_bwd': lambda a: True, 'srhip_bn_train_bwd_acc': lambda a: True, 'srhip_bn_train_bwd_bwd': lambda a: True,
               'srhip_bn_train_bwd_bwd_acc': lambda a: True},
    'lrelu_bwd': {'srhip_lrelu_bwd': lambda a: True},
    'wgrad_big': {'srhip_conv2d_wgrad': lambda a: a[14] == 3 and a[12] >= 64, 'srhip_conv2d_wgrad_multi': lambda a: True},   # every non-RAB 3x3 weight gradient
    'rab_wgrad': {'srhip_conv2d_wgrad_pp': lambda a: True},
    'pp_pass': {'srhip_pp_from_f32': lambda a: True},
}
RULES['s2_all'] = {}
for k in ('s2_fwd', 's2_dgrad', 's2_wgrad'):
    RULES['s2_all'].update(RULES[k])
RULES['bn_all'] = dict(RULES['bn_fwd']); RULES['bn_all'].update(RULES['bn_bwd'])


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name == 'srhip_adam_step':          # never update: a garbage step must not poison the weights of the next measurement
            return lambda *a: 0                   # (NaN operands also draw less power: the clock rises and the comparison is void)
        preds = [r[name] for k, r in RULES.items() if k in active and name in r]
        if not preds:
            return fn

        def stub(*a):
            for p in preds:
                if p(a):
                    return 0
            return fn(*a)
        return stub


_hip._lib = Proxy()
dev = torch.device('cuda:0')
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
gen = torch.Generator().manual_seed(1)
B = 32
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)


def run(n=12):
    for _ in range(3): step(lr, hr, al)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step(lr, hr, al)
    torch.cuda.synchronize()
    bad = sum(int(not torch.isfinite(p).all()) for p in list(G.parameters()) + list(D.parameters()))
    return (time.perf_counter() - t0) / n * 1e3, bad


for _ in range(10): step(lr, hr, al)
base, bad = run()
print('%-12s %7.2f ms per step (weights kept: no Adam step in any run; %d non-finite tensors)' % ('(nothing)', base, bad))
fams = sys.argv[1:] or ['s2_fwd', 's2_dgrad', 's2_wgrad', 's2_all', 'c3', 'bn_fwd', 'bn_bwd', 'bn_all', 'lrelu_bwd', 'wgrad_big', 'rab_wgrad', 'pp_pass']
for f in fams:
    active.clear(); active.update(f.split('+'))
    t, bad = run()
    active.clear()
    t2, bad2 = run(6)
    print('%-12s %7.2f ms per step   %+6.2f ms   (then nothing removed: %.2f; non-finite weight tensors %d)' % (f, t, t - base, t2, bad + bad2), flush=True)
