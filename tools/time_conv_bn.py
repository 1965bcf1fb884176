"""Does a BatchNorm forward (statistics pass + apply pass) run faster when the conv in front of it leaves its output in the caches?
D's block 2 / 4 / 6 / 8 (stride-2 convs on the LDS-DMA kernel, whose epilogue stores are non-temporal by default): conv + batch_norm_act
timed as a pair, with srhip_debug_set(3, 0x400) = plain epilogue stores against the default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib()


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for cin, cout, h in ((64, 64, 216), (128, 128, 108), (256, 256, 54), (512, 512, 27)):
    x = torch.randn(32, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05); b = torch.zeros(cout, device=dev)
    bn = torch.nn.BatchNorm2d(cout).to(dev).train()
    res = []
    for flag in (0, 0x400, 0, 0x400):
        lib.srhip_debug_set(3, flag)
        with torch.no_grad():
            tc = timed(lambda: ops.conv2d_fwd_raw(x, w, b, 2, 1))
            tp = timed(lambda: ops.batch_norm_act(ops.conv2d_fwd_raw(x, w, b, 2, 1), bn, 0.2))
        res.append('%s conv %.1f us, conv + BN %.1f us (BN %.1f)' % ('plain stores:' if flag else 'nt stores:   ', tc, tp, tp - tc))
    lib.srhip_debug_set(3, 0)
    print('%d -> %d s2 @%d (out %.0f MB)\n   ' % (cin, cout, h, 32 * cout * (h // 2) ** 2 * 4 / 1e6) + '\n   '.join(res), flush=True)
