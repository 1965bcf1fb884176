import sys, torch, time
sys.path.insert(0, '/root/repo')
from sradsgan_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(32, 64, 216, 216, device=dev).contiguous(memory_format=torch.channels_last)
dy = torch.randn(32, 3, 216, 216, device=dev).contiguous(memory_format=torch.channels_last)
x3 = torch.randn(32, 3, 216, 216, device=dev).contiguous(memory_format=torch.channels_last)
dy64 = torch.randn(32, 64, 216, 216, device=dev).contiguous(memory_format=torch.channels_last)
for name, fn in (('wgrad 64->3', lambda: ops.conv2d_wgrad_raw(x, dy, (3, 64, 3, 3), 1, 1, True)), ('wgrad 3->64 (+bias)', lambda: ops.conv2d_wgrad_raw(x3, dy64, (64, 3, 3, 3), 1, 1, True))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    print(name, '%.1f us' % (s.elapsed_time(e) / 20 * 1e3))
