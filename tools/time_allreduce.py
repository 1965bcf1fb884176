import os, time, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
x = torch.randn(11_069_568, device='cuda'); y = torch.randn(4_702_016, device='cuda')
for _ in range(3):
    dist.all_reduce(x); dist.all_reduce(y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    hs = [dist.all_reduce(b, async_op=True) for b in (x[:8 << 20], x[8 << 20:], y)]
    for h in hs: h.wait()
torch.cuda.synchronize()
print('all-reduce of G+D gradient arenas, 1 rank: %.3f ms per iteration' % ((time.perf_counter() - t0) / 10 * 1e3))
dist.destroy_process_group()
