"""Achievable HBM write / copy bandwidth on this GPU with plain torch kernels
(yardstick for the store-bound projection kernels)."""
import torch

def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us

for mb in (16, 64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device='cuda')
    y = torch.empty(n, dtype=torch.float32, device='cuda')
    t_fill = timed(lambda: x.fill_(1.0))
    t_zero = timed(lambda: x.zero_())
    t_copy = timed(lambda: y.copy_(x))
    t_read = timed(lambda: x.sum())
    print(f'{mb:5d} MB: fill {t_fill:7.1f} us = {mb / 1e3 / t_fill * 1e6 / 1e3:5.2f} TB/s | memset {t_zero:7.1f} us = {mb / t_zero:5.2f} TB/s'
          f' | copy {t_copy:7.1f} us = {2 * mb / t_copy:5.2f} TB/s (r+w) | sum {t_read:7.1f} us = {mb / t_read:5.2f} TB/s')
