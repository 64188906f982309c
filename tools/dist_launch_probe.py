"""Does an initialised RCCL process group slow the HOST side of a launch-bound loop?  (bench.py --workload c4 with
PPGS_BENCH_FORCE_DIST=1: 625 ms per pass against 389 ms without a process group, same kernels.)

    python tools/dist_launch_probe.py        one rank; prints ms per 1000 small encodes in each state
"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, 'bf16')
feats = torch.randn(2, 80, 100).half().cuda()


def loop(tag, n=1000):
    for _ in range(50):
        model.encode(feats, [100, 100])
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(n):
        model.encode(feats, [100, 100])
    host = time.perf_counter() - start
    torch.cuda.synchronize()
    total = time.perf_counter() - start
    print(f'{tag:60s} host {1e6 * host / n:7.1f} us per encode, with the GPU {1e6 * total / n:7.1f} us', flush=True)


os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29544')
loop('no process group')
mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
if mode == 'eager':
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    loop('nccl group, device_id given (communicator built at init)')
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    loop('... after a barrier on it')
elif mode == 'lazy':
    dist.init_process_group('nccl', rank=0, world_size=1)
    loop('nccl group, lazy (no communicator yet)')
    side = dist.new_group(backend='gloo')
    dist.barrier(group=side)
    loop('... after a barrier on a gloo side group')
    t = torch.zeros(1, device='cuda')
    dist.all_reduce(t)
    torch.cuda.synchronize()
    loop('... after the first nccl collective')
elif mode == 'gloo':
    dist.init_process_group('gloo', rank=0, world_size=1)
    dist.barrier()
    loop('gloo group')
dist.destroy_process_group()
loop('process group destroyed')
