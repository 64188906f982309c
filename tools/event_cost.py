"""What the HIP-event roofline leg of bench.py costs the timed region (GPU): 20-step blocks of the C2 step with
the layer kernel's events off / on (stride 6, 1), several blocks each, ms per step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

state = ppgs_amd.weights.seeded_state_dict(seed=1234)
model = E.Engine(state, 0, 'bf16')
audio = 0.1 * torch.randn(32, 1, 160000, generator=torch.Generator().manual_seed(1234)).cuda()
lengths = [1000] * 32


def step():
    return model.encode(ppgs_amd.preprocess.mel.from_audios(audio), lengths)


def block(steps=20):
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - start) / steps


for _ in range(30):
    step()
print('off      ', [round(block(), 4) for _ in range(4)])
for stride in (6, 1, 6):
    model.profile(True, classes=['ffn'], stride=stride)
    print(f'stride {stride} ', [round(block(), 4) for _ in range(4)], model.profile_read()['ffn'])
    model.profile(False)
print('off      ', [round(block(), 4) for _ in range(4)])
model.profile(True, classes=['ffn'], stride=6)
vals = []
for _ in range(4):
    vals.append(round(block(), 4))
    model.profile_read()
    model.profile(True, classes=['ffn'], stride=6)      # reset between blocks, as bench.py's single timed block sees it
print('stride 6, reset each block', vals)
# host time of a step (no synchronisation): is the loop host-bound?
torch.cuda.synchronize()
start = time.perf_counter()
for _ in range(200):
    step()
host = 1e3 * (time.perf_counter() - start) / 200
torch.cuda.synchronize()
print('host ms per step (enqueue only, 200 steps)', round(host, 4))
