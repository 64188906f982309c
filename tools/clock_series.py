"""Time series of 20-step blocks of the C2 step (ms per step of each block against the time since the first launch):
what a short timed region sees of the chip's clock management.

    python tools/clock_series.py            cold start (2 s idle), 60 blocks back to back, then blocks preceded by
                                            idle gaps of 0.05 ... 20 ms (5 blocks after each gap)
"""
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


step()
torch.cuda.synchronize()
time.sleep(2.0)                                    # cold start: the chip idles
series = [round(block(), 4) for _ in range(60)]
print('cold start, 20-step blocks back to back:')
for i in range(0, len(series), 10):
    print('  ', series[i:i + 10])
print('idle gap before a block -> ms/step of the next five 20-step blocks:')
for gap_ms in (0.05, 0.2, 0.5, 1, 2, 5, 10, 20, 100):
    for _ in range(10):
        block()
    time.sleep(gap_ms * 1e-3)
    print(f'   {gap_ms:6.2f} ms:', [round(block(), 4) for _ in range(5)])
print('idle gap before a block -> ms/step of the next 5-step blocks (3.7 ms each):')
for gap_ms in (0.05, 1, 5, 20):
    for _ in range(10):
        block()
    time.sleep(gap_ms * 1e-3)
    print(f'   {gap_ms:6.2f} ms:', [round(block(5), 4) for _ in range(8)])
