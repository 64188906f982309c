"""Throughput of the C2 step when consecutive steps are issued round-robin on S caller streams (as
ppgs_amd.from_dataloader does for consecutive batches) instead of on one: the frontend, head and output kernels of one
step then run beside the layer kernels of the other.

    python tools/two_stream_steps.py [--streams 1 2 3] [--steps 200]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('--streams', type=int, nargs='+', default=[1, 2, 3])
parser.add_argument('--steps', type=int, default=200)
args = parser.parse_args()
BATCH, FRAMES = 32, 1000
model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, 'bf16')
gen = torch.Generator().manual_seed(1234)
audio = (0.1 * torch.randn(BATCH, 1, FRAMES * 160, generator=gen)).cuda()
lengths = [FRAMES] * BATCH


def step():
    return model.encode(ppgs_amd.preprocess.mel.from_audios(audio), lengths)


ref = step()
for _ in range(50):
    step()
torch.cuda.synchronize()
for count in args.streams:
    streams = [torch.cuda.Stream() for _ in range(count)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    outs = [None] * count
    for warm in range(2):
        torch.cuda.synchronize()
        start = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(streams[i % count]):
                outs[i % count] = step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - start
    same = all(torch.equal(o, ref) for o in outs if o is not None)
    print(f'{count} caller stream(s): {1e3 * elapsed / args.steps:.4f} ms/step, {BATCH * FRAMES * args.steps / elapsed / 1e6:.2f} M frames/s, outputs equal: {same}')
