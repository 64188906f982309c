"""configs[4]: causal model, streaming 160-frame chunks, batch = 64.

    python tools/bench_streaming.py [--steps 200]

Each step is an independent forward of 64 chunks of 160 frames (what the
reference computes for this configuration: no state is carried between
chunks, positions restart).  Reports steps/s and frames/s for eager launches
and for one HIP-graph replay per step.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                              # noqa: E402

import ppgs_amd                                           # noqa: E402
from ppgs_amd import engine as E                          # noqa: E402


def timed(fn, steps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - start) / steps


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--steps', type=int, default=200)
    parser.add_argument('--batch', type=int, default=64)
    parser.add_argument('--frames', type=int, default=160)
    args = parser.parse_args()
    state = ppgs_amd.weights.seeded_state_dict(seed=1234)
    model = E.Engine(state, 0, 'bf16', is_causal=True)
    generator = torch.Generator().manual_seed(1234)
    feats = torch.randn(
        args.batch, 80, args.frames, generator=generator).half().cuda()
    lengths = [args.frames] * args.batch
    eager = timed(lambda: model.encode(feats, lengths), args.steps)
    run = model.graphed(args.batch, args.frames)
    graphed = timed(lambda: run(feats), args.steps)
    per_step = args.batch * args.frames
    for name, seconds in (('eager', eager), ('hipGraph replay', graphed)):
        print(f'{name}: {seconds * 1e6:.1f} us/step, {1 / seconds:.0f} steps/s, '
              f'{per_step / seconds / 1e6:.2f} M frames/s')


if __name__ == '__main__':
    main()
