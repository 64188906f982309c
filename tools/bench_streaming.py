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
    import json
    parser = argparse.ArgumentParser()
    parser.add_argument('--steps', type=int, default=200)
    parser.add_argument('--batch', type=int, default=64)
    parser.add_argument('--frames', type=int, default=160)
    parser.add_argument('--precision', default='bf16')
    args = parser.parse_args()
    state = ppgs_amd.weights.seeded_state_dict(seed=1234)
    model = E.Engine(state, 0, args.precision, is_causal=True)
    generator = torch.Generator().manual_seed(1234)
    feats = torch.randn(
        args.batch, 80, args.frames, generator=generator).half().cuda()
    lengths = [args.frames] * args.batch
    eager = timed(lambda: model.encode(feats, lengths), args.steps)
    run = model.graphed(args.batch, args.frames)
    # (the graph reads its own static input buffer, eager reads `feats`: both read their input where it lies)
    run.static_input.copy_(feats)
    graphed = timed(lambda: run(), args.steps)
    model.profile(True)
    for _ in range(5):
        model.encode(feats, lengths)
    torch.cuda.synchronize()
    kernels = {n: v[0] / 5 for n, v in model.profile_read().items()}
    launches = {n: v[1] / 5 for n, v in model.profile_read().items()}
    # KV-cached stream of ONE utterance (ppg_stream_*): latency of a push in steady state (frames 200 .. 480 of a
    # 500-frame window), the whole step synchronised (what a caller that consumes the posteriors sees)
    def stream_latency(n):
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            stream = model.stream(500)
            chunk = torch.randn(80, n, generator=generator).half().cuda()
            received = 0
            while received + n <= 480:
                torch.cuda.synchronize()
                start = time.perf_counter()
                stream.push(chunk)
                torch.cuda.synchronize()
                if received >= 200:
                    times.append(time.perf_counter() - start)
                received += n
        times.sort()
        return {'frames_per_push': n, 'median_us': times[len(times) // 2] * 1e6, 'p90_us': times[int(len(times) * 0.9)] * 1e6,
                'pushes_timed': len(times)}
    kv = [stream_latency(n) for n in (16, 48, 160)]

    # the same for `batch` utterances advanced together (ppg_stream_push_batch: ONE launch sequence per step)
    def batched_latency(n):
        torch.cuda.synchronize()
        times = []
        for _ in range(2):
            stream = model.batched_stream(args.batch, 500)
            chunk = torch.randn(args.batch, 80, n, generator=generator).half().cuda()
            received = 0
            while received + n <= 480:
                torch.cuda.synchronize()
                start = time.perf_counter()
                stream.push(chunk)
                torch.cuda.synchronize()
                if received >= 200:
                    times.append(time.perf_counter() - start)
                received += n
        times.sort()
        return {'streams': args.batch, 'frames_per_push': n, 'median_us': times[len(times) // 2] * 1e6,
                'p90_us': times[int(len(times) * 0.9)] * 1e6, 'pushes_timed': len(times),
                'frames_per_s': args.batch * n / times[len(times) // 2]}
    kvb = [batched_latency(n) for n in (16, 48, 160)]
    per_step = args.batch * args.frames
    # single 160-frame windows: 13 414 400 FLOP per frame + 5120 Tc^2 per window (causal
    # attention computes about half of the Tc^2 term)
    flops = args.batch * (13_414_400 * args.frames + 5120 * args.frames ** 2)
    layer_flops = (4 * 256 * 2048 + 2 * 256 * 256 + 6 * 256 * 256 * 4 / 5) * per_step
    layer_ms = kernels['ffn'] / max(launches['ffn'], 1)
    peak = 157.3 if args.precision == 'fp32' else 2500.0
    print(json.dumps({
        'config': f'configs[4]: causal_transformer, streaming {args.frames}-frame chunks, batch = {args.batch}, 1 MI355X, '
                  'one hipGraph replay per step (each chunk an independent forward, as in the reference)',
        'dtype': args.precision,
        'eager': {'us_per_step': eager * 1e6, 'steps_per_s': 1 / eager, 'frames_per_s': per_step / eager},
        'hipgraph_replay': {'us_per_step': graphed * 1e6, 'steps_per_s': 1 / graphed, 'frames_per_s': per_step / graphed},
        'end_to_end_tflops': flops / graphed / 1e12,
        'kernel_ms_per_step': kernels,
        'kv_cached_stream': {'what': 'one utterance, K / V^T and residual rows of all layers cached on the device, 14 + 4 x 5 '
                                     'launches per push on row-mapped launches of the token-split kernels; wall time of push() + synchronize', 'steps': kv},
        'kv_cached_batched_stream': {'what': f'{args.batch} utterances advanced together, each with its own K / V^T cache and frontier: ONE '
                                             'launch sequence (24 launches) per step for all of them; wall time of push() + synchronize '
                                             '(includes the per-item result copies)', 'steps': kvb},
        'roofline': {'kernel': 'layer kernel launches (10 240 token rows: layer32 kernel, sub-tile workgroups of two token blocks)',
                     'bound': 'mfma', 'achieved': layer_flops / layer_ms / 1e9, 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': layer_flops / layer_ms / 1e9 / peak, 'mean_launch_ms': layer_ms},
    }))


if __name__ == '__main__':
    main()
