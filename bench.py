"""Benchmark of the PPG hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of the whole hot path over one batch of synthetic input
that is already resident in HBM: mel frontend (STFT + mel + log) -> 5-layer
transformer encoder with the reference's 500/50 chunking -> per-frame softmax,
for BASELINE.json configs[1]: mel representation, batch = 32 x 1000 frames
(32 x 160000 samples of 16 kHz audio), bf16 MFMA arithmetic.  Weights are a
seeded random checkpoint of the reference architecture (no network).

N > 1 (launched by torch.distributed.run, one rank per GPU): utterances are
independent, so every rank runs its own 32 x 1000 batch (weak scaling, no
data-path collective); value = frames of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused
FFN): algorithmic FLOPs per launch / its mean launch duration measured with
HIP events on the launch stream inside the timed region.  `cpu_baseline` is
the CPU oracle (fp32 restatement of the reference path, proven equal to the
reference modules by tests/test_oracle_golden.py) timed on the host cores on
a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                                             # noqa: E402
import torch.distributed as dist                         # noqa: E402

BATCH = 32
FRAMES = 1000
SAMPLES = FRAMES * 160
EVENT_STRIDE = 6               # roofline leg: HIP events around every 6th layer-kernel launch of the timed region
PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3


def parse():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=50)
    parser.add_argument('--warmup', type=int, default=10)
    parser.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    parser.add_argument('--cpu-seconds', type=float, default=12.0,
                        help='budget of the CPU-baseline leg')
    parser.add_argument('--no-cpu', action='store_true')
    return parser.parse_args()


def cpu_baseline(state, seconds):
    """Oracle (CPU port of the reference path, fp32, autocast off) on a
    bounded sample: batches of 4 x 160000 samples for about `seconds`, at the
    torch thread count that is fastest on this host (the default -- one thread
    per logical core -- oversubscribes the small GEMMs)."""
    from oracle import ppg_oracle
    generator = torch.Generator().manual_seed(1234)
    audio = 0.1 * torch.randn(4, 1, SAMPLES, generator=generator)
    default_threads = torch.get_num_threads()
    best = (float('inf'), default_threads)
    for threads in sorted({default_threads, 64, 32, 16}):
        if threads > default_threads:
            continue
        torch.set_num_threads(threads)
        ppg_oracle.from_audio(state, audio[:1, :, :16000])      # warm-up
        start = time.perf_counter()
        ppg_oracle.from_audio(state, audio[:2])
        best = min(best, (time.perf_counter() - start, threads))
    torch.set_num_threads(best[1])
    start = time.perf_counter()
    batches = 0
    while True:
        ppg_oracle.from_audio(state, audio)
        batches += 1
        elapsed = time.perf_counter() - start
        if elapsed > seconds or batches >= 64:
            break
    torch.set_num_threads(default_threads)
    return {
        'value': batches * 4 * FRAMES / elapsed,
        'unit': 'frames/s',
        'cores': best[1],
        'kind': 'port',
        'sample': f'{batches} batches of 4 x {FRAMES} frames '
                  f'(mel frontend + encoder + softmax, fp32 CPU oracle), '
                  f'{elapsed:.1f} s with {best[1]} torch threads on '
                  f'{os.cpu_count()} logical cores',
    }


def pmc_traffic(kernel='ffn_'):
    """HBM-side bytes per launch of the dominant kernel from the newest
    committed rocprofv3 PMC summary (profiles/r*_pmc_summary.txt; separate
    --pmc passes of this same command): 2 x FETCH_SIZE (gfx950 reports half of
    a wide coalesced read, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.txt')))
    if not files:
        return None
    fetch = write = None
    lines = [line for line in open(files[-1]) if line.startswith(kernel)]
    # several variants of the kernel in one run: the one with the Q/K/V tail
    # (4 of the 5 launches of a step) is the one the roofline line describes
    tail = [line for line in lines if 'true>' in line.split(':')[0]]
    for line in tail or lines:
        if line.startswith(kernel):
            m = re.search(r'FETCH_SIZE=([0-9.e+]+)', line)
            fetch = float(m.group(1)) if m else fetch
            m = re.search(r'WRITE_SIZE=([0-9.e+]+)', line)
            write = float(m.group(1)) if m else write
    if fetch is None or write is None:
        return None
    return (2.0 * fetch + write) * 1024.0


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU path')
    torch.cuda.set_device(local_rank)
    # (PPGS_BENCH_FORCE_DIST=1: exercise the RCCL barrier / max-over-ranks path with one rank)
    use_dist = world > 1 or bool(os.environ.get('PPGS_BENCH_FORCE_DIST'))
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))

    import ppgs_amd
    from ppgs_amd import data, engine as E

    state = ppgs_amd.weights.seeded_state_dict(seed=1234)
    model = E.Engine(state, local_rank, args.precision)
    generator = torch.Generator().manual_seed(1234 + rank)
    audio = (0.1 * torch.randn(BATCH, 1, SAMPLES, generator=generator)).cuda()
    lengths = [FRAMES] * BATCH

    def step():
        mel = ppgs_amd.preprocess.mel.from_audios(audio)
        return model.encode(mel, lengths)

    out = step()                         # setup: window plan built and uploaded, workspace allocated
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    assert out.shape == (BATCH, 40, FRAMES) and bool(torch.isfinite(out).all())

    # timed region: HIP events only around the dominant kernel (roofline leg)
    # (every 6th launch: 6 is coprime with the 5 layers, so the samples rotate
    # through the five launches of a step; two event records per launch cost
    # ~1.5 us each on the stream, 1.5 % of the step if every launch is timed)
    model.profile(True, classes=['ffn'], stride=EVENT_STRIDE)
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    elapsed = time.perf_counter() - start
    if use_dist:
        worst = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        elapsed = float(worst.item())
    ffn_ms, ffn_samples = model.profile_read()['ffn']
    # untimed extra pass: per-kernel-class breakdown (events around every launch)
    breakdown_steps = 5
    model.profile(True)
    E.frontend_profile(local_rank, True)
    for _ in range(breakdown_steps):
        step()
    torch.cuda.synchronize()
    kernels = model.profile_read()
    kernels['frontend'] = E.frontend_profile_read(local_rank)
    model.profile(False)
    E.frontend_profile(local_rank, False)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        frames_per_s = world * BATCH * FRAMES * args.steps / elapsed
        _, info = E.plan_windows(BATCH, FRAMES, lengths)
        hidden, ffn = 256, 2048
        # FLOPs of ONE launch: the batch's FFN work of one layer, divided by the
        # launches per layer (>1 when the engine splits the batch over streams)
        launches_per_layer = max(kernels['ffn'][1] // (5 * breakdown_steps), 1)
        # (FFN 2*2*H*F per processed frame; the attention
        # out-projection's 2*H*H ride along when the engine fuses it into the
        # same kernel -- then no separate out-proj launch shows up)
        # and so does the NEXT layer's Q/K/V projection (6*H*H) in all but the last
        # layer's launch when that is fused as the kernel's tail (then only layer
        # 0 launches a Q/K/V kernel of its own); flops_per_launch is the mean
        # over the five launches of a step
        layers = 5
        op_fused = kernels['outproj_ln'][1] == 0
        qkv_own = kernels['qkv'][1] / breakdown_steps / launches_per_layer    # stand-alone Q/K/V launches per step
        qkv_fused_layers = max(layers - qkv_own, 0.0) if op_fused else 0.0
        flops_per_frame = (4.0 * hidden * ffn + (2.0 * hidden * hidden if op_fused else 0.0)
                           + 6.0 * hidden * hidden * qkv_fused_layers / layers)
        ffn_flops = flops_per_frame * info.processed_frames / launches_per_layer
        ffn_tflops = ffn_flops / (1e-3 * ffn_ms / max(ffn_samples, 1)) / 1e12
        peak = PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_FP32_TFLOPS
        step_flops = BATCH * data.flops(FRAMES)
        line = {
            'metric': 'PPG frames/sec (whole node), mel repr, batch=32x1000 frames',
            'value': frames_per_s,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.precision,
            'data': 'synthetic 16 kHz audio (0.1*randn, seed 1234), '
                    'seeded random weights of the reference architecture',
            'config': {
                'workload': 'configs[1]: mel representation, batch=32 x 1000 '
                            'frames per GPU, audio resident in HBM -> '
                            '(32,40,1000) fp32 posteriors in HBM',
                'batch': BATCH, 'frames': FRAMES, 'per_gpu_batches': 1,
                'parallelism': f'utterance-sharded x{world}, no data-path collective',
            },
            'roofline': {
                'kernel': ('ffn_mixed_kernel (the layer kernel at this shape: out-proj+residual+LN1, W1+ReLU+W2+residual+LN2'
                           + (', next layer Q/K/V)' if qkv_fused_layers else ')') if op_fused
                           else 'ffn_kernel (fused W1+ReLU+W2+residual+LayerNorm)'),
                'bound': 'mfma',
                'achieved': ffn_tflops,
                'peak': peak,
                'unit': 'TFLOP/s',
                'frac': ffn_tflops / peak,
                'traffic': pmc_traffic(),
                'flops_per_launch': ffn_flops,
                'mean_launch_ms': ffn_ms / max(ffn_samples, 1),
                'timed_launches': ffn_samples,
            },
            'end_to_end_tflops': step_flops * args.steps / elapsed / 1e12,
            'end_to_end_mfma_frac': step_flops * args.steps / elapsed / 1e12 / peak,
            'kernel_ms_per_step': {
                k: v[0] / breakdown_steps for k, v in kernels.items()},
        }
        if world == 1 and not args.no_cpu:
            line['cpu_baseline'] = cpu_baseline(state, args.cpu_seconds)
            line['speedup_vs_cpu'] = frames_per_s / line['cpu_baseline']['value']
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
