"""Benchmark of the PPG hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4] [--precision bf16|fp16|fp32]

Default workload = BASELINE.json configs[1] ("c2"): one step = one pass of the
whole hot path over one batch of synthetic input that is already resident in
HBM -- mel frontend (STFT + mel + log) -> 5-layer transformer encoder with the
reference's 500/50 chunking -> per-frame softmax -- for mel representation,
batch = 32 x 1000 frames (32 x 160000 samples of 16 kHz audio), bf16 MFMA
operands with fp32 accumulation.  Weights are a seeded random checkpoint of
the reference architecture (no network).

N > 1: one rank per GPU.  Launched by ``python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`` the ranks come from the environment;
launched bare (``python bench.py --gpus N``) the script spawns the N ranks
itself.  Either way --gpus must equal the world size and the node must have
that many GPUs, or the run fails.  Utterances are independent, so every rank
runs its own 32 x 1000 batch (weak scaling, no data-path collective); value =
frames of all ranks / max-over-ranks time of exactly K steps between two
barrier + synchronize brackets.

``--workload c4`` = configs[3]: 10 000 utterances of randint(50, 3001) frames
(seed 1234), packed under max_frames = 32000, LPT-sharded over the ranks; every
rank generates and holds ONLY its shard; one step = one pass over the shard
(frontend + encoder per packed batch); the final gatherv of the posteriors to
rank 0 (RCCL) is timed separately (``gather_ms``).

Between the W warm-up steps and the K timed steps the same step runs untimed for
``--prewarm-s`` seconds (default 1.0, reported as ``prewarm_s``): the metric is the
steady-state loop, and a GPU that was idle for a millisecond spends its next ~16 ms of
load below its sustained clocks (tools/clock_series.py; round 2: the first 20-step
block 0.837 ms/step, the next two 0.771 / 0.765 on the same box).  ``--prewarm-s 0``
with ``--warmup 5`` times that ramp.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the
layer kernel): algorithmic FLOPs per launch / its mean launch duration
measured with HIP events on the launch stream inside the timed region, priced
against the WHOLE chip's dense MFMA peak (`roofline.frac`).
The engine runs a batch of this size as TWO pipelines (half-batches on two HIP
streams; the same bits at this batch): a layer launch is then 128 one-per-CU workgroups
beside the other pipeline's kernels, so one launch can reach at most half the chip's
peak (`frac_of_occupied_cus` prices it against the CUs it occupies); the step as a
whole is `end_to_end_mfma_frac`, and `alt_single_pipeline` is the same step with
PPGS_AMD_STREAMS=1, where a launch has the whole chip to itself; `alt_graph_replay` is the
same step with the encoder's launches as one hipGraph replay (bit-equal; the step is not launch-bound).
`cpu_baseline` is the CPU oracle (fp32 restatement of the reference path,
proven equal to the reference modules by tests/test_oracle_golden.py) timed on
the host cores on the same 32 x 1000 batch, plus its bf16-autocast variant
(the arithmetic the reference ships with on CPU).
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
PEAK_16BIT_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA peak, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3
HIDDEN, FFN, LAYERS = 256, 2048, 5
# where a rank's tensors live.  'cuda' always; tests/test_dist_gloo.py sets 'cpu' to walk rank_main's N-rank
# bookkeeping (sharding, barriers, the gatherv, the record's fields) over gloo with a stub engine -- no such record is a
# measurement, and the engine itself has no CPU path
DEVICE = 'cuda'


# PPGS_AMD_* switches that select a CONFIGURATION of the same computed work (recorded in the line as `env_switches`);
# every other PPGS_AMD_* variable is an experiment / ablation switch of the engine (kernel selection, phase skipping in
# debug builds, timing dumps): with one of those set this script refuses to produce a line (--allow-ablation: it
# prints the line with `value` nulled and the variables under `ablation_env`)
CONFIG_SWITCHES = {'PPGS_AMD_STREAMS', 'PPGS_AMD_LIB', 'PPGS_AMD_BIND_CPUS', 'PPGS_AMD_SYSFS_ROOT', 'PPGS_AMD_CHECKPOINT'}


def env_guard(args):
    """-> (config switches set, ablation switches set); exits when an ablation switch is set without --allow-ablation"""
    mine = {k: v for k, v in os.environ.items() if k.startswith('PPGS_AMD_')}
    config = {k: v for k, v in mine.items() if k in CONFIG_SWITCHES}
    ablation = {k: v for k, v in mine.items() if k not in CONFIG_SWITCHES}
    if ablation and not args.allow_ablation:
        raise SystemExit(f'bench.py: experiment switches are set ({ablation}); a line measured under them is not the '
                         'benchmark -- unset them, or pass --allow-ablation for a line with `value` nulled')
    return config, ablation


class ClockSampler:
    """Shader clock of the GPU while a block of steps runs, from the driver's sysfs file (pp_dpm_sclk: the line with
    the `*` is the current level), read by a thread every few milliseconds; best effort -- None where the container
    does not show the file.  The 2.5 PF peak is quoted at 2.4 GHz; `mfma_busy_frac` divides by that nominal clock."""

    def __init__(self, device_index):
        import glob
        # the card whose PCI address is the torch device's (a box shows every GPU of the node in sysfs, one of them in HIP)
        self.path = None
        try:
            props = torch.cuda.get_device_properties(device_index)
            want = f'{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}'
            for path in sorted(glob.glob('/sys/class/drm/card*/device')):
                if os.path.basename(os.path.realpath(path)).startswith(want) and os.path.exists(path + '/pp_dpm_sclk'):
                    self.path = path + '/pp_dpm_sclk'
                    break
        except (AttributeError, OSError):
            pass
        self.samples = []
        self._stop = False
        self._thread = None

    def _read(self):
        import re
        try:
            with open(self.path) as f:
                text = f.read()
            m = re.search(r'(\d+)\s*Mhz\s*\*', text, re.I)
            return float(m.group(1)) if m else None
        except OSError:
            return None

    def __enter__(self):
        import threading
        if self.path is not None:
            def loop():
                while not self._stop:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    time.sleep(0.0005)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join()

    def summary(self):
        if not self.samples:
            return None
        ordered = sorted(self.samples)
        return {'mean_mhz': sum(ordered) / len(ordered), 'min_mhz': ordered[0], 'max_mhz': ordered[-1],
                'median_mhz': ordered[len(ordered) // 2], 'samples': len(ordered), 'source': self.path}


def dense_mfma_probe():
    """tools/bin/mfma_clock_probe (stand-alone HIP program, built by tools/probes/build.sh): v_mfma_f32_32x32x16_bf16 back
    to back on every SIMD of the chip -- the clock the chip DELIVERS with all matrix pipes busy (s_memtime cycles per
    s_memrealtime microsecond) and the FLOP/s that is.  The 2.5 PFLOP/s every `frac` here divides by is quoted at
    2.4 GHz; under its power limit the chip runs such a loop at ~1.9 GHz.  Reported beside the line, never used in it.
    -> dict or None (binary absent / failed)."""
    import re
    import subprocess
    path = os.path.join(ROOT, 'tools', 'bin', 'mfma_clock_probe')
    if not os.path.exists(path):
        return None
    try:
        text = subprocess.run([path, '400'], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.TimeoutExpired):
        return None
    return parse_mfma_probe(text)


def parse_mfma_probe(text):
    import re
    rows = {}
    for line in text.splitlines():
        m = re.match(r'(.+?)\s+grid\s+(\d+)\s+([0-9.]+) us\s+clock GHz p10/50/90 ([0-9.]+) ([0-9.]+) ([0-9.]+)\s+cycles per MFMA p50 ([0-9.]+)\s+([0-9.]+) TFLOP/s', line)
        if m:
            rows[m.group(1).strip()] = {'clock_ghz_p50': float(m.group(5)), 'cycles_per_mfma': float(m.group(7)), 'tflops': float(m.group(8))}
    dense = rows.get('MFMA only, 1 wave per SIMD, 10x longer') or rows.get('MFMA only, 1 wave per SIMD')
    if not dense:
        return None
    return {'what': 'stand-alone dense v_mfma_f32_32x32x16_bf16 loop on every SIMD (tools/probes/mfma_clock_probe.hip), run after the '
                    'timed region: what the chip delivers under its power limit with all matrix pipes busy',
            'dense': dense, 'with_3_valu_per_mfma': rows.get('MFMA + 3 VALU, 1 wave per SIMD'),
            'with_6_valu_per_mfma': rows.get('MFMA + 6 VALU, 1 wave per SIMD'), 'half_the_cus': rows.get('MFMA only, half the CUs')}


def parse(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=None)
    parser.add_argument('--warmup', type=int, default=None)
    parser.add_argument('--workload', default='c2', choices=['c2', 'c4'])
    parser.add_argument('--precision', default='bf16', choices=['bf16', 'fp16', 'fp32', 'fp16x2'])
    parser.add_argument('--utterances', type=int, default=10000, help='c4: corpus size')
    parser.add_argument('--cpu-seconds', type=float, default=20.0,
                        help='budget of the CPU-baseline leg')
    parser.add_argument('--no-cpu', action='store_true')
    parser.add_argument('--allow-ablation', action='store_true',
                        help='run although PPGS_AMD_* experiment switches are set; the line then carries them under '
                             '`ablation_env` and its `value` is null')
    parser.add_argument('--no-alt', action='store_true', help='skip the fp16-operand leg of c2')
    parser.add_argument('--prewarm-s', type=float, default=1.0,
                        help='c2: seconds of the same step run untimed after the W warm-up steps and before the '
                             'timed K steps (the metric is the steady-state loop; a GPU that has been idle '
                             'takes ~16 ms of load to reach its sustained clocks); reported as prewarm_s, 0 disables')
    args = parser.parse_args(argv)
    if args.steps is None:
        args.steps = 1000 if args.workload == 'c2' else 3      # ~1 s of timed work by default
    if args.warmup is None:
        args.warmup = 20 if args.workload == 'c2' else 1
    return args


###############################################################################
# CPU baseline (rank 0, one GPU only)
###############################################################################


def cpu_baseline(state, seconds):
    """The oracle (CPU port of the reference path) on the REAL 32 x 1000
    batch: thread sweep up to the physical cores on one batch each, then the
    best setting timed on whole batches until the budget is used; fp32
    (autocast off: the parity definition) and the bf16-autocast variant (what
    the reference ships with on CPU, ppgs/core.py:586)."""
    from oracle import ppg_oracle
    generator = torch.Generator().manual_seed(1234)
    audio = 0.1 * torch.randn(BATCH, 1, SAMPLES, generator=generator)
    default_threads = torch.get_num_threads()
    logical = os.cpu_count() or default_threads
    physical = max(logical // 2, 1)
    ppg_oracle.from_audio(state, audio[:1, :, :16000])           # warm-up
    sweep = {}
    budget_start = time.perf_counter()
    for threads in sorted({t for t in (8, 16, 32, 64, 128, physical) if t <= logical}):
        if time.perf_counter() - budget_start > 0.5 * seconds:
            break
        torch.set_num_threads(threads)
        start = time.perf_counter()
        ppg_oracle.from_audio(state, audio)
        sweep[threads] = time.perf_counter() - start
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)

    def timed(fn, budget):
        start = time.perf_counter()
        batches = 0
        while True:
            fn()
            batches += 1
            elapsed = time.perf_counter() - start
            if elapsed > budget or batches >= 8:
                return batches, elapsed
    batches, elapsed = timed(lambda: ppg_oracle.from_audio(state, audio), 0.3 * seconds)

    def autocast():
        with torch.autocast('cpu', dtype=torch.bfloat16):
            ppg_oracle.from_audio(state, audio)
    autocast()
    ab, ae = timed(autocast, 0.15 * seconds)
    torch.set_num_threads(default_threads)
    return {
        'value': batches * BATCH * FRAMES / elapsed,
        'unit': 'frames/s',
        'cores': best,
        'kind': 'port',
        'sample': f'{batches} batches of {BATCH} x {FRAMES} frames (the benchmark batch; mel frontend + '
                  f'encoder + softmax, fp32 CPU oracle), {elapsed:.1f} s with {best} torch threads '
                  f'(best of the sweep {{threads: s/batch}} = '
                  f'{ {k: round(v, 2) for k, v in sweep.items()} }) on {logical} logical cores',
        'bf16_autocast_value': ab * BATCH * FRAMES / ae,
        'bf16_autocast_sample': f'{ab} batches, {ae:.1f} s, same threads, torch.autocast(cpu, bfloat16)',
    }


def pmc_summary(suffix='', precision='bf16'):
    """The committed rocprofv3 PMC summary of THIS build: profiles/r*_pmc_summary<suffix>.txt whose `# lib_sha256_16=` header
    equals the sha256 of the library this process loaded (tests/pmc_summary.py writes it; a clean `make` reproduces
    the library byte for byte).  PMC counters cannot be read inside this process, so the figures are never of the
    run that prints them -- and a summary of another build is not quoted at all.
    -> ({kernel: {counter: mean per dispatch, 'dispatches': n}}, file) or (None, reason)."""
    import glob
    import re
    from ppgs_amd import engine as E
    sha = E.library_sha16()
    seen = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_pmc_summary{suffix}.txt')), reverse=True):
        with open(path) as f:
            text = f.read()
        m = re.search(r'^# lib_sha256_16=([0-9a-f]+)', text, re.M)
        seen.append(f'{os.path.basename(path)}:{m.group(1) if m else "unlabelled"}')
        if not m or m.group(1) != sha:
            continue
        # the counters are of ONE arithmetic mode (`# precision=`; summaries older than the header: bf16, or fp16x2 by
        # their file name): a step of another mode runs other kernels for other durations
        mode = re.search(r'^# precision=(\w+)', text, re.M)
        mode = mode.group(1) if mode else ('fp16x2' if 'fp16x2' in suffix else 'bf16')
        if mode != precision:
            seen[-1] += f'(precision {mode})'
            continue
        kernels = {}
        for line in text.splitlines():
            if line.startswith(('#', '==')) or ': ' not in line:
                continue
            name, rest = line.split(': ', 1)
            entry = kernels.setdefault(name, {})
            for key, value in re.findall(r'(\w+)=([0-9.e+-]+)', rest):
                entry[key] = max(entry.get(key, 0.0), float(value)) if key == 'dispatches' else float(value)
        return kernels, os.path.relpath(path, ROOT)
    return None, f'no committed PMC summary of this build (library {sha}; committed: {", ".join(seen) or "none"})'


def pmc_traffic(kernels, prefix):
    """HBM-side bytes per launch of the dominant kernel: 2 x FETCH_SIZE (gfx950 reports half of a wide coalesced
    read, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB; the variant with the Q/K/V tail (4 of the 5 launches of a
    step) is the one the roofline line describes."""
    import re
    names = [k for k in kernels if k.startswith(prefix)]
    tail = [k for k in names if re.search(r', true\b|Lb1', k)]
    for name in tail or names:
        entry = kernels[name]
        if 'FETCH_SIZE' in entry and 'WRITE_SIZE' in entry:
            return (2.0 * entry['FETCH_SIZE'] + entry['WRITE_SIZE']) * 1024.0
    return None


def pmc_mfma_busy(kernels):
    """Matrix-pipe busy cycles of ONE step, all kernels: sum of SQ_VALU_MFMA_BUSY_CYCLES per dispatch x dispatches per
    step (dispatches relative to the frontend's, which runs once per step)."""
    front = [k for k in kernels if 'FrontendTables' in k or k.startswith('frontend_kernel')]
    if not front or not kernels[front[0]].get('dispatches'):
        return None
    steps = kernels[front[0]]['dispatches']
    return sum(e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) * e.get('dispatches', 0.0) / steps for e in kernels.values())


###############################################################################
# Workloads
###############################################################################


def layer_flops_per_frame(kernels, breakdown_steps, launches_per_layer):
    """Algorithmic FLOPs per processed frame of one layer-kernel launch, the MEAN over a step's LAYERS launches (the
    roofline leg's events rotate through them): FFN 4*H*F; + the out-projection 2*H*H when it is fused
    into the kernel (no separate out-proj launch shows up); + the NEXT layer's Q/K/V projection 6*H*H for the launches
    that compute it -- never the last layer's, and one launch fewer when the head kernel makes layer 0's Q/K/V (no
    gather launch shows up) and no stand-alone Q/K/V launch exists: 4 of the 5."""
    op_fused = kernels['outproj_ln'][1] == 0
    qkv_own = kernels['qkv'][1] / breakdown_steps / launches_per_layer      # stand-alone Q/K/V launches per step and pipeline
    head_fused = 1 if kernels['gather'][1] == 0 else 0                       # layer 0's Q/K/V inside the head kernel
    qkv_fused_layers = max(min(LAYERS - qkv_own - head_fused, LAYERS - 1), 0.0) if op_fused else 0.0
    return (4.0 * HIDDEN * FFN + (2.0 * HIDDEN * HIDDEN if op_fused else 0.0)
            + 6.0 * HIDDEN * HIDDEN * qkv_fused_layers / LAYERS), op_fused, qkv_fused_layers


def run_c2(args, rank, world, local_rank, use_dist):
    import ppgs_amd
    from ppgs_amd import data, engine as E

    state = ppgs_amd.weights.seeded_state_dict(seed=1234)
    model = E.Engine(state, local_rank, args.precision)
    generator = torch.Generator().manual_seed(1234 + rank)
    host_audio = (0.1 * torch.randn(BATCH, 1, SAMPLES, generator=generator)).pin_memory()
    audio = host_audio.cuda(non_blocking=True)
    lengths = [FRAMES] * BATCH

    def step(engine=model):
        mel = ppgs_amd.preprocess.mel.from_audios(audio)
        return engine.encode(mel, lengths)

    nccl = use_dist and dist.get_backend() == 'nccl'

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank]) if nccl else dist.barrier()
            torch.cuda.synchronize()

    def timed_block(engine=model):
        barrier()
        start = time.perf_counter()
        for _ in range(args.steps):
            out = step(engine)
        # this rank's own time to ITS synchronize (before the closing barrier): which rank bends a scaling curve
        torch.cuda.synchronize()
        own = time.perf_counter() - start
        barrier()
        elapsed = time.perf_counter() - start
        if use_dist:
            worst = torch.tensor([elapsed], device='cuda' if nccl else 'cpu', dtype=torch.float64)
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            elapsed = float(worst.item())
            mine = torch.tensor([own], device='cuda' if nccl else 'cpu', dtype=torch.float64)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank[:] = [float(t.item()) for t in every]
        else:
            per_rank[:] = [own]
        return elapsed, out

    per_rank = []

    def prewarm(engine=model):
        """Disclosed, untimed: the same step for >= --prewarm-s seconds of wall time, so that the K timed
        steps see the steady-state clocks of a loaded GPU (the metric is the steady-state
        loop) instead of the ramp of a chip that was idle a moment ago."""
        steps, start = 0, time.perf_counter()
        while args.prewarm_s > 0 and time.perf_counter() - start < args.prewarm_s:
            for _ in range(20):
                step(engine)
            torch.cuda.synchronize()
            steps += 20
        return steps, time.perf_counter() - start

    out = step()                         # setup: window plan built and uploaded, workspace allocated
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    # (the first torch kernel of the process loads torch's code objects: tens of milliseconds of idle GPU --
    # here, not between the warm-up and the timed region)
    assert out.shape == (BATCH, 40, FRAMES) and bool(torch.isfinite(out).all())
    # the events of the roofline leg exist before the timed region starts
    model.profile(True, classes=['ffn'], stride=EVENT_STRIDE)
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    model.profile_read()
    barrier()                            # (N > 1: RCCL builds its communicator on the first collective, not in the timed region)
    # Nothing may idle the GPU between here and the timed region: after ~1 ms without work the chip drops its
    # clocks and needs ~16 ms of load to get them back (tools/clock_series.py: first 20-step block after an idle
    # gap 0.81 - 0.83 ms/step, every later one 0.74) -- a 20-step timed region IS 16 ms.
    prewarm_steps, prewarm_seconds = prewarm()
    model.profile(True, classes=['ffn'], stride=EVENT_STRIDE)         # (reset of the counters: two ctypes calls)

    # timed region: HIP events only around the dominant kernel (roofline leg)
    # (every 6th launch: 6 is coprime with the 5 layers, so the samples rotate
    # through the five launches of a step; two event records per launch cost
    # ~1.5 us each on the stream, 1.5 % of the step if every launch is timed)
    elapsed, out = timed_block()
    rank_ms = [1e3 * t / args.steps for t in per_rank]
    ffn_ms, ffn_samples = model.profile_read()['ffn']
    model.profile(False)
    # two more blocks of the same K steps: how much the figure moves from block to block
    repeats = [1e3 * timed_block()[0] / args.steps for _ in range(2)]
    # the shader clock while the same block of steps runs once more (untimed; a sampler thread reads sysfs)
    with ClockSampler(local_rank) as sampler:
        timed_block()
    clock = sampler.summary()
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

    # host <-> device legs, timed on their own (never part of `value`): pinned audio H2D, posteriors D2H
    host_out = torch.empty(out.shape, dtype=out.dtype).pin_memory()
    copies = {}
    for name, fn in (('h2d_ms', lambda: audio.copy_(host_audio, non_blocking=True)),
                     ('d2h_ms', lambda: host_out.copy_(out, non_blocking=True))):
        fn()
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        copies[name] = 1e3 * (time.perf_counter() - start) / 10

    alt = alt_x2 = None
    if rank == 0 and world == 1 and not args.no_alt and args.precision == 'bf16':
        # the same step with fp16 MFMA operands (same MFMA rate, 8x smaller operand rounding)
        other = E.Engine(state, local_rank, 'fp16')
        for _ in range(max(args.warmup, 3)):
            step(other)
        prewarm(other)
        alt_elapsed, alt_out = timed_block(other)
        alt = {'dtype': 'fp16 operands, fp32 accumulate', 'ms_per_step': 1e3 * alt_elapsed / args.steps,
               'value': BATCH * FRAMES * args.steps / alt_elapsed,
               'max_abs_vs_bf16': float((alt_out - out).abs().max())}
        del other
        # ... and with split operands (fp16 hi + lo planes, three MFMAs per product): the mode that meets the north
        # star's 1e-4 (measured 3e-6 against the fp32 oracle) without the f32-input MFMAs' 1/16 rate
        other = E.Engine(state, local_rank, 'fp16x2')
        for _ in range(max(args.warmup, 3)):
            step(other)
        prewarm(other)
        x2_elapsed, x2_out = timed_block(other)
        alt_x2 = {'dtype': 'fp16x2: fp32 values as fp16 hi + lo, 3 fp16 MFMAs per product, fp32 accumulate',
                  'ms_per_step': 1e3 * x2_elapsed / args.steps, 'value': BATCH * FRAMES * args.steps / x2_elapsed,
                  'end_to_end_tflops_algorithmic': BATCH * data.flops(FRAMES) * args.steps / x2_elapsed / 1e12,
                  'max_abs_vs_fp16': float((x2_out - alt_out).abs().max())}
        # matrix-pipe busy cycles of an fp16x2 step (three MFMAs per product) from the committed PMC pass of this build
        # with --precision fp16x2, over 4 pipes x CUs x this leg's step time at the nominal 2.4 GHz
        x2_pmc, x2_from = pmc_summary('_fp16x2', 'fp16x2')
        x2_busy = pmc_mfma_busy(x2_pmc) if x2_pmc else None
        alt_x2['mfma_busy_cycles_per_step'] = x2_busy
        alt_x2['mfma_busy_frac'] = (x2_busy / (4.0 * torch.cuda.get_device_properties(local_rank).multi_processor_count
                                               * alt_x2['ms_per_step'] * 2.4e6)) if x2_busy else None
        alt_x2['mfma_busy_from'] = x2_from
        del other
    alt_streams = None
    _, info = E.plan_windows(BATCH, FRAMES, lengths)
    pipelines = model.pipelines(info.tokens)
    if rank == 0 and world == 1 and not args.no_alt and 'PPGS_AMD_STREAMS' not in os.environ and pipelines > 1:
        # the same step as ONE pipeline (PPGS_AMD_STREAMS=1): every launch then has the whole chip to itself, which is
        # the configuration a per-kernel whole-chip roofline fraction can be read from; it is the slower one (the
        # half-batches' memory-bound phases no longer run beside each other's MFMA phases), so it is not the default.
        os.environ['PPGS_AMD_STREAMS'] = '1'
        try:
            other = E.Engine(state, local_rank, args.precision)
        finally:
            del os.environ['PPGS_AMD_STREAMS']
        for _ in range(max(args.warmup, 3)):
            step(other)
        prewarm(other)
        other.profile(True, classes=['ffn'], stride=EVENT_STRIDE)
        s1_elapsed, s1_out = timed_block(other)
        s1_ms, s1_samples = other.profile_read()['ffn']
        other.profile(False)
        alt_streams = {'pipelines': 1, 'ms_per_step': 1e3 * s1_elapsed / args.steps,
                       'value': BATCH * FRAMES * args.steps / s1_elapsed,
                       'mean_launch_ms': s1_ms / max(s1_samples, 1), 'timed_launches': s1_samples,
                       'max_abs_vs_default': float((s1_out - out).abs().max())}
        del other

    alt_graph = None
    if rank == 0 and world == 1 and not args.no_alt:
        # the same step with the encoder's launch sequence (both pipelines, their fork / join) as ONE hipGraph replay
        # (Engine.graphed; the mel frontend launched in front of it as before).  Reported beside the line, never `value`:
        # the timed region above is the eager path, whose launches the roofline leg's HIP events can bracket.
        try:
            replay = model.graphed(BATCH, FRAMES, lengths)

            def graph_step():
                return replay(ppgs_amd.preprocess.mel.from_audios(audio))
            for _ in range(max(args.warmup, 3)):
                graph_step()
            start = time.perf_counter()
            while args.prewarm_s > 0 and time.perf_counter() - start < args.prewarm_s:
                for _ in range(20):
                    graph_step()
                torch.cuda.synchronize()
            torch.cuda.synchronize()
            start = time.perf_counter()
            for _ in range(args.steps):
                g_out = graph_step()
            torch.cuda.synchronize()
            g_elapsed = time.perf_counter() - start
            alt_graph = {'what': 'encoder as one hipGraph replay per step (Engine.graphed), frontend in front of it',
                         'ms_per_step': 1e3 * g_elapsed / args.steps, 'value': BATCH * FRAMES * args.steps / g_elapsed,
                         'max_abs_vs_default': float((g_out - out).abs().max())}
            del replay
        except Exception as error:                              # the leg is optional: the line must not depend on it
            alt_graph = {'skipped': f'{type(error).__name__}: {error}'}

    if rank != 0:
        return None
    ms_per_step = 1e3 * elapsed / args.steps
    frames_per_s = world * BATCH * FRAMES * args.steps / elapsed
    launches_per_layer = max(kernels['ffn'][1] // (LAYERS * breakdown_steps), 1)
    flops_per_frame, op_fused, qkv_fused_layers = layer_flops_per_frame(kernels, breakdown_steps, launches_per_layer)
    ffn_flops = flops_per_frame * info.processed_frames / launches_per_layer
    ffn_tflops = ffn_flops / (1e-3 * ffn_ms / max(ffn_samples, 1)) / 1e12
    peak = PEAK_FP32_TFLOPS if args.precision == 'fp32' else PEAK_16BIT_TFLOPS
    step_flops = BATCH * data.flops(FRAMES)
    # CUs a launch of the dominant kernel can occupy: one workgroup per CU (its LDS tile), one workgroup per token tile
    launch_workgroups = None
    x2 = args.precision == 'fp16x2'                     # ffn32x2_kernel: 96-token tiles
    layer32 = args.precision in ('bf16', 'fp16') and os.environ.get('PPGS_AMD_LAYER32', '1') != '0'
    if layer32 or x2:
        launch_workgroups = -(-info.tokens // (96 if x2 else 160)) // launches_per_layer
    cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
    launch_cus = min(launch_workgroups, cus) if launch_workgroups else cus
    launch_peak = peak * launch_cus / cus
    pmc, pmc_from = pmc_summary('_fp16x2' if x2 else '', args.precision)
    traffic = pmc_traffic(pmc, 'ffn32x2_' if x2 else ('layer32_' if layer32 else 'ffn_')) if pmc else None
    mfma_busy = pmc_mfma_busy(pmc) if pmc else None
    kernel_name = ('ffn32x2_kernel (feature-split layer kernel on fp16 hi + lo operand pairs, three v_mfma_f32_32x32x16_f16 '
                   'per product: out-proj+residual+LN1, W1+ReLU+W2+residual+LN2, next layer Q/K/V)') if x2 else (
                   'layer32_kernel (feature-split layer kernel on v_mfma_f32_32x32x16: out-proj+residual+LN1, '
                   'W1+ReLU+W2+residual+LN2' + (', next layer Q/K/V)' if qkv_fused_layers else ')')) if layer32 else (
        'ffn_mixed_kernel (token-split layer kernel: out-proj+residual+LN1, W1+ReLU+W2+residual+LN2'
        + (', next layer Q/K/V)' if qkv_fused_layers else ')') if op_fused
        else 'ffn_kernel (fused W1+ReLU+W2+residual+LayerNorm)')
    line = {
        'metric': 'PPG frames/sec (whole node), mel repr, batch=32x1000 frames',
        'value': frames_per_s,
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'prewarm_s': prewarm_seconds,
        'prewarm_steps': prewarm_steps,
        'ms_per_step': ms_per_step,
        # every rank's own time to finish its K steps (before the closing barrier), in rank order
        'per_rank_ms_per_step': {'min': min(rank_ms), 'max': max(rank_ms), 'all': rank_ms},
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
            'kernel': kernel_name,
            'bound': 'mfma',
            'achieved': ffn_tflops,
            'peak': peak,
            'unit': 'TFLOP/s',
            'frac': ffn_tflops / peak,
            'peak_scope': f'whole chip ({cus} CUs, dense {args.precision} MFMA peak of MI355X_MICROARCH.md).  A launch is '
                          f'{launch_workgroups or "?"} workgroups, one per CU, and {pipelines} pipeline(s) of the batch run '
                          'beside each other: `achieved` is ONE launch\'s FLOPs over its own duration while the other '
                          'pipeline\'s kernels share the chip -- the step as a whole is end_to_end_mfma_frac, and '
                          'alt_single_pipeline has the per-launch figure with the chip to itself',
            'frac_of_occupied_cus': ffn_tflops / launch_peak,
            'occupied_cus': launch_cus,
            'pipelines': pipelines,
            'traffic': traffic,
            'traffic_from': f'{pmc_from} (committed rocprofv3 --pmc passes of this command and this build, not this run)'
                            if pmc else pmc_from,
            'flops_per_launch': ffn_flops,
            'mean_launch_ms': ffn_ms / max(ffn_samples, 1),
            'timed_launches': ffn_samples,
        },
        'end_to_end_tflops': step_flops * args.steps / elapsed / 1e12,
        'end_to_end_mfma_frac': step_flops * args.steps / elapsed / 1e12 / peak,
        # HIP-event durations summed over BOTH pipelines' streams: with two pipelines the classes overlap in time and
        # the sum exceeds the step (divide a class by `pipelines` for its share of the wall clock, roughly)
        'kernel_ms_per_step_stream_summed': {k: v[0] / breakdown_steps for k, v in kernels.items()},
        'kernel_ms_pipelines': pipelines,
        # matrix-pipe busy cycles of a step (all kernels, rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES of the committed PMC pass
        # of this build) over 4 pipes x CUs x the step's duration at the nominal 2.4 GHz the 2.5 PF peak is quoted at
        'mfma_busy_cycles_per_step': mfma_busy,
        'mfma_busy_frac': mfma_busy / (4.0 * cus * ms_per_step * 2.4e6) if mfma_busy else None,
        # the shader clock sampled while one more block of the same K steps ran, and the same busy cycles over the
        # cycles that clock DELIVERED in a step (the pipes are busier than the nominal-clock fraction says when the chip
        # runs below 2.4 GHz)
        'shader_clock_under_load': clock,
        'mfma_busy_frac_of_delivered_cycles': (mfma_busy / (4.0 * cus * ms_per_step * 1e3 * clock['mean_mhz'])
                                               if mfma_busy and clock else None),
        'env_switches': args.env_config,
        'repeat_blocks_ms_per_step': repeats,
        'h2d_ms': copies['h2d_ms'],
        'd2h_ms': copies['d2h_ms'],
        'pcie_inclusive_ms_per_step': ms_per_step + copies['h2d_ms'] + copies['d2h_ms'],
    }
    if alt:
        line['alt_precision'] = alt
    if alt_x2:
        line['alt_precision_fp16x2'] = alt_x2
    if alt_streams:
        # the one-pipeline run's layer launches are whole-batch launches on the whole chip
        alt_streams['roofline_frac_whole_chip'] = (ffn_flops * launches_per_layer / (1e-3 * alt_streams['mean_launch_ms'])
                                                   / 1e12 / peak)
        line['alt_single_pipeline'] = alt_streams
    if alt_graph:
        line['alt_graph_replay'] = alt_graph
    if rank == 0 and world == 1:
        probe = dense_mfma_probe()
        line['dense_mfma_probe'] = probe
        if probe and args.precision != 'fp32':
            # the same fractions against what the probe measured on THIS box instead of the nominal 2.5 PFLOP/s
            delivered = probe['dense']['tflops']
            line['roofline']['frac_of_delivered_dense_peak'] = ffn_tflops / delivered
            line['end_to_end_frac_of_delivered_dense_peak'] = line['end_to_end_tflops'] / delivered
    if world == 1 and not args.no_cpu:
        line['cpu_baseline'] = cpu_baseline(state, args.cpu_seconds)
        line['speedup_vs_cpu'] = frames_per_s / line['cpu_baseline']['value']
    if args.env_ablation:
        line['ablation_env'] = args.env_ablation
        line['value_under_ablation'] = line['value']
        line['value'] = None
    return line


def run_c4(args, rank, world, local_rank, use_dist):
    """configs[3]: a corpus of ragged utterances, LPT-sharded, every rank
    holding only its shard; posteriors gathered to rank 0 at the end."""
    import ppgs_amd
    from ppgs_amd import config, data, distributed, engine as E

    state = ppgs_amd.weights.seeded_state_dict(seed=1234)
    model = E.Engine(state, local_rank, args.precision)
    generator = torch.Generator().manual_seed(1234)
    frames = torch.randint(50, 3001, (args.utterances,), generator=generator).tolist()
    shards = distributed.shard_lpt([data.flops(f) for f in frames], world)
    mine = shards[rank]
    # (row budget: no batch is one tile more than a whole round of the layer kernel's workgroups)
    batches = data.pack_batches([frames[i] for i in mine], 32000, max_rows=data.row_budget(32000, gpu=local_rank))
    # this rank's padded batches, generated on its own GPU (nobody else ever holds them)
    device_generator = torch.Generator(device=DEVICE).manual_seed(1234 + rank)
    padded = []
    for batch in batches:
        longest = max(frames[mine[j]] for j in batch) * config.HOPSIZE
        block = 0.1 * torch.randn((len(batch), 1, longest), device=DEVICE, generator=device_generator)
        for row, j in enumerate(batch):
            block[row, :, frames[mine[j]] * config.HOPSIZE:] = 0.
        padded.append(block)

    def one_pass(keep=False):
        outs = []
        for batch, block in zip(batches, padded):
            lens = [frames[mine[j]] for j in batch]
            mel = ppgs_amd.preprocess.mel.from_audios(block)
            out = model.encode(mel, lens)
            if keep:
                outs.append(torch.cat([out[row, :, :n].T for row, n in enumerate(lens)], dim=0))
        return outs

    nccl = use_dist and dist.get_backend() == 'nccl'

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank]) if nccl else dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    model.profile(True, classes=['ffn'], stride=1)
    barrier()
    start = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier()
    elapsed = time.perf_counter() - start
    ffn_ms, ffn_launches = model.profile_read()['ffn']
    model.profile(False)
    if use_dist:
        worst = torch.tensor([elapsed], device='cuda' if nccl else 'cpu', dtype=torch.float64)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        elapsed = float(worst.item())
    # the only collective of the path: gatherv of the posteriors to rank 0
    outs = one_pass(keep=True)
    local = torch.cat(outs, dim=0) if outs else torch.zeros((0, 40), device=DEVICE)
    order = [mine[j] for batch in batches for j in batch]
    barrier()
    start = time.perf_counter()
    gathered = distributed.gather_ragged(local, [frames[i] for i in order])
    barrier()
    gather_ms = 1e3 * (time.perf_counter() - start)
    if rank != 0:
        return None
    assert sum(len(g) for g in gathered) == args.utterances
    total_frames = sum(frames)
    # FLOPs of my own shard's launches (rank 0's) for the roofline of its layer kernel
    processed = 0
    chip_share = 0.0                 # sum over batches of frames / pipelines: a launch of a p-pipeline batch holds 1/p of the CUs
    for batch in batches:
        lens = [frames[mine[j]] for j in batch]
        _, info = E.plan_windows(len(batch), max(lens), lens)
        processed += info.processed_frames
        chip_share += info.processed_frames / model.pipelines(info.tokens)
    per_frame = 4.0 * HIDDEN * FFN + 2.0 * HIDDEN * HIDDEN + 6.0 * HIDDEN * HIDDEN * (LAYERS - 1) / LAYERS
    # plain whole-chip accounting: the layer FLOPs of the pass over the SUM of the layer launches' durations (HIP events,
    # both pipelines' streams), every launch priced against the whole chip -- a launch of a two-pipeline batch shares
    # the chip with the other pipeline's kernels, so this is a lower bound; the chip-share weighting (a launch of a
    # p-pipeline batch holds 1/p of the CUs) is kept beside it as frac_of_occupied_cus
    ffn_tflops = per_frame * processed * LAYERS * args.steps / (1e-3 * ffn_ms) / 1e12 if ffn_ms else 0.0
    chip_ms = ffn_ms * chip_share / max(processed, 1)
    occupied_tflops = per_frame * processed * LAYERS * args.steps / (1e-3 * chip_ms) / 1e12 if ffn_ms else 0.0
    peak = PEAK_FP32_TFLOPS if args.precision == 'fp32' else PEAK_16BIT_TFLOPS
    padded_frames = sum(len(b) * max(frames[mine[j]] for j in b) for b in batches)
    return {
        'metric': 'PPG frames/sec (whole node), mel repr, 10k ragged utterances packed under max_frames',
        'value': total_frames * args.steps / elapsed,
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': args.precision,
        'data': 'synthetic 16 kHz audio (0.1*randn per rank, frame counts randint(50,3001) seed 1234), '
                'seeded random weights of the reference architecture',
        'config': {
            'workload': f'configs[3]: {args.utterances} utterances of 50..3000 frames ({total_frames} frames), packed under '
                        f'max_frames=32000, LPT-sharded by chunk-aware FLOPs over {world} rank(s), each rank holding only '
                        'its shard in HBM; one step = one pass over the corpus',
            'utterances': args.utterances, 'rank0_batches': len(batches),
            'rank0_padding_efficiency': sum(frames[i] for i in mine) / max(padded_frames, 1),
            'parallelism': f'utterance-sharded x{world}, gatherv to rank 0 only',
        },
        'gather_ms': gather_ms,
        'gather_bytes': total_frames * 40 * 4,
        'roofline': {
            'kernel': 'layer kernel launches of rank 0 (layer32_kernel / token-split kernels by batch size)',
            'bound': 'mfma', 'achieved': ffn_tflops, 'peak': peak, 'unit': 'TFLOP/s',
            'frac': ffn_tflops / peak, 'traffic': None,
            'frac_of_occupied_cus': occupied_tflops / peak,
            'mean_launch_ms': ffn_ms / max(ffn_launches, 1), 'timed_launches': ffn_launches,
            'chip_share_of_a_launch': chip_share / max(processed, 1),
            'end_to_end_mfma_frac': sum(data.flops(f) for f in frames) * args.steps / elapsed / 1e12 / peak / world,
        },
    }


###############################################################################
# Rank set-up
###############################################################################


def rank_main(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU path')
    if args.gpus != world:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)')
    # PPGS_BENCH_ALIAS_GPUS=1: a DRY RUN of the N-rank path on fewer GPUs than ranks -- rank r uses GPU
    # r mod (GPUs present).  RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the process
    # group is gloo and the gatherv is staged through host memory: what such a run checks is the N-process
    # host side (sharding, per-rank shards only, CPU binding, N launch loops sharing a GPU, the gatherv's
    # bookkeeping); its frames/s say nothing about N GPUs and the record is labelled so.
    alias = bool(os.environ.get('PPGS_BENCH_ALIAS_GPUS')) and torch.cuda.device_count() < world
    if alias:
        local_rank = local_rank % torch.cuda.device_count()
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'bench.py: rank {rank} wants GPU {local_rank}, the node shows {torch.cuda.device_count()}')
    torch.cuda.set_device(local_rank)
    # (PPGS_BENCH_FORCE_DIST=1: exercise the RCCL barrier / max-over-ranks / gather path with one rank)
    use_dist = world > 1 or bool(os.environ.get('PPGS_BENCH_FORCE_DIST'))
    cpus = []
    if world > 1:
        from ppgs_amd import distributed
        cpus = distributed.bind_cpus(int(os.environ.get('LOCAL_RANK', '0')), world, device=local_rank if alias else None)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if alias or os.environ.get('PPGS_BENCH_BACKEND') == 'gloo':
            dist.init_process_group('gloo', rank=rank, world_size=world)
        elif os.environ.get('PPGS_BENCH_BACKEND') == 'nccl-eager':
            # (experiment only.  With device_id= PyTorch builds the communicator at init and binds the group to the
            # device; a process in that state ran the allocation-heavy C4 pass -- 496 batch shapes, new tensors per
            # batch -- 60 % slower on the HOST side: 624 ms against 387 ms for the same kernels, one rank,
            # profiles/r4_c4_process_group.txt.  The lazy group below, whose communicator appears with the first
            # barrier, costs nothing.)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world)
    line = None
    try:
        line = (run_c2 if args.workload == 'c2' else run_c4)(args, rank, world, local_rank, use_dist)
        if rank == 0 and use_dist:
            line['backend'] = ('gloo: PPGS_BENCH_ALIAS_GPUS dry run, ranks share GPUs -- NOT a multi-GPU measurement' if alias
                               else ('nccl (RCCL)' if dist.get_backend() == 'nccl' else dist.get_backend()))
            line['rccl_world_size' if dist.get_backend() == 'nccl' else 'gloo_world_size'] = dist.get_world_size()
            line['rank0_cpus'] = len(cpus)
    finally:
        if use_dist:
            dist.destroy_process_group()
    # (after the process group is gone: RCCL prints its version banner on stdout when it shuts down,
    # and the JSON record has to be the LAST line)
    if rank == 0 and line is not None:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


def _spawned(local_rank, args, port):
    os.environ.update(
        RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(args.gpus),
        MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    rank_main(args)


def main():
    args = parse()
    args.env_config, args.env_ablation = env_guard(args)
    if 'RANK' in os.environ or args.gpus == 1:
        rank_main(args)                  # launched by torch.distributed.run (or a single rank)
        return
    # bare `python bench.py --gpus N`: spawn the N ranks here
    found = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if found < args.gpus and not (found and os.environ.get('PPGS_BENCH_ALIAS_GPUS')):
        raise SystemExit(f'bench.py: --gpus {args.gpus} but this node shows {found} GPU(s) '
                         '(PPGS_BENCH_ALIAS_GPUS=1: dry run of the N-rank host path on the GPUs present, over gloo)')
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    mp.spawn(_spawned, args=(args, port), nprocs=args.gpus, join=True)


if __name__ == '__main__':
    main()
