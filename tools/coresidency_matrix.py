"""Co-residency forensics: which ingredient of an attention kernel makes round 2's 256-thread mel
frontend (-DPPG_FE_R2 build: PPGS_AMD_LIB=tools/bin/libppgs_amd_r2.so) compute a frame pair wrong?

    PPGS_AMD_LIB=tools/bin/libppgs_amd_r2.so python tools/coresidency_matrix.py [aggressor ...]

aggressors: sdpa, encode (the known ones), or a bit mask of tools/probes/aggressors.hip's instruction classes
(`m<mask>[:lds_bytes[:grid]]`, e.g. m1 = v_exp_f32 only, m3 = exp + MFMA, m7:65536 ...), or `victims` = the
synthetic victim kernels (VALU-only / LDS-exchange-only self checks) beside sdpa.
Prints one line per aggressor: wrong mel tensors of 240 frontend launches.
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

agg = ctypes.CDLL(os.path.join(ROOT, 'tools', 'bin', 'libaggressors.so'))
agg.aggressor_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
agg.victim_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2

BATCH, FRAMES = 32, 1000
REPS = int(os.environ.get('REPS', '40'))
gen = torch.Generator().manual_seed(1234)
audio = (0.1 * torch.randn(BATCH, 1, FRAMES * 160, generator=gen)).cuda()
lengths = [FRAMES] * BATCH
mel_ref = ppgs_amd.preprocess.mel.from_audios(audio)
torch.cuda.synchronize()
a, b = torch.cuda.Stream(), torch.cuda.Stream()
src = torch.randn(1 << 20, device='cuda')
sink = torch.zeros(256, device='cuda')
q_ = torch.randn(32, 2, 1000, 128, device='cuda', dtype=torch.bfloat16)
model = None


def synthetic(mask, lds, grid, iters):
    rc = agg.aggressor_launch(mask, grid, iters, lds, src.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def calibrate(mask, lds, grid, target_us=400.0):
    """iterations for a launch of about target_us on an otherwise idle chip"""
    iters = 200
    for _ in range(3):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        synthetic(mask, lds, grid, iters)
        start.record()
        synthetic(mask, lds, grid, iters)
        stop.record()
        torch.cuda.synchronize()
        us = start.elapsed_time(stop) * 1e3
        iters = max(1, int(iters * target_us / max(us, 1.0)))
    return iters, us


def frontends_beside(run_aggressor):
    total = bad = 0
    for rep in range(REPS):
        with torch.cuda.stream(a):
            run_aggressor()
        with torch.cuda.stream(b):
            mels = [ppgs_amd.preprocess.mel.from_audios(audio) for _ in range(6)]
        torch.cuda.synchronize()
        for m in mels:
            total += 1
            bad += not torch.equal(m, mel_ref)
    return bad, total


def sdpa():
    for _ in range(6):
        torch.nn.functional.scaled_dot_product_attention(q_, q_, q_)


for name in sys.argv[1:] or ['sdpa']:
    if name == 'sdpa':
        bad, total = frontends_beside(sdpa)
    elif name == 'none':
        bad, total = frontends_beside(lambda: None)
    elif name == 'encode':
        model = model or E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, 'bf16')
        bad, total = frontends_beside(lambda: [model.encode(mel_ref, lengths) for _ in range(2)])
    elif name == 'victims':
        counters = torch.zeros(8, dtype=torch.int64, device='cuda')
        for mode in (1, 2, 4, 8, 15):
            for lds in (81408, 1024):
                for beside in ('quiet', 'sdpa'):
                    counters.zero_()
                    for rep in range(REPS):
                        if beside == 'sdpa':
                            with torch.cuda.stream(a):
                                sdpa()
                        with torch.cuda.stream(b):
                            for _ in range(4):
                                rc = agg.victim_launch(mode, 512, 40, lds, counters.data_ptr(), torch.cuda.current_stream().cuda_stream)
                                assert rc == 0
                        torch.cuda.synchronize()
                    c = counters.tolist()
                    print(f'victim mode {mode:2d} lds {lds:6d} {beside:5s}: mismatches pk {c[0]} sqrt {c[1]} dpp {c[2]} lds {c[3]} (workgroups {c[4]})', flush=True)
        continue
    else:
        parts = name[1:].split(':')
        mask = int(parts[0])
        lds = int(parts[1]) if len(parts) > 1 else 65536
        grid = int(parts[2]) if len(parts) > 2 else 512
        iters, us = calibrate(mask, lds, grid)
        bad, total = frontends_beside(lambda: [synthetic(mask, lds, grid, iters) for _ in range(2)])
        name = f'{name} (mask {mask}, lds {lds}, grid {grid}, iters {iters}, ~{us:.0f} us alone)'
    print(f'beside {name}: {bad} of {total} mel tensors differ', flush=True)
