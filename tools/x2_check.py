"""The split-precision mode (fp16x2) against the reference fixtures and the oracle (GPU)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from oracle import ppg_oracle as O                # noqa: E402
from ppgs_amd import engine as E, weights as W    # noqa: E402


def golden(name):
    with np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz')) as data:
        return {k: data[k] for k in data.files}


state = W.seeded_state_dict(seed=1234)
g2, g3 = golden('g2_single_window'), golden('g3_chunked')
engine = E.Engine(state, 0, 'fp16x2')
errs = {}
ppg = engine.encode(torch.from_numpy(g2['features']).cuda(), g2['lengths'].tolist()).cpu().numpy()
errs['g2'] = np.abs(ppg - g2['ppg']).max()
print('g2', errs['g2'], 'finite', np.isfinite(ppg).all(), flush=True)
ppg = engine.encode(torch.from_numpy(g3['features_a']).cuda(), g3['lengths_a'].tolist()).cpu().numpy()
errs['g3a'] = np.abs(ppg - g3['ppg_a']).max()
ppg = engine.encode(torch.from_numpy(g3['features_c']).cuda(), g3['lengths_c'].tolist()).cpu().numpy()
errs['g3c'] = np.abs(ppg - g3['ppg_c']).max()
print(errs, flush=True)
generator = torch.Generator().manual_seed(7)
audio = 0.1 * torch.randn(32, 1, 160000, generator=generator)
ref = O.from_audio(state, audio[[0, 31]]).numpy()
mel = ppgs_amd.preprocess.mel.from_audios(audio.cuda())
ppg = engine.encode(mel, [1000] * 32).cpu().numpy()
errs['c2'] = np.abs(ppg[[0, 31]] - ref).max()
print({k: f'{v:.2e}' for k, v in errs.items()}, flush=True)
for _ in range(5):
    engine.encode(mel, [1000] * 32)
torch.cuda.synchronize()
start = time.perf_counter()
for _ in range(20):
    engine.encode(mel, [1000] * 32)
torch.cuda.synchronize()
print('ms per encode (32 x 1000):', 50 * (time.perf_counter() - start))
engine.profile(True)
for _ in range(3):
    engine.encode(mel, [1000] * 32)
torch.cuda.synchronize()
print({k: round(v[0] / 3, 4) for k, v in engine.profile_read().items()})
