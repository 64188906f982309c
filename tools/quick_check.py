"""Fast correctness gate for kernel experiments (GPU): the 16-bit modes of the
library PPGS_AMD_LIB points at against the reference fixtures G2 / G3 (chunked,
ragged) and a 32 x 1000 batch against the fp32 oracle on two spot utterances.

    PPGS_AMD_LIB=ppgs_amd/libppgs_amd_exp.so python tools/quick_check.py [bf16 fp16 ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from oracle import ppg_oracle as O                # noqa: E402
from ppgs_amd import engine as E, weights as W    # noqa: E402

TOL = {'bf16': 6e-3, 'fp16': 8e-4, 'fp32': 1e-4}


def golden(name):
    with np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz')) as data:
        return {k: data[k] for k in data.files}


def main():
    modes = sys.argv[1:] or ['bf16']
    state = W.seeded_state_dict(seed=1234)
    g2, g3 = golden('g2_single_window'), golden('g3_chunked')
    generator = torch.Generator().manual_seed(7)
    audio = 0.1 * torch.randn(32, 1, 160000, generator=generator)
    ref = O.from_audio(state, audio[[0, 31]]).numpy()
    mel = ppgs_amd.preprocess.mel.from_audios(audio.cuda())
    ok = True
    for mode in modes:
        engine = E.Engine(state, 0, mode)
        errs = {}
        ppg = engine.encode(torch.from_numpy(g2['features']).cuda(), g2['lengths'].tolist()).cpu().numpy()
        errs['g2'] = np.abs(ppg - g2['ppg']).max()
        ppg = engine.encode(torch.from_numpy(g3['features_a']).cuda(), g3['lengths_a'].tolist()).cpu().numpy()
        errs['g3a'] = np.abs(ppg - g3['ppg_a']).max()
        ppg = engine.encode(torch.from_numpy(g3['features_c']).cuda(), g3['lengths_c'].tolist()).cpu().numpy()
        errs['g3c'] = np.abs(ppg - g3['ppg_c']).max()
        ppg = engine.encode(mel, [1000] * 32).cpu().numpy()
        errs['c2'] = np.abs(ppg[[0, 31]] - ref).max()
        good = all(v < TOL[mode] for v in errs.values()) and np.isfinite(ppg).all()
        ok = ok and good
        print(mode, 'OK' if good else 'FAIL', {k: f'{v:.2e}' for k, v in errs.items()}, flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
