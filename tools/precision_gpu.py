"""Posterior error of the engine's three arithmetic modes on the GPU, against the
fp32 CPU oracle and the reference fixtures (run on the MI355X box):

    python tools/precision_gpu.py

Reports, per checkpoint (seeded / sharpened) and mode (fp32 / fp16 / bf16):
max-abs error on the G2 (single window) and G3 (chunked) fixtures, and at the C2
size (32 x 1000 frames, sharpened checkpoint, fixture g6_sharp_stats): max-abs on
the stored frames, per-frame argmax agreement with the reference's fp32 forward,
and the agreement restricted to frames whose top-1/top-2 margin exceeds 0.02.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E, weights as W    # noqa: E402


def golden(name):
    with np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz')) as data:
        return {k: data[k] for k in data.files}


def main():
    out = {}
    g2, g3, g6 = golden('g2_single_window'), golden('g3_chunked'), golden('g6_sharp_stats')
    states = {'seeded': W.seeded_state_dict(seed=1234), 'sharp': W.seeded_state_dict(seed=4321, sharpen=2.0)}
    generator = torch.Generator().manual_seed(1234)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=generator)).cuda()
    mel = ppgs_amd.preprocess.mel.from_audios(audio)
    for precision in ('fp32', 'fp16', 'bf16'):
        for name, state in states.items():
            engine = E.Engine(state, 0, precision)
            row = {}
            f2 = torch.from_numpy(g2['features']).cuda()
            ppg = engine.encode(f2, g2['lengths'].tolist()).cpu().numpy()
            row['g2'] = float(np.abs(ppg - g2['ppg' if name == 'seeded' else 'ppg_sharp']).max())
            f3 = torch.from_numpy(g3['features_a']).cuda()
            ppg = engine.encode(f3, g3['lengths_a'].tolist()).cpu().numpy()
            row['g3'] = float(np.abs(ppg - g3['ppg_a' if name == 'seeded' else 'ppg_a_sharp']).max())
            if name == 'sharp':
                ppg = engine.encode(mel, [1000] * 32).cpu()
                row['c2_first64'] = float((ppg[0, :, :64] - torch.from_numpy(g6['ppg_item0_first64'])).abs().max())
                row['c2_last64'] = float((ppg[31, :, -64:] - torch.from_numpy(g6['ppg_item31_last64'])).abs().max())
                row['c2_max_of_max'] = float((ppg.amax(-1) - torch.from_numpy(g6['ppg_max'])).abs().max())
                ref_arg = torch.from_numpy(g6['argmax'].astype(np.int64))
                agree = ppg.argmax(1) == ref_arg
                margin = torch.from_numpy(g6['margin'].astype(np.float32))
                row['c2_argmax_agreement'] = float(agree.float().mean())
                row['c2_argmax_agreement_margin_gt_0.02'] = float(agree[margin > 0.02].float().mean())
                row['c2_disagreeing_frames_max_margin'] = float(margin[~agree].max()) if (~agree).any() else 0.0
            out[f'{precision}/{name}'] = row
            print(precision, name, json.dumps(row), flush=True)
            del engine
    print('reference shipped (bf16 autocast) vs its fp32 at C2/sharp: max-abs',
          float(g6['shipped_max_abs']), 'argmax agreement', float(g6['shipped_argmax_agreement']))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'precision_gpu.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
