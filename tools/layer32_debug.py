"""Compare the feature-split layer kernel (PPGS_AMD_LAYER32=1) with the token-split one on the GPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppgs_amd import engine as E, weights as W    # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
for layers in (2, 5):
    state = W.seeded_state_dict(seed=1234, num_layers=layers)
    for batch, frames in [(32, 1000), (20, 777), (7, 333)]:
        g = torch.Generator().manual_seed(frames)
        feats = torch.randn(batch, 80, frames, generator=g).half().cuda()
        lengths = [frames] * batch
        lengths[-1] = max(frames // 3, 1)
        outs = {}
        for flag in ('0', '1'):
            os.environ['PPGS_AMD_LAYER32'] = flag
            eng = E.Engine(state, 0, precision)
            outs[flag] = eng.encode(feats, lengths, softmax=False).cpu().numpy()
            del eng
        a, b = outs['0'], outs['1']
        bad = ~np.isfinite(b)
        diff = np.abs(np.nan_to_num(a) - np.nan_to_num(b))
        print(f'layers {layers} {batch}x{frames}: non-finite {bad.sum()} of {b.size}; max|diff| {diff.max():.3e}; |a| max {np.abs(a).max():.3f}', flush=True)
        if bad.any():
            f0 = np.where(bad[0].any(axis=0))[0]
            print('   item 0 bad frames:', f0[:12], '... count', len(f0), ' item 1:', int(bad[1].any(axis=0).sum()))
        else:
            worst = np.unravel_index(diff.argmax(), diff.shape)
            print('   worst at', worst, a[worst], b[worst])
