"""Compare the feature-split layer kernel (PPGS_AMD_LAYER32=1) with the token-split one on the GPU."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppgs_amd import engine as E, weights as W    # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
for cin, hidden, shapes in ((768, 512, [(16, 1000), (6, 600), (9, 333)]), (80, 256, [(32, 1000), (20, 777)])):
    state = W.seeded_state_dict(seed=31, input_channels=cin, hidden_channels=hidden)
    for batch, frames in shapes:
        g = torch.Generator().manual_seed(frames)
        feats = torch.randn(batch, cin, frames, generator=g).half().cuda()
        lengths = [frames] * batch
        lengths[-1] = max(frames // 3, 1)
        outs, ms = {}, {}
        for flag in ('0', '1'):
            os.environ['PPGS_AMD_LAYER32'] = flag
            eng = E.Engine(state, 0, precision)
            outs[flag] = eng.encode(feats, lengths, softmax=False).cpu().numpy()
            torch.cuda.synchronize()
            start = time.perf_counter()
            for _ in range(10):
                eng.encode(feats, lengths)
            torch.cuda.synchronize()
            ms[flag] = 100 * (time.perf_counter() - start)
            del eng
        a, b = outs['0'], outs['1']
        bad = ~np.isfinite(b)
        diff = np.abs(np.nan_to_num(a) - np.nan_to_num(b))
        print(f'hidden {hidden} {batch}x{frames}: non-finite {bad.sum()} of {b.size}; max|diff| {diff.max():.3e}; |a| max {np.abs(a).max():.3f}; '
              f'ms/encode token-split {ms["0"]:.3f} feature-split {ms["1"]:.3f}', flush=True)
