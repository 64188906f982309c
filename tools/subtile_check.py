"""Sub-tile workgroups of the layer kernel (PPGS_AMD_SUBTILE) against whole tiles and the oracle (GPU)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ppg_oracle as O                # noqa: E402
from ppgs_amd import engine as E, weights as W    # noqa: E402

state = W.seeded_state_dict(seed=1234)
ok = True
for precision in sys.argv[1:] or ['bf16', 'fp16']:
    for causal, batch, frames in ((True, 64, 160), (False, 20, 300), (False, 7, 1000), (False, 40, 333)):
        g = torch.Generator().manual_seed(frames)
        feats = torch.randn(batch, 80, frames, generator=g).half()
        lengths = [frames] * batch
        lengths[-1] = max(frames // 3, 1)
        lengths[0] = frames - 7
        ref = O.from_features(state, feats[:3].float(), torch.tensor(lengths[:3]), is_causal=causal).numpy() if max(lengths[:3]) == frames else None
        outs, us = {}, {}
        for flag in ('0', '1'):
            os.environ['PPGS_AMD_SUBTILE'] = flag
            eng = E.Engine(state, 0, precision, causal)
            outs[flag] = eng.encode(feats.cuda(), lengths).cpu().numpy()
            for _ in range(20):
                eng.encode(feats.cuda(), lengths)
            torch.cuda.synchronize()
            start = time.perf_counter()
            for _ in range(100):
                eng.encode(feats.cuda(), lengths)
            torch.cuda.synchronize()
            us[flag] = 1e4 * (time.perf_counter() - start)
            del eng
        d = np.abs(outs['0'] - outs['1']).max()
        err = np.abs(outs['1'][:3] - ref).max() if ref is not None else float('nan')
        # (with sub-tiles the head kernel replaces the three head launches: another order of roundings)
        good = np.isfinite(outs['1']).all() and d < (4e-3 if precision == 'bf16' else 6e-4) and not err > (4e-3 if precision == 'bf16' else 6e-4)
        ok = ok and good
        print(f'{precision} causal={causal} {batch}x{frames}: |sub - whole| {d:.2e}  |sub - oracle| {err:.2e}  us/encode whole {us["0"]:.1f} sub {us["1"]:.1f}', 'OK' if good else 'FAIL', flush=True)
sys.exit(0 if ok else 1)
