import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppgs_amd import engine as E, weights as W
from oracle import ppg_oracle as O
state = W.seeded_state_dict(seed=1234)
for batch, frames in [(7, 333), (3, 333), (7, 160), (1, 333)]:
    g = torch.Generator().manual_seed(frames)
    feats = torch.randn(batch, 80, frames, generator=g).half()
    lengths = [frames] * batch
    lengths[-1] = max(frames // 3, 1)
    # poison the allocator's free memory so uninitialised reads show up
    junk = torch.full((64 << 20,), float('nan'), device='cuda'); del junk
    eng = E.Engine(state, 0, 'bf16')
    out = eng.encode(feats.cuda(), lengths).cpu().numpy()
    ref = O.from_features(state, feats, torch.tensor(lengths)).numpy()
    print(batch, frames, 'non-finite', int((~np.isfinite(out)).sum()), 'max err', float(np.abs(np.nan_to_num(out) - ref).max()), flush=True)
