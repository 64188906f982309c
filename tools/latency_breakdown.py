"""Per-kernel-class time (HIP events) of one encode at small batch shapes.

    python tools/latency_breakdown.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                              # noqa: E402

import ppgs_amd                                           # noqa: E402
from ppgs_amd import engine as E                          # noqa: E402

state = ppgs_amd.weights.seeded_state_dict(seed=1234)
model = E.Engine(state, 0, 'bf16')
for B, T in ((1, 100), (1, 1000), (1, 3000), (4, 1000), (8, 1000)):
    feats = torch.randn(B, 80, T).half().cuda()
    lengths = [T] * B
    for _ in range(5):
        model.encode(feats, lengths)
    model.profile(True)
    for _ in range(10):
        model.encode(feats, lengths)
    torch.cuda.synchronize()
    k = model.profile_read()
    model.profile(False)
    print(B, T, {n: (round(v[0] / 10 * 1e3, 1), v[1] // 10) for n, v in k.items() if v[1]},
          'sum us', round(sum(v[0] for v in k.values()) / 10 * 1e3, 1))
