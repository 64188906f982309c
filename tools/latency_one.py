"""One utterance through the encoder, repeatedly (for rocprofv3 kernel traces of the small-batch path).

    python tools/latency_one.py [batch frames steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                              # noqa: E402

import ppgs_amd                                           # noqa: E402
from ppgs_amd import engine as E                          # noqa: E402

batch, frames, steps = (int(v) for v in (sys.argv[1:4] + ['1', '1000', '50'][len(sys.argv) - 1:]))
model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, 'bf16')
feats = torch.randn(batch, 80, frames).half().cuda()
lengths = [frames] * batch
for _ in range(5):
    model.encode(feats, lengths)
torch.cuda.synchronize()
start = time.perf_counter()
for _ in range(steps):
    model.encode(feats, lengths)
torch.cuda.synchronize()
print(f'{batch} x {frames} frames: {(time.perf_counter() - start) / steps * 1e6:.1f} us per encode')
