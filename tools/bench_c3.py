"""configs[2] kernel path: w2v2fb-shaped features (16, 768, 1000) fp16 -> hidden-512 PPG network (bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppgs_amd
from ppgs_amd import data, engine as E

state = ppgs_amd.weights.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
model = E.Engine(state, 0, sys.argv[1] if len(sys.argv) > 1 else 'bf16')
feats = torch.randn(16, 768, 1000).half().cuda()
lengths = [1000] * 16
for _ in range(5):
    model.encode(feats, lengths)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
steps = 20
for _ in range(steps):
    model.encode(feats, lengths)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / steps
model.profile(True)
for _ in range(5):
    model.encode(feats, lengths)
torch.cuda.synchronize()
k = {n: round(v[0] / 5, 3) for n, v in model.profile_read().items()}
flops = 16 * (35_594_240 * 1250 + 10_240 * (500 ** 2 * 2 + 250 ** 2))
print(f'C3 kernel path: {ms:.3f} ms/step = {16000 / ms / 1e3:.2f} M frames/s, {flops / ms / 1e9:.0f} TFLOP/s end to end; per class {k}')
