"""Time the mel frontend alone (C2 shape: 32 x 160000 samples), HIP events around 50 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppgs_amd import engine as E

audio = (0.1 * torch.randn(32, 1, 160000)).cuda()
for _ in range(3):
    E.frontend(audio, spectrogram=False, mel=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    E.frontend(audio, spectrogram=False, mel=True)
b.record()
torch.cuda.synchronize()
print(f'frontend: {a.elapsed_time(b) / 50 * 1e3:.1f} us per launch (32 x 1000 frames)')
