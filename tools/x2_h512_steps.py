"""N steps of the hidden-512 PPG network in fp16x2 at configs[2]'s size, for a rocprofv3 pass (tools/x2_h512_check.py times it)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                           # noqa: E402
from ppgs_amd import engine as E                          # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512), 0, 'fp16x2')
feats = torch.randn(16, 768, 1000, generator=torch.Generator().manual_seed(3)).half().cuda()
for _ in range(steps):
    model.encode(feats, [1000] * 16)
torch.cuda.synchronize()
print('steps', steps)
