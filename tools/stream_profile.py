"""A steady run of batched streaming pushes (64 streams x 16 frames) for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

batch, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 16
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 500          # frames an item can hold: the distance between two items' rows
model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, 'bf16', is_causal=True)
for _ in range(4 * 480 // max(cap - 20, n)):
    stream = model.batched_stream(batch, cap)
    chunk = torch.randn(batch, 80, n).half().cuda()
    for _ in range(max(cap - 20, n) // n):
        stream.push(chunk)
    torch.cuda.synchronize()
