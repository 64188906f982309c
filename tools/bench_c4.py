"""configs[3]-style ragged workload: utterances of 50..3000 frames packed under
max_frames = 32000, features resident on the device; frames/s through
model.encode (every batch has a new shape: plan build + upload included)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppgs_amd
from ppgs_amd import data, engine as E

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(1234)
frames = rng.integers(50, 3001, count).tolist()
batches = data.pack_batches(frames, 32000)
model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, 'bf16')
pool = torch.randn(max(len(b) for b in batches), 80, 3000).half().cuda()
work = []
for batch in batches:
    lengths = [frames[i] for i in batch]
    work.append((pool[:len(batch), :, :max(lengths)].contiguous(), lengths))
for feats, lengths in work[:3]:
    model.encode(feats, lengths)
torch.cuda.synchronize()
start = time.perf_counter()
for feats, lengths in work:
    model.encode(feats, lengths)
torch.cuda.synchronize()
seconds = time.perf_counter() - start
total = sum(frames)
padded = sum(f.shape[0] * f.shape[2] for f, _ in work)
print(f'{count} utterances, {len(batches)} batches, {total} frames ({total / padded:.1%} of the padded frames) in '
      f'{seconds * 1e3:.1f} ms -> {total / seconds / 1e6:.2f} M frames/s, {seconds / len(batches) * 1e3:.3f} ms per batch')
