"""Frontend beside an encode: map of the wrong frame pairs of one launch -- (workgroup, round) -> pairs and wrong bins."""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

BATCH, FRAMES = 32, 1000
model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, 'bf16')
gen = torch.Generator().manual_seed(1234)
audio = (0.1 * torch.randn(BATCH, 1, FRAMES * 160, generator=gen)).cuda()
lengths = [FRAMES] * BATCH
spec_ref, mel_ref = E.frontend(audio, spectrogram=True, mel=True)
torch.cuda.synchronize()
a, b = torch.cuda.Stream(), torch.cuda.Stream()
groups_per_row = (FRAMES + 15) // 16
total = groups_per_row * BATCH
rounds = (total + 511) // 512
grid = (total + rounds - 1) // rounds
shown = 0
for rep in range(40):
    with torch.cuda.stream(a):
        for _ in range(2):
            model.encode(mel_ref, lengths)
    with torch.cuda.stream(b):
        outs = [E.frontend(audio, spectrogram=True, mel=True) for _ in range(4)]
    torch.cuda.synchronize()
    for s, m in outs:
        if shown < 2 and not torch.equal(s, spec_ref):
            shown += 1
            wrong = (s != spec_ref).sum(1).cpu()           # (item, frame) -> wrong bins
            by_wg = collections.defaultdict(list)
            for item, frame in torch.nonzero(wrong).tolist():
                if frame % 2:
                    continue
                grp = item * groups_per_row + frame // 16
                by_wg[grp % grid].append((grp // grid, (frame % 16) // 2, int(wrong[item, frame])))
            print(f'launch with {len(by_wg)} affected workgroups of {grid} (rounds per workgroup {rounds}); (round, pair, wrong bins) per workgroup:')
            for wg in sorted(by_wg)[:24]:
                print('  wg', wg, sorted(by_wg[wg]))
