"""Frontend beside an encode on another stream: spectrogram and mel of the same launch, which of the two differs."""
import os
import sys

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
mel_only = ppgs_amd.preprocess.mel.from_audios(audio)
print('mel of the two instantiations equal:', torch.equal(mel_ref, mel_only))
torch.cuda.synchronize()
a, b = torch.cuda.Stream(), torch.cuda.Stream()
n = bad_spec = bad_mel = 0
for rep in range(40):
    with torch.cuda.stream(a):
        for _ in range(2):
            model.encode(mel_ref, lengths)
    with torch.cuda.stream(b):
        outs = [E.frontend(audio, spectrogram=True, mel=True) for _ in range(4)]
    torch.cuda.synchronize()
    for s, m in outs:
        n += 1
        ds, dm = not torch.equal(s, spec_ref), not torch.equal(m, mel_ref)
        bad_spec += ds
        bad_mel += dm
        if (ds or dm) and bad_spec + bad_mel <= 4:
            ix = torch.nonzero((s != spec_ref)) if ds else None
            im = torch.nonzero((m != mel_ref))
            print(' spec differs' if ds else ' spec equal', '| mel differs at', im.shape[0], 'elements',
                  '| spec elements', None if ix is None else ix.shape[0],
                  '| spec frames', None if ix is None else sorted(set(ix[:, 2].tolist()))[:10], '| mel frames', sorted(set(im[:, 2].tolist()))[:10],
                  '| spec bins', None if ix is None else (int(ix[:, 1].min()), int(ix[:, 1].max())))
print(f'{bad_spec} spectrograms and {bad_mel} mels of {n} differ')

# structure of one wrong frame pair
done = False
for rep in range(40):
    if done:
        break
    with torch.cuda.stream(a):
        for _ in range(2):
            model.encode(mel_ref, lengths)
    with torch.cuda.stream(b):
        outs = [E.frontend(audio, spectrogram=True, mel=True) for _ in range(4)]
    torch.cuda.synchronize()
    for s, m in outs:
        if not torch.equal(s, spec_ref):
            ix = torch.nonzero(s != spec_ref)
            item, _, frame = ix[0].tolist()
            frame -= frame % 2
            for f in (frame, frame + 1):
                got, want = s[item, :, f].float().cpu(), spec_ref[item, :, f].float().cpu()
                wrong = (got != want)
                print(f'item {item} frame {f}: {int(wrong.sum())} bins differ; first wrong bins {torch.nonzero(wrong)[:12, 0].tolist()}; '
                      f'right bins {torch.nonzero(~wrong)[:20, 0].tolist()}')
                print('   got ', [round(float(v), 3) for v in got[:12]])
                print('   want', [round(float(v), 3) for v in want[:12]])
                # does the wrong spectrum belong to another frame of the same item?
                spec_item = spec_ref[item].float().cpu()
                err = (spec_item - got[:, None]).abs().sum(0)
                print('   closest reference frame of this item:', int(err.argmin()), 'L1', float(err.min()), '(own frame L1', float(err[f]), ')')
                # ... or to the mean of two frames' spectra / swapped pair?
            done = True
            break
