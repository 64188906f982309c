"""Which stage differs when two caller streams run the C2 step at the same time on one engine."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

BATCH, FRAMES = 32, 1000
precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
model = E.Engine(ppgs_amd.weights.seeded_state_dict(seed=1234), 0, precision)
gen = torch.Generator().manual_seed(1234)
audio = (0.1 * torch.randn(BATCH, 1, FRAMES * 160, generator=gen)).cuda()
lengths = [FRAMES] * BATCH
mel_ref = ppgs_amd.preprocess.mel.from_audios(audio)
ref = model.encode(mel_ref, lengths)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
for name, fn, want in (('frontend', lambda: ppgs_amd.preprocess.mel.from_audios(audio), mel_ref),
                       ('encode', lambda: model.encode(mel_ref, lengths), ref)):
    bad = 0
    worst = 0.0
    for rep in range(20):
        outs = []
        for i in range(8):
            with torch.cuda.stream(streams[i % 2]):
                outs.append(fn())
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, want):
                bad += 1
                worst = max(worst, float((o.float() - want.float()).abs().max()))
    print(f'{name}: {bad} of 160 concurrent results differ from the one-stream result, max abs {worst:.3e} '
          f'(pipelines per call: {model.pipelines(40960)})')

# whole steps (frontend + encode) round-robin on the two streams, many in flight
for inflight in (8, 64):
    bad_mel = bad_out = 0
    for rep in range(6):
        mels, outs = [], []
        for i in range(inflight):
            with torch.cuda.stream(streams[i % 2]):
                m = ppgs_amd.preprocess.mel.from_audios(audio)
                mels.append(m)
                outs.append(model.encode(m, lengths))
        torch.cuda.synchronize()
        bad_mel += sum(not torch.equal(m, mel_ref) for m in mels)
        bad_out += sum(not torch.equal(o, ref) for o in outs)
    print(f'steps, {inflight} in flight, tensors kept: {bad_mel} mel and {bad_out} posterior tensors of {6 * inflight} differ')
bad_out = 0
for rep in range(6):
    outs = [None, None]
    for i in range(64):
        with torch.cuda.stream(streams[i % 2]):
            outs[i % 2] = model.encode(ppgs_amd.preprocess.mel.from_audios(audio), lengths)
    torch.cuda.synchronize()
    bad_out += sum(not torch.equal(o, ref) for o in outs)
print(f'steps, 64 in flight, tensors dropped as the loop goes: {bad_out} of 12 final posterior tensors differ')
