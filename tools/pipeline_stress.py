"""Many two-pipeline steps against the one-pipeline result, per operand mode, and the wav2vec2 body the same way:
does any kernel of one pipeline come out different beside the other pipeline's kernels?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                   # noqa: E402
from ppgs_amd import engine as E                  # noqa: E402

BATCH, FRAMES = 32, 1000
state = ppgs_amd.weights.seeded_state_dict(seed=1234)
gen = torch.Generator().manual_seed(1234)
feats = torch.randn(BATCH, 80, FRAMES, generator=gen).half().cuda()
lengths = [FRAMES] * BATCH
for precision in ('bf16', 'fp16', 'fp16x2', 'fp32'):
    os.environ['PPGS_AMD_STREAMS'] = '1'
    ref = E.Engine(state, 0, precision).encode(feats, lengths)
    del os.environ['PPGS_AMD_STREAMS']
    model = E.Engine(state, 0, precision)
    worst, exact = 0.0, 0
    runs = 60 if precision != 'fp32' else 20
    for _ in range(runs):
        out = model.encode(feats, lengths)
        torch.cuda.synchronize()
        d = float((out - ref).abs().max())
        worst = max(worst, d)
        exact += d == 0.0
    print(f'{precision}: {runs} two-pipeline steps, {exact} bit-identical to one pipeline, worst max-abs difference {worst:.3e}')
import transformers                               # noqa: E402
transformers.utils.logging.set_verbosity_error()
torch.manual_seed(5)
hf = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=4)).eval().cuda()
x = torch.randn(16, 499, 512, generator=torch.Generator().manual_seed(4)).cuda()
os.environ['PPGS_AMD_W2V2_STREAMS'] = '1'
ref = E.W2v2Body(hf, 0, 'bf16')(x, [499] * 16).clone()
del os.environ['PPGS_AMD_W2V2_STREAMS']
body = E.W2v2Body(hf, 0, 'bf16')
bad = 0
for _ in range(40):
    out = body(x, [499] * 16)
    torch.cuda.synchronize()
    bad += not torch.equal(out, ref)
print(f'w2v2 body: 40 two-pipeline forwards, {bad} differ from one pipeline')
