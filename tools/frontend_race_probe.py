"""The mel frontend while other kernels run on another stream: how many elements differ from the exclusive run, by how much, where."""
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
mel_ref = ppgs_amd.preprocess.mel.from_audios(audio)
torch.cuda.synchronize()
a, b = torch.cuda.Stream(), torch.cuda.Stream()
other = sys.argv[1] if len(sys.argv) > 1 else 'encode'
big = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
if other in ('body', 'features'):
    import transformers
    transformers.utils.logging.set_verbosity_error()
    torch.manual_seed(5)
    hf = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=4)).eval().cuda()
    if other == 'body':
        body = E.W2v2Body(hf, 0, 'bf16')
        body_in = torch.randn(16, 499, 512, device='cuda')
    else:
        feats_model = E.W2v2FeatureEncoder(hf.state_dict(), 0, 'bf16')
        audio16 = audio[:16].contiguous()
if other == 'sdpa':
    q_ = torch.randn(32, 2, 1000, 128, device='cuda', dtype=torch.bfloat16)
total = bad = 0
for rep in range(40):
    with torch.cuda.stream(a):
        if other == 'encode':
            for _ in range(2):
                model.encode(mel_ref, lengths)
        elif other == 'gemm':
            for _ in range(4):
                big @ big
        elif other == 'body':
            for _ in range(2):
                body(body_in, [499] * 16)
        elif other == 'features':
            for _ in range(2):
                feats_model(audio16)
        elif other == 'sdpa':
            for _ in range(6):
                torch.nn.functional.scaled_dot_product_attention(q_, q_, q_)
        elif other == 'frontend':
            for _ in range(4):
                ppgs_amd.preprocess.mel.from_audios(audio)
    with torch.cuda.stream(b):
        mels = [ppgs_amd.preprocess.mel.from_audios(audio) for _ in range(6)]
    torch.cuda.synchronize()
    for m in mels:
        total += 1
        if not torch.equal(m, mel_ref):
            bad += 1
            if bad <= 3:
                d = (m.float() - mel_ref.float()).abs()
                idx = torch.nonzero(d > 0)
                print(f'  differing elements {idx.shape[0]} of {d.numel()}, max abs {float(d.max()):.4f}, items {sorted(set(idx[:, 0].tolist()))[:8]}, '
                      f'mels {sorted(set(idx[:, 1].tolist()))[:10]}, frames {sorted(set(idx[:, 2].tolist()))[:12]}')
print(f'beside {other}: {bad} of {total} mel tensors differ')
