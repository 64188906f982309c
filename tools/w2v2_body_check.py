"""GPU check of the HIP wav2vec2 body against the HF modules it replaces (seeded random weights):
    python tools/w2v2_body_check.py [fp32 fp16 bf16] [--layers N] [--frames T]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppgs_amd import engine as E    # noqa: E402


def hf_body(model, features, valid):
    """feature_projection + encoder of HF Wav2Vec2Model.forward with a frame-level mask"""
    frames = features.shape[1]
    mask = (torch.arange(frames, device=features.device)[None] < torch.as_tensor(valid, device=features.device)[:, None])
    hidden, _ = model.feature_projection(features)
    return model.encoder(hidden, attention_mask=mask).last_hidden_state


def main():
    import transformers
    transformers.utils.logging.set_verbosity_error()
    modes = [a for a in sys.argv[1:] if a in ('fp32', 'fp16', 'bf16')] or ['fp32']
    layers = int(sys.argv[sys.argv.index('--layers') + 1]) if '--layers' in sys.argv else 2
    frames = int(sys.argv[sys.argv.index('--frames') + 1]) if '--frames' in sys.argv else 150
    torch.manual_seed(5)
    model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=layers)).eval().cuda()
    generator = torch.Generator().manual_seed(6)
    features = torch.randn(3, frames, 512, generator=generator).cuda()
    valid = [frames, frames * 2 // 3, 33]
    with torch.no_grad():
        ref = hf_body(model, features, valid)
    ok = True
    for mode in modes:
        body = E.W2v2Body(model, 0, mode)
        out = body(features, valid)
        torch.cuda.synchronize()
        errs = [(out[b, :v] - ref[b, :v]).abs().max().item() for b, v in enumerate(valid)]
        scale = ref.abs().max().item()
        tol = {'fp32': 2e-4, 'fp16': 2e-2, 'bf16': 1e-1}[mode]
        good = max(errs) < tol and torch.isfinite(out).all().item()
        ok = ok and good
        print(mode, 'OK' if good else 'FAIL', [f'{e:.2e}' for e in errs], f'|ref| max {scale:.2f}', flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
