"""wav2vec 2.0 latents ('w2v2fb' representation), reference
ppgs/preprocess/w2v2fb/core.py:32-75.

The reference delegates the body to the third-party HF ``Wav2Vec2Model``
('facebook/wav2vec2-base'); so does this module for now -- on the HIP device
through stock PyTorch-ROCm ops.  It is the boundary of the hot path for the
w2v2fb configuration: everything after it -- the
768-channel, hidden-512 PPG network -- runs in the hand-written HIP engine.
A native HIP wav2vec2 body is the first "next" row.

Same arithmetic as the reference: zero-pad 40 samples each side, attention
mask over the first ``length + 80`` samples, ``last_hidden_state`` (B, ~T/2,
768) -> transpose -> nearest-neighbour upsample to ``samples // 160`` frames
-> fp16.
"""
import os

import torch

from .. import config

W2V2FB_CONFIG = 'facebook/wav2vec2-base'
WINDOW_SIZE = 400
HOP_SIZE = 320

_models = {}


def model_for(device):
    """Cached HF model on `device` (reference caches on the function object,
    w2v2fb/core.py:44-48).  Pretrained weights come from the HF cache/hub;
    with PPGS_AMD_W2V2_RANDOM_INIT=<seed> a seeded random model of the same
    architecture is built instead (offline tests and benchmarks)."""
    key = str(device)
    if key not in _models:
        import transformers
        transformers.utils.logging.set_verbosity_error()
        seed = os.environ.get('PPGS_AMD_W2V2_RANDOM_INIT')
        if seed is not None:
            torch.manual_seed(int(seed))
            model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config())
        else:
            model = transformers.Wav2Vec2Model.from_pretrained(W2V2FB_CONFIG)
        _models[key] = model.eval().to(device)
    return _models[key]


def from_audios(audio, lengths, sample_rate=None, gpu=None):
    """(batch, 1, samples) fp32 + sample lengths -> (batch, 768, samples // 160)
    fp16 on the GPU (reference w2v2fb/core.py:32-75)."""
    from .. import core
    with torch.no_grad():
        if sample_rate is None:
            sample_rate = config.SAMPLE_RATE
        device = core.device_for(gpu, audio)
        model = model_for(device)
        audio = core.resample(audio, sample_rate, config.SAMPLE_RATE).to(device)
        lengths = torch.ceil(
            torch.as_tensor(lengths, dtype=torch.float64).reshape(-1) *
            (config.SAMPLE_RATE / sample_rate)).to(torch.long)
        pad = WINDOW_SIZE // 2 - HOP_SIZE // 2
        padded = torch.nn.functional.pad(audio, (pad, pad)).squeeze(dim=1)
        # reference mask_from_lengths(lengths, pad): arange(max + 2 pad) - 2 pad < len
        positions = torch.arange(int(lengths.max()) + 2 * pad) - 2 * pad
        mask = (positions[None] < lengths[:, None]).to(torch.long).to(device)
        output = model(padded, mask).last_hidden_state.transpose(1, 2)
        upsampled = torch.nn.functional.interpolate(
            output, size=audio.shape[-1] // config.HOPSIZE, mode='nearest')
        return upsampled.to(torch.float16)


def from_audio(audio, sample_rate=None, gpu=None):
    """reference w2v2fb/core.py:77-95"""
    dims = audio.dim()
    if dims == 1:
        audio = audio.unsqueeze(dim=0)
    if audio.dim() == 2:
        audio = audio.unsqueeze(dim=0)
    out = from_audios(
        audio, torch.tensor([audio.shape[-1]]), sample_rate=sample_rate, gpu=gpu)
    return out
