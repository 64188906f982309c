"""wav2vec 2.0 latents ('w2v2fb' representation), reference
ppgs/preprocess/w2v2fb/core.py:32-75.

The reference delegates the body to the third-party HF ``Wav2Vec2Model``
('facebook/wav2vec2-base').  Here the whole model runs in the hand-written HIP
engine: the convolutional feature encoder (``engine.W2v2FeatureEncoder``:
layer 0 with its GroupNorm statistics from 65 moments of the audio in
ppg_w2v2.hip, layers 1-6 as strided-row MFMA GEMMs with an exact-GELU epilogue)
and the feature projection + 12-layer transformer (``engine.W2v2Body``: every
projection on the feature-split ``gemm32_kernel`` of ppg_gemm32.hip, the
grouped positional convolution as its own kernel ppg_posconv.hip, attention at
head dimension 64, LayerNorm-768 row kernels) -- SURVEY.md 8(f) rank 1,
DESIGN.md 4.5.  Everything after the latents -- the 768-channel, hidden-512
PPG network -- runs in the HIP engine too.  PPGS_AMD_W2V2_BODY=0 keeps the HF
projection / encoder modules on PyTorch-ROCm, PPGS_AMD_W2V2_NATIVE=0 runs the
whole HF model there.

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
_encoders = {}


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


def w2v2_precision():
    """Operand precision of the wav2vec2 engines for the package's PRECISION: the same mode.  'fp16x2' (<= 1e-4)
    runs the feature encoder's six GEMM layers, every projection of the body and its attention on fp16 hi + lo operand
    pairs (the positional convolution on f32-input MFMAs); PPGS_AMD_W2V2_FP32=1 runs the wav2vec2 engines of that mode
    in fp32 instead (the route until round 5: 2.3 x slower)."""
    from .. import core
    if core.PRECISION == 'fp16x2' and os.environ.get('PPGS_AMD_W2V2_FP32', '0') == '1':
        return 'fp32'
    return core.PRECISION


def feature_encoder_for(device, model):
    """Cached HIP feature encoder holding `model`'s convolution weights."""
    from .. import engine
    key = (str(device), id(model), w2v2_precision())
    entry = _encoders.get(key)
    # (the entry keeps the model alive: the id of a freed model can be handed to the next one, and an engine found
    # under it would hold the OLD model's weights)
    if entry is None or entry[0] is not model:
        entry = _encoders[key] = (model, engine.W2v2FeatureEncoder(
            model.feature_extractor.state_dict(), device.index, w2v2_precision()))
    return entry[1]


_bodies = {}


def body_for(device, model):
    """Cached HIP transformer body holding `model`'s projection / encoder weights."""
    from .. import engine
    key = (str(device), id(model), w2v2_precision())
    entry = _bodies.get(key)
    if entry is None or entry[0] is not model:
        entry = _bodies[key] = (model, engine.W2v2Body(model, device.index, w2v2_precision()))
    return entry[1]


def clear():
    """Drop the cached models and the HIP engines built from them."""
    _models.clear()
    _encoders.clear()
    _bodies.clear()


def last_hidden_state(model, padded, mask):
    """HF ``Wav2Vec2Model.forward`` (modeling_wav2vec2.py) with the feature
    encoder replaced by the HIP kernels: extract_features -> frame-rate
    attention mask -> feature_projection -> encoder."""
    if os.environ.get('PPGS_AMD_W2V2_NATIVE', '1') == '0':
        return model(padded, mask).last_hidden_state
    extract = feature_encoder_for(padded.device, model)(padded)
    attention_mask = model._get_feature_vector_attention_mask(
        extract.shape[1], mask, add_adapter=False)
    if os.environ.get('PPGS_AMD_W2V2_BODY', '1') != '0':
        # feature projection + encoder on the HIP engine (engine.W2v2Body): the frame-level mask is a
        # prefix mask, i.e. a valid length per item
        return body_for(padded.device, model)(extract, attention_mask.sum(dim=1))
    # PPGS_AMD_W2V2_BODY=0: the HF modules on PyTorch-ROCm; in the 16-bit engine modes under fp16
    # autocast, which is how the reference itself runs the whole model on a GPU
    # (ppgs/preprocess/core.py:207: torch.autocast('cuda')); fp32 engine mode: fp32 throughout
    with torch.autocast('cuda', dtype=torch.float16, enabled=w2v2_precision() not in ('fp32', 'fp16x2')):
        hidden, _ = model.feature_projection(extract)
        out = model.encoder(hidden, attention_mask=attention_mask).last_hidden_state
    return out.float()


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
        output = last_hidden_state(model, padded, mask).transpose(1, 2)
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
