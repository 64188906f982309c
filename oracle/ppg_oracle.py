"""CPU oracle for the PPG forward path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch fp32 restatement (torch CPU functional ops + numpy) of the
reference path  audio -> spectrogram -> log-mel -> Transformer -> softmax.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; nothing under ``ppgs_amd/`` does.

Pinning: the reference has no tests or golden vectors, so this
oracle is pinned against outputs of the reference's own modules run in the
build container (``oracle/make_golden.py`` imports them from /root/reference
with a stub namespace and writes ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` checks this file against those fixtures.
Third-party arithmetic the reference delegates to un-vendored, unpinned
packages (setup.py:26-41) is restated here from the published algorithms:
  * ``librosa.filters.mel(sr=16000, n_fft=1024, n_mels=80)`` (Slaney scale,
    Slaney area normalisation)  -> :func:`mel_basis`
  * ``torch.nn.TransformerEncoderLayer`` (post-norm, ReLU, eps 1e-5) and
    ``F.multi_head_attention_forward``                  -> :func:`window_forward`
  * ``torchutil.inference.context`` = eval + inference_mode, autocast OFF
    (the fp32 parity definition of SURVEY.md 7.2).

Every function cites the reference file:line it follows.
"""
import math

import numpy as np
import torch

HOPSIZE = 160            # ppgs/config/defaults.py:20
NUM_FFT = 1024           # :23
NUM_MELS = 80            # :26
SAMPLE_RATE = 16000      # :29
CHUNK_OVERLAP = 50       # :158
CHUNK_LENGTH = 500       # :161
LN_EPS = 1e-5            # torch TransformerEncoderLayer default


###############################################################################
# Frontend
###############################################################################


def hz_to_mel(f):
    """Slaney mel scale (librosa.core.convert.hz_to_mel, htk=False)."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(
        f >= min_log_hz,
        min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep,
        mels)


def mel_to_hz(m):
    """Inverse Slaney mel scale (librosa.core.convert.mel_to_hz)."""
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(
        m >= min_log_mel,
        min_log_hz * np.exp(logstep * (m - min_log_mel)),
        f_sp * m)


def mel_basis(sr=SAMPLE_RATE, n_fft=NUM_FFT, n_mels=NUM_MELS):
    """(n_mels, n_fft//2+1) fp32 filterbank.

    Restates librosa.filters.mel(sr, n_fft, n_mels) with its defaults
    (fmin=0, fmax=sr/2, htk=False, norm='slaney'), the call made at reference
    ppgs/preprocess/mel.py:61-64: triangular filters on the Slaney mel scale,
    computed in float64, each scaled by 2/(f[i+2]-f[i]), cast to float32.
    """
    fftfreqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    mel_f = mel_to_hz(
        np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def spectrogram_fp32(audio):
    """Magnitude spectrogram before the fp16 rounding, (B, 513, frames) fp32.

    Follows reference ppgs/preprocess/spectrogram.py:27-47: reflect-pad
    (1024-160)//2 = 432 both sides, frames of 1024 at hop 160 (center=False),
    periodic Hann, one-sided DFT, sqrt(re^2 + im^2 + 1e-6).
    audio: (B, 1, N) fp32.
    """
    size = (NUM_FFT - HOPSIZE) // 2
    padded = torch.nn.functional.pad(audio, (size, size), mode='reflect')
    window = torch.hann_window(NUM_FFT, dtype=audio.dtype)
    frames = padded.squeeze(1).unfold(-1, NUM_FFT, HOPSIZE)      # (B, T, 1024)
    stft = torch.fft.rfft(frames * window, dim=-1)               # (B, T, 513)
    stft = torch.view_as_real(stft).transpose(1, 2)              # (B, 513, T, 2)
    return torch.sqrt(stft.pow(2).sum(-1) + 1e-6)


def spectrogram(audio):
    """reference spectrogram.from_audios (spectrogram.py:14-50): -> fp16."""
    return spectrogram_fp32(audio).to(torch.float16)


def linear_to_mel(spec):
    """reference mel.linear_to_mel (mel.py:56-76), fp32 matmul (autocast off).

    spec: (B, 513, T) fp16 -> (B, 80, T) in spec's dtype.
    """
    basis = torch.from_numpy(mel_basis())
    mel = torch.matmul(basis, spec.to(torch.float))
    return torch.log(torch.clamp(mel, min=1e-5)).to(spec.dtype)


def mel_from_audios(audio):
    """reference mel.from_audios (mel.py:14-19): (B,1,N) fp32 -> (B,80,N//160) fp16."""
    return linear_to_mel(spectrogram(audio)).to(torch.float16)


###############################################################################
# Chunk planner
###############################################################################


def plan_windows(T, lengths):
    """Window plan of reference Transformer.forward (transformer.py:49-64).

    Returns a list of windows (start, Tc, clens, keep_lo, keep_hi):
    ``start`` is the first frame of the window in the *left-replicate-padded*
    sequence (padded[j] = x[max(j-50, 0)]), ``Tc`` the window length,
    ``clens[b]`` the per-item valid length inside the window, and window
    columns [keep_lo, keep_hi) are the ones concatenated into the output.
    For T <= 500 there is one window covering the unpadded sequence.
    """
    lengths = [int(v) for v in lengths]
    if T <= CHUNK_LENGTH:
        return [dict(start=0, Tc=T, clens=lengths, keep_lo=0, keep_hi=T,
                     padded=False)]
    stride = CHUNK_LENGTH - 2 * CHUNK_OVERLAP
    windows = []
    rem = list(lengths)
    for i in range(math.ceil(T / stride)):
        start = i * stride
        stop = min(start + CHUNK_LENGTH, T + CHUNK_OVERLAP)
        clens = [min(max(r + CHUNK_OVERLAP, 0), CHUNK_LENGTH) for r in rem]
        clens = [0 if c == CHUNK_OVERLAP else c for c in clens]
        rem = [max(r - stride, 0) for r in rem]
        Tc = stop - start
        windows.append(dict(
            start=start, Tc=Tc, clens=clens, keep_lo=CHUNK_OVERLAP,
            keep_hi=min(CHUNK_LENGTH - CHUNK_OVERLAP, Tc), padded=True))
    return windows


###############################################################################
# Model
###############################################################################


def num_layers(state):
    n = 0
    while f'model.layers.{n}.linear1.weight' in state:
        n += 1
    return n


def window_forward(state, x, clens, is_causal=False, heads=2, quant=None):
    """One unchunked forward, reference transformer.py:65-81.

    x: (B, Cin, Tc) fp32; clens: (B,) ints (valid frames per item).
    Every one of the Tc positions is computed, padded ones included.
    Attention follows F.multi_head_attention_forward + SDPA: key-padding mask
    and causal mask are additive -inf; rows whose keys are all masked give 0.

    ``quant(stage, tensor) -> tensor`` (default: identity = the fp32 parity
    definition) rounds the MFMA OPERANDS of the engine's 16-bit modes at the
    points where the HIP kernels round them -- 'feat', 'w' (weights), 'x0'
    (in-conv output as the layer-0 operand), 'qkv', 'p' (softmax numerators),
    'ao', 'x1' / 'x2' (LayerNorm outputs as operands; the residual stays fp32),
    'h' -- for the error attribution of tools/precision_attribution.py.
    """
    q_ = quant if quant is not None else (lambda stage, tensor: tensor)
    B, _, Tc = x.shape
    clens = torch.as_tensor(clens, dtype=torch.long)
    t = torch.arange(Tc)
    mask = t[None, :] < clens[:, None]                          # (B, Tc)
    h = torch.nn.functional.conv1d(
        q_('feat', x), q_('w', state['input_layer.weight']), state['input_layer.bias'],
        padding='same') * mask[:, None, :]
    H = h.shape[1]
    d = H // heads
    z = h.permute(0, 2, 1) + state['position.encoding'][:Tc, 0][None]   # (B,Tc,H)
    zop = q_('x0', z)                       # operand copy of the residual stream

    bias = torch.zeros(B, 1, Tc, Tc)
    bias = bias.masked_fill(~mask[:, None, None, :], float('-inf'))
    if is_causal:
        causal = t[None, :] > t[:, None]                        # key > query
        bias = bias.masked_fill(causal[None, None], float('-inf'))

    for l in range(num_layers(state)):
        p = f'model.layers.{l}.'
        qkv = q_('qkv', zop @ q_('w', state[p + 'self_attn.in_proj_weight']).T +
                  state[p + 'self_attn.in_proj_bias'])
        q, k, v = qkv.split(H, dim=-1)
        q = q.reshape(B, Tc, heads, d).transpose(1, 2)          # (B,h,Tc,d)
        k = k.reshape(B, Tc, heads, d).transpose(1, 2)
        v = v.reshape(B, Tc, heads, d).transpose(1, 2)
        scores = (q @ k.transpose(-1, -2)) / math.sqrt(d) + bias
        smax = scores.max(dim=-1, keepdim=True).values
        smax = torch.where(torch.isinf(smax), torch.zeros_like(smax), smax)
        e = torch.exp(scores - smax)
        denom = e.sum(dim=-1, keepdim=True)
        if quant is None:
            attn = torch.where(denom > 0, e / denom, torch.zeros_like(e))
            o = attn @ v
        else:       # the kernels round the numerators and divide by the fp32 row sum afterwards
            o = torch.where(denom > 0, (q_('p', e) @ v) / denom, torch.zeros_like(v))
        o = q_('ao', o.transpose(1, 2).reshape(B, Tc, H))
        o = o @ q_('w', state[p + 'self_attn.out_proj.weight']).T + \
            state[p + 'self_attn.out_proj.bias']
        z = torch.nn.functional.layer_norm(
            z + o, (H,), state[p + 'norm1.weight'], state[p + 'norm1.bias'],
            LN_EPS)
        f = q_('h', torch.relu(
            q_('x1', z) @ q_('w', state[p + 'linear1.weight']).T + state[p + 'linear1.bias']))
        f = f @ q_('w', state[p + 'linear2.weight']).T + state[p + 'linear2.bias']
        z = torch.nn.functional.layer_norm(
            z + f, (H,), state[p + 'norm2.weight'], state[p + 'norm2.bias'],
            LN_EPS)
        zop = q_('x2', z)

    y = torch.nn.functional.conv1d(
        zop.permute(0, 2, 1), q_('w', state['output_layer.weight']),
        state['output_layer.bias'], padding='same')
    return y * mask[:, None, :]


def forward(state, features, lengths, is_causal=False, legacy_mode=False, quant=None):
    """reference Transformer.forward (transformer.py:45-81) -> logits (B,40,T)."""
    features = features.to(torch.float)
    T = features.shape[-1]
    if legacy_mode or T <= CHUNK_LENGTH:
        return window_forward(state, features, lengths, is_causal, quant=quant)
    padded = torch.nn.functional.pad(
        features, (CHUNK_OVERLAP, 0), mode='replicate')
    outputs = []
    for w in plan_windows(T, lengths):
        split = padded[..., w['start']:w['start'] + w['Tc']]
        out = window_forward(state, split, w['clens'], is_causal, quant=quant)
        outputs.append(out[..., w['keep_lo']:w['keep_hi']])
    return torch.cat(outputs, dim=-1)


def from_features(state, features, lengths, softmax=True, is_causal=False,
                  legacy_mode=False, quant=None):
    """reference ppgs.from_features / infer (core.py:72-128, 551-596), fp32."""
    with torch.inference_mode():
        logits = forward(state, features, lengths, is_causal, legacy_mode, quant)
        if softmax:
            return torch.softmax(logits, dim=1)
        return logits


def from_audio(state, audio, is_causal=False):
    """fp32-oracle entry of SURVEY.md 8(c): mel.from_audios(...).float() ->
    from_features with per-row lengths = N // 160 (reference core.py:52-69)."""
    with torch.inference_mode():
        features = mel_from_audios(audio)
        frames = features.shape[-1]
        lengths = torch.full((audio.shape[0],), frames, dtype=torch.long)
        return from_features(state, features.float(), lengths, True, is_causal)


###############################################################################
# Batch packing
###############################################################################


def sampler_batches(lengths, max_frames, seed=1234, epoch=0, buckets=1):
    """reference Dataset.buckets + Sampler.batch (dataset.py:109-128,
    sampler.py:46-82): list of index batches."""
    lengths = np.asarray(lengths)
    n = len(lengths)
    size = n // buckets
    indices = np.argsort(lengths)
    sorted_lengths = np.sort(lengths)
    bucket_list = [
        np.stack((indices[i:i + size], sorted_lengths[i:i + size])).T
        for i in range(0, n, size)]
    if len(bucket_list) == buckets + 1:
        residual = bucket_list.pop()
        bucket_list[-1] = np.concatenate((bucket_list[-1], residual), axis=0)
    generator = torch.Generator()
    generator.manual_seed(seed + epoch)
    batches = []
    for bucket in bucket_list:
        bucket = bucket[torch.randperm(len(bucket), generator=generator).tolist()]
        batch, max_length = [], 0
        for index, length in bucket:
            max_length = max(max_length, length)
            if batch and (len(batch) + 1) * max_length > max_frames:
                batches.append(batch)
                max_length = length
                batch = [int(index)]
            else:
                batch.append(int(index))
        if batch:
            batches.append(batch)
    order = torch.randperm(len(batches), generator=generator).tolist()
    return [batches[i] for i in order]


def resample(audio, sample_rate, target_rate=16000):
    """torchaudio.transforms.Resample with its defaults (sinc_interp_hann,
    lowpass_filter_width 6, rolloff 0.99), which reference ppgs/core.py:599-608
    applies to audio that is not at 16 kHz.  torchaudio is absent here: this is
    its published algorithm restated (functional.py: _get_sinc_resample_kernel +
    _apply_sinc_resample_kernel), i.e. parity-unpinned against the real package."""
    if sample_rate == target_rate:
        return audio
    orig, new = int(sample_rate), int(target_rate)
    gcd = math.gcd(orig, new)
    orig, new = orig // gcd, new // gcd
    lowpass_filter_width, rolloff = 6, 0.99
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t)
    kernels = (kernels * window * scale).to(torch.float32)
    shape = audio.shape
    flat = audio.reshape(-1, shape[-1]).to(torch.float32)
    length = flat.shape[-1]
    padded = torch.nn.functional.pad(flat, (width, width + orig))
    out = torch.nn.functional.conv1d(padded[:, None], kernels, stride=orig)
    out = out.transpose(1, 2).reshape(flat.shape[0], -1)
    target_length = math.ceil(new * length / orig)
    return out[..., :target_length].reshape(shape[:-1] + (target_length,))


def distance(ppg_x, ppg_y, similarity=None, exponent=1.2, reduction='mean'):
    """Reference ppgs/core.py:399-472 (similarity=None: normalize=False)."""
    x = ppg_x.float().clamp(1e-8, 1 - 1e-8)
    y = ppg_y.float().clamp(1e-8, 1 - 1e-8)
    if similarity is not None:
        mix = similarity.float().T ** exponent
        x, y = torch.mm(mix, x), torch.mm(mix, y)
    x, y = x.T, y.T
    log_average = torch.log((x + y) / 2)
    kl_x = x * (torch.log(x) - log_average)
    kl_y = y * (torch.log(y) - log_average)
    jsd = torch.sqrt(((kl_x + kl_y) / 2).clamp(min=0)).sum(dim=1)
    if reduction == 'mean':
        return jsd.mean(dim=0)
    if reduction == 'sum':
        return jsd.sum(dim=0)
    return jsd


def sparsify(ppg, method='percentile', threshold=0.85):
    """Reference ppgs/core.py:510-543 for one threshold, (batch, 40, frames) in
    and out (the reference's extra leading axis for 'percentile' is the caller's)."""
    ppg = ppg.float()
    if method == 'percentile':
        cut = torch.quantile(ppg, float(threshold), dim=-2, keepdim=True)
        ppg = torch.where(ppg > cut, ppg, torch.zeros_like(ppg))
    elif method == 'constant':
        ppg = torch.where(ppg > float(threshold), ppg, torch.zeros_like(ppg))
    elif method == 'topk':
        values, indices = ppg.topk(int(threshold), dim=-2)
        ppg = torch.zeros_like(ppg).scatter(-2, indices, values)
    else:
        raise ValueError(method)
    return torch.softmax(torch.log(ppg + 1e-8), -2)



def grid_sample(ppg, grid):
    """Reference ppgs/edit/grid.py:13-45: (..., frames) at fractional frame
    indices grid (length,) -> (..., length)."""
    frames = ppg.shape[-1]
    low = torch.floor(grid)
    weight = grid - low
    upper = torch.where(grid < 0, torch.zeros_like(low), low + 1).clamp(max=frames).long()
    extended = torch.cat([ppg, ppg[..., -1:]], dim=-1)          # final frame replicated
    return (1. - weight) * extended[..., upper - 1] + weight * extended[..., upper]


###############################################################################
# wav2vec 2.0 feature encoder (w2v2fb representation)
###############################################################################


W2V2_KERNELS = (10, 3, 3, 3, 3, 2, 2)      # transformers Wav2Vec2Config().conv_kernel (transformers 5.15.0 here)
W2V2_STRIDES = (5, 2, 2, 2, 2, 2, 2)       # .conv_stride


def w2v2_feature_encoder(state, audio):
    """HF transformers ``Wav2Vec2FeatureEncoder.forward`` restated (the third-party
    body reference ppgs/preprocess/w2v2fb/core.py:66 runs; models/wav2vec2/
    modeling_wav2vec2.py: Wav2Vec2GroupNormConvLayer for layer 0 -- Conv1d without
    bias, GroupNorm(512 groups, 512 channels, eps 1e-5, affine), GELU -- and
    Wav2Vec2NoLayerNormConvLayer for layers 1..6 -- Conv1d without bias, GELU):
    audio (B, N) fp32 -> extract_features (B, frames, 512) fp32.  `state` = the
    feature_extractor's state dict.  Pinned by tests/golden/g11_w2v2_features.npz,
    the output of the HF module itself (oracle/make_golden_w2v2.py)."""
    x = audio[:, None].to(torch.float)
    for layer, stride in enumerate(W2V2_STRIDES):
        x = torch.nn.functional.conv1d(x, state[f'conv_layers.{layer}.conv.weight'], None, stride=stride)
        if layer == 0:
            x = torch.nn.functional.group_norm(
                x, x.shape[1], state['conv_layers.0.layer_norm.weight'],
                state['conv_layers.0.layer_norm.bias'], 1e-5)
        x = torch.nn.functional.gelu(x)
    return x.transpose(1, 2)


def w2v2_body(state, features, valid, heads=12, eps=1e-5):
    """HF transformers ``Wav2Vec2Model.forward`` after the feature encoder, restated (the
    third-party body reference ppgs/preprocess/w2v2fb/core.py:66 runs; models/wav2vec2/
    modeling_wav2vec2.py): Wav2Vec2FeatureProjection (LayerNorm 512, Linear 512 -> hidden),
    Wav2Vec2Encoder with do_stable_layer_norm = False (rows outside the frame-level
    attention mask zeroed, Wav2Vec2PositionalConvEmbedding = weight-normed Conv1d k = 128,
    padding 64, 16 groups, last output dropped, GELU; residual; LayerNorm) and its
    Wav2Vec2EncoderLayers (self-attention with q scaled by d^-0.5 and padded keys masked,
    residual, LayerNorm, Linear-GELU-Linear, residual, LayerNorm).
    features (B, T, 512) fp32, valid: frames of each item inside the mask ->
    last_hidden_state (B, T, hidden) fp32.  `state` = the model's state dict.  Pinned by
    tests/golden/g12_w2v2_body.npz, the output of the HF modules themselves
    (oracle/make_golden_w2v2_body.py)."""
    F = torch.nn.functional
    x = features.to(torch.float)
    B, T, _ = x.shape
    mask = torch.arange(T)[None] < torch.as_tensor(valid).reshape(-1, 1)           # (B, T)
    x = F.layer_norm(x, (x.shape[-1],), state['feature_projection.layer_norm.weight'],
                     state['feature_projection.layer_norm.bias'], eps)
    x = F.linear(x, state['feature_projection.projection.weight'], state['feature_projection.projection.bias'])
    x = x * mask[..., None]
    prefix = 'encoder.pos_conv_embed.conv.'
    if prefix + 'weight' in state:
        weight = state[prefix + 'weight']
    else:                                                                           # weight norm over dims (0, 1): one gain per tap
        g_key = prefix + ('parametrizations.weight.original0' if prefix + 'parametrizations.weight.original0' in state else 'weight_g')
        v_key = prefix + ('parametrizations.weight.original1' if prefix + 'parametrizations.weight.original1' in state else 'weight_v')
        v = state[v_key]
        weight = state[g_key] * v / v.norm(dim=(0, 1), keepdim=True)
    kernel = weight.shape[-1]
    groups = x.shape[-1] // weight.shape[1]
    pos = F.conv1d(x.transpose(1, 2), weight, state[prefix + 'bias'], padding=kernel // 2, groups=groups)
    if kernel % 2 == 0:
        pos = pos[..., :-1]
    x = x + F.gelu(pos).transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), state['encoder.layer_norm.weight'], state['encoder.layer_norm.bias'], eps)
    hidden = x.shape[-1]
    d = hidden // heads
    bias = torch.zeros(B, 1, 1, T)
    bias.masked_fill_(~mask[:, None, None, :], torch.finfo(torch.float).min)
    layer = 0
    while f'encoder.layers.{layer}.attention.q_proj.weight' in state:
        p = f'encoder.layers.{layer}.'
        def lin(name, v):
            return F.linear(v, state[p + name + '.weight'], state[p + name + '.bias'])
        q = (lin('attention.q_proj', x) * d ** -0.5).view(B, T, heads, d).transpose(1, 2)
        k = lin('attention.k_proj', x).view(B, T, heads, d).transpose(1, 2)
        v = lin('attention.v_proj', x).view(B, T, heads, d).transpose(1, 2)
        weights = torch.softmax(q @ k.transpose(-1, -2) + bias, dim=-1)
        attn = (weights @ v).transpose(1, 2).reshape(B, T, hidden)
        x = x + lin('attention.out_proj', attn)
        x = F.layer_norm(x, (hidden,), state[p + 'layer_norm.weight'], state[p + 'layer_norm.bias'], eps)
        x = x + lin('feed_forward.output_dense', F.gelu(lin('feed_forward.intermediate_dense', x)))
        x = F.layer_norm(x, (hidden,), state[p + 'final_layer_norm.weight'], state[p + 'final_layer_norm.bias'], eps)
        layer += 1
    return x

