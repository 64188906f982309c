"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
committed reference fixtures.  Run with `pytest -m gpu` on an MI355X.

Tolerances (stated here, per the north star):
* fp32 mode: posteriors within 1e-4 max-abs of the reference / oracle.
* 16-bit operand modes (fp32 accumulation, fp32 residual / LayerNorm / softmax).  Where the
  error comes from is measured stage by stage in tools/precision_attribution.py: it is spread
  evenly over the operand roundings (weights 1.4e-3, attention output 9e-4, x1/x2/h 7-8e-4
  each at bf16 on the seeded checkpoint), i.e. it is the format, not one fixable stage.
  - bf16: 4e-3 max-abs on the seeded checkpoint (measured 2.2e-3 .. 3.2e-3; the reference's
    OWN shipped arithmetic -- bf16 autocast, fixture g7_glue -- is 2.4e-3 from its fp32 result
    on these weights); on the sharpened checkpoint (posteriors up to 0.5+) 5e-2, where the
    reference's shipped arithmetic is 1.3e-2 (C1) / 1.9e-2 (C2) from its fp32 result, and
    per-frame argmax agreement >= 99.9 % (100 % on frames whose top-2 margin exceeds 0.02).
  - fp16 (same MFMA rate, 11-bit significand): 1e-3 seeded (measured 3e-4), 5e-3 sharpened
    (measured 2.8e-3), argmax agreement 100 % at C2 size.
* frontend: fp16 outputs bit-equal for >= 99.5 % of values, never more than
  1 fp16 ulp apart (a different FFT factorisation flips rare roundings).
"""
import numpy as np
import pytest
import torch

import ppgs_amd
from oracle import ppg_oracle as O
from ppgs_amd import engine as E
from ppgs_amd import weights as W

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4
BF16_TOL = 4e-3          # seeded checkpoint
BF16_SHARP_TOL = 5e-2    # sharpened checkpoint (see the module docstring)
FP16_TOL = 1e-3
FP16_SHARP_TOL = 5e-3
TOL = {'fp32': FP32_TOL, 'bf16': BF16_TOL, 'fp16': FP16_TOL}
SHARP_TOL = {'fp32': FP32_TOL, 'bf16': BF16_SHARP_TOL, 'fp16': FP16_SHARP_TOL}


def t(x):
    return torch.from_numpy(np.asarray(x))


def ulp_diff(a, b):
    a = a.view(np.int16).astype(np.int32)
    b = b.view(np.int16).astype(np.int32)
    return np.abs(a - b)


_engines = {}


def eng(seed=1234, precision='fp32', causal=False, sharpen=1.0, cin=80,
        hidden=256, layers=5):
    key = (seed, precision, causal, sharpen, cin, hidden, layers)
    if key not in _engines:
        state = W.seeded_state_dict(
            seed=seed, sharpen=sharpen, input_channels=cin,
            hidden_channels=hidden, num_layers=layers)
        _engines[key] = (E.Engine(state, 0, precision, causal), state)
    return _engines[key]


def run(engine, feats, lengths, softmax=True):
    out = engine.encode(t(feats).cuda(), lengths, softmax=softmax)
    torch.cuda.synchronize()
    return out.cpu().numpy()


# ---------------------------------------------------------------- frontend --

def test_frontend_matches_reference_fixture(golden):
    g = golden('g1_frontend')
    audio = t(g['audio']).cuda()
    spec, mel = E.frontend(audio, spectrogram=True, mel=True)
    spec, mel = spec.cpu().numpy(), mel.cpu().numpy()
    assert spec.shape == g['spec16'].shape and mel.shape == g['mel16'].shape
    d = ulp_diff(spec, g['spec16'])
    assert d.max() <= 1 and (d == 0).mean() >= 0.995
    d = ulp_diff(mel, g['mel16'])
    assert d.max() <= 1 and (d == 0).mean() >= 0.995
    # ragged batch: zero-extended rows, reflect at the batch edge
    mel = ppgs_amd.preprocess.mel.from_audios(t(g['ragged_audio']).cuda()).cpu().numpy()
    d = ulp_diff(mel, g['ragged_mel16'])
    assert d.max() <= 1 and (d == 0).mean() >= 0.995


def test_frontend_silence_and_short(golden):
    g = golden('g1_frontend_silence')
    mel = ppgs_amd.preprocess.mel.from_audios(t(g['audio']).cuda()).cpu().numpy()
    assert mel.shape == (1, 80, 10)
    assert ulp_diff(mel, g['mel16']).max() <= 1
    with pytest.raises(ValueError):
        ppgs_amd.preprocess.mel.from_audios(torch.zeros(1, 1, 400).cuda())


def test_frontend_odd_frames_vs_oracle():
    # frame count not a multiple of the 16-frame group / of the frame pair
    gen = torch.Generator().manual_seed(5)
    audio = 0.1 * torch.randn(3, 1, 160 * 37 + 59, generator=gen)
    ref = O.mel_from_audios(audio).numpy()
    mel = ppgs_amd.preprocess.mel.from_audios(audio.cuda()).cpu().numpy()
    assert mel.shape == ref.shape == (3, 80, 37)
    d = ulp_diff(mel, ref)
    assert d.max() <= 1 and (d == 0).mean() >= 0.995


def test_frontend_sample_staging_paths_vs_oracle():
    """The kernel stages a group's samples by DMA: 16 bytes per lane for groups inside their
    row when the rows are 16-byte aligned, 4 bytes per lane with one address per sample at the
    reflect-padded row ends and for unaligned rows.  Long rows (interior groups exist) in all
    three situations against the oracle."""
    gen = torch.Generator().manual_seed(6)
    frames = 16 * 9 + 5
    for extra in (0, 2):                       # row length a multiple of 4 samples or not
        audio = 0.1 * torch.randn(3, 1, 160 * frames + extra, generator=gen)
        ref = O.mel_from_audios(audio).numpy()
        mel = ppgs_amd.preprocess.mel.from_audios(audio.cuda()).cpu().numpy()
        assert mel.shape == ref.shape == (3, 80, frames)
        d = ulp_diff(mel, ref)
        assert d.max() <= 1 and (d == 0).mean() >= 0.995
        if extra == 0:
            # the same rows from a buffer that starts 4 bytes past a 16-byte boundary
            buf = torch.empty(audio.numel() + 1, device='cuda')
            view = buf[1:].view_as(audio)
            view.copy_(audio)
            assert view.data_ptr() % 16 == 4
            shifted = ppgs_amd.preprocess.mel.from_audios(view).cpu().numpy()
            assert np.array_equal(shifted, mel)


@pytest.mark.parametrize('rate', [48000, 44100, 22050, 8000, 16001])
def test_resample_matches_closed_form_fixture(golden, rate):
    """ppg_resample (device) against fixture G10: the published
    torchaudio.transforms.Resample filter evaluated in closed form in float64
    (oracle/make_golden_resample.py).  fp32 dot products of <= ~500 taps, hence
    2e-6 absolute on 0.1-scale audio; the oracle's kernel-bank restatement is
    checked against the same fixture in tests/test_oracle_golden.py."""
    g = golden('g10_resample')
    audio, ref = t(g[f'audio_{rate}']), g[f'out_{rate}']
    out = ppgs_amd.resample(audio.cuda(), rate)
    assert out.is_cuda and tuple(out.shape) == ref.shape
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6
    # host tensors (file loading) take the same kernel and come back on the host
    host = ppgs_amd.resample(audio, rate)
    assert not host.is_cuda and torch.equal(host, out.cpu())
    # a longer seeded signal against the oracle restatement
    gen = torch.Generator().manual_seed(rate)
    audio = 0.1 * torch.randn(3, 1, rate // 3 + 17, generator=gen)
    ref = O.resample(audio, rate)
    out = ppgs_amd.resample(audio.cuda(), rate)
    assert (out.cpu() - ref).abs().max() < 2e-6
    # through the API's preprocessing: 48 kHz audio -> 16 kHz mel frames
    if rate == 48000:
        mel = ppgs_amd.preprocess.from_audio(audio[:1].cuda(), sample_rate=rate, gpu=0)
        assert mel.shape == (1, 80, ref.shape[-1] // 160)
        assert ulp_diff(mel.cpu().numpy(), O.mel_from_audios(ref[:1]).numpy()).max() <= 1


def test_postops_match_reference_fixture(golden):
    """ppgs_amd.distance / sparsify / interpolate on the GPU against the outputs
    of the reference's own functions (fixture g9; the similarity matrix is an
    input stored in it).  Tolerances: fp32 sums of 40 terms, logs, one sqrt."""
    g = golden('g9_postops')
    x, y, sim = (t(g[k]).cuda() for k in ('x', 'y', 'similarity'))
    for normalize in (1, 0):
        for reduction in ('mean', 'sum', 'none'):
            out = ppgs_amd.distance(x, y, reduction=reduction, normalize=bool(normalize), similarity=sim)
            ref = np.asarray(g[f'distance_{normalize}_{reduction}'])
            assert np.allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-6), (normalize, reduction)
    assert float(ppgs_amd.distance(x, x, similarity=sim)) < 1e-3 and float(g['distance_same']) < 1e-3
    assert np.allclose(ppgs_amd.interpolate(x, y, 0.3).cpu().numpy(), g['interpolate_scalar'], atol=1e-7)
    assert np.allclose(ppgs_amd.interpolate(x, y, t(g['interp']).cuda()).cpu().numpy(), g['interpolate_vector'], atol=1e-7)
    batch = t(g['batch']).cuda()
    out = ppgs_amd.sparsify(batch)                                   # default: percentile 0.85
    assert out.shape == g['sparsify_percentile'].shape                 # (1, batch, 40, frames), like the reference
    assert np.allclose(out.cpu().numpy(), g['sparsify_percentile'], atol=1e-6)
    assert np.allclose(ppgs_amd.sparsify(batch, 'percentile', torch.tensor([0.5])).cpu().numpy(), g['sparsify_percentile_50'], atol=1e-6)
    assert np.allclose(ppgs_amd.sparsify(batch, 'constant', torch.tensor([0.1])).cpu().numpy(), g['sparsify_constant'], atol=1e-6)
    assert np.allclose(ppgs_amd.sparsify(batch[:1], 'topk', 3).cpu().numpy(), g['sparsify_topk3'], atol=1e-6)
    with pytest.raises(ValueError):
        ppgs_amd.sparsify(batch, 'median')


def test_grid_sample_matches_reference_fixture(golden):
    """ppgs_amd.edit.grid.sample on the GPU against the reference's own
    ppgs.edit.grid.sample (fixture g9): bit-exact -- two products and a sum per
    output value, no contraction.  'edges' holds exact-integer, past-the-end and
    negative indices; then a long random grid against the oracle."""
    from ppgs_amd.edit import grid
    g = golden('g9_postops')
    x, batch = t(g['x']).cuda(), t(g['batch']).cuda()
    for name in ('slow', 'fast', 'edges'):
        out = grid.sample(x, t(g[f'grid_{name}']))
        assert out.is_cuda and np.array_equal(out.cpu().numpy(), g[f'sample_{name}']), name
    assert np.array_equal(grid.sample(batch, grid.of_length(batch, 50)).cpu().numpy(), g['sample_batch'])
    half = grid.sample(x.half(), t(g['grid_slow']).cuda())
    assert half.dtype == torch.float32 and np.array_equal(half.cpu().numpy(), g['sample_half'])
    assert torch.equal(grid.constant(x, 0.7).cpu(), t(g['grid_slow'])) and grid.constant(x, 0.7).is_cuda
    assert grid.sample(x, torch.zeros(0)).shape == (40, 0)
    gen = torch.Generator().manual_seed(11)
    ppg = torch.softmax(torch.randn(3, 40, 2999, generator=gen), dim=1)
    index = torch.rand(7001, generator=gen) * 3010 - 5
    assert torch.equal(grid.sample(ppg.cuda(), index).cpu(), O.grid_sample(ppg, index))
    with pytest.raises(ValueError):
        grid.sample(x, torch.zeros(2, 2))


# ------------------------------------------------------------------- model --

@pytest.mark.parametrize('layers', [0, 1])
def test_stagewise_small_models(layers):
    """0 layers = gather + in-conv + out-conv + softmax; 1 layer adds
    QKV / attention / out-proj+LN / FFN+LN."""
    engine, state = eng(seed=7, layers=layers)
    gen = torch.Generator().manual_seed(3)
    feats = torch.randn(2, 80, 75, generator=gen).half()
    lengths = torch.tensor([75, 40])
    ref = O.from_features(state, feats, lengths, softmax=False).numpy()
    out = run(engine, feats, lengths, softmax=False)
    assert np.abs(out - ref).max() < FP32_TOL


def test_single_window_fp32(golden):
    g = golden('g2_single_window')
    engine, _ = eng()
    logits = run(engine, g['features'], g['lengths'], softmax=False)
    assert np.abs(logits - g['logits']).max() < 2e-4        # (measured <= 6e-5: the un-normalised path gets its own bound)
    assert np.all(logits[1, :, 100:] == 0) and np.all(logits[2, :, 37:] == 0)
    ppg = run(engine, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg']).max() < FP32_TOL
    assert np.allclose(ppg[2, :, 37:], 1 / 40)
    sharp, _ = eng(seed=4321, sharpen=2.0)
    ppg = run(sharp, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg_sharp']).max() < FP32_TOL
    assert g['ppg_sharp'].max() > 0.5        # the sharpened case is discriminating


def test_causal_fp32(golden):
    g = golden('g2_single_window')
    engine, _ = eng(causal=True)
    ppg = run(engine, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg_causal']).max() < FP32_TOL


def test_chunked_fp32(golden):
    g = golden('g3_chunked')
    engine, _ = eng()
    for tag in 'abc':
        ppg = run(engine, g[f'features_{tag}'], g[f'lengths_{tag}'])
        assert ppg.shape == g[f'ppg_{tag}'].shape
        assert np.abs(ppg - g[f'ppg_{tag}']).max() < FP32_TOL, tag
    sharp, _ = eng(seed=4321, sharpen=2.0)
    ppg = run(sharp, g['features_a'], g['lengths_a'])
    assert np.abs(ppg - g['ppg_a_sharp']).max() < FP32_TOL


def test_halo_rule_fp32(golden):
    g = golden('g4_halo')
    engine, _ = eng()
    out100 = run(engine, g['features'], g['lengths'])
    assert np.abs(out100 - g['ppg_T100']).max() < FP32_TOL
    out62 = run(engine, g['features'][:, :, :62], [62, 60])
    assert np.abs(out62 - g['ppg_T62']).max() < FP32_TOL


def test_w2v2fb_geometry_fp32(golden):
    g = golden('g5_w2v2fb')
    engine, _ = eng(seed=55, cin=768, hidden=512)
    ppg = run(engine, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg']).max() < FP32_TOL


def test_c1_entry_point(golden, tmp_path):
    """config C1 through the public API: from_audio (1,1,16000) -> (1,40,100),
    checkpoint read from a .pt in the reference's {'model': ...} wrapping."""
    g = golden('g7_c1_entry')
    path = tmp_path / 'seeded.pt'
    torch.save({'model': W.seeded_state_dict(seed=1234)}, path)
    old = ppgs_amd.core.PRECISION
    ppgs_amd.core.PRECISION = 'fp32'
    try:
        ppg = ppgs_amd.from_audio(
            t(g['audio']), 16000, checkpoint=str(path), gpu=0)
        assert ppg.shape == (1, 40, 100) and ppg.dtype == torch.float32
        assert ppg.is_cuda
        # end to end includes rare fp16 feature flips of the frontend: still inside the north star's 1e-4
        assert np.abs(ppg.cpu().numpy() - g['ppg']).max() < FP32_TOL
        feats = t(g['mel16']).cuda()
        ppg = ppgs_amd.from_features(
            feats, torch.tensor([100]), checkpoint=str(path), gpu=0)
        assert np.abs(ppg.cpu().numpy() - g['ppg']).max() < FP32_TOL
    finally:
        ppgs_amd.core.PRECISION = old


def test_gpu_none_returns_host_tensors_like_the_reference(golden):
    """BASELINE configs[0]'s literal call: ppgs.from_audio(audio_cpu, 16000) with gpu=None.  The reference computes on the
    CPU there and returns a CPU tensor (ppgs/core.py:22-69, :106); this engine computes on the current HIP device, and
    hands a caller who passed host tensors a host tensor back (`.numpy()` on the result works as it does on the
    reference's).  A tensor already on the device, or an explicit gpu=, keeps the result on the device."""
    g = golden('g7_c1_entry')
    old = ppgs_amd.core.PRECISION
    ppgs_amd.core.PRECISION = 'fp32'
    try:
        state = W.seeded_state_dict(seed=1234)
        ppg = ppgs_amd.from_audio(t(g['audio']), 16000, checkpoint=state)
        assert not ppg.is_cuda and ppg.shape == (1, 40, 100) and ppg.dtype == torch.float32
        assert np.abs(ppg.numpy() - g['ppg']).max() < FP32_TOL
        ppg = ppgs_amd.from_features(t(g['mel16']), torch.tensor([100]), checkpoint=state)
        assert not ppg.is_cuda
        assert np.abs(ppg.numpy() - g['ppg']).max() < FP32_TOL
        assert ppgs_amd.from_audio(t(g['audio']).cuda(), 16000, checkpoint=state).is_cuda
        assert ppgs_amd.from_audio(t(g['audio']), 16000, checkpoint=state, gpu=0).is_cuda
    finally:
        ppgs_amd.core.PRECISION = old


def test_random_ragged_batches_vs_oracle():
    """Seeded ragged batches incl. zero-length items, lengths < frames,
    T just above the chunk size -- against the oracle."""
    engine, state = eng()
    gen = torch.Generator().manual_seed(99)
    for T, lengths in ((16, [16]), (33, [33, 1, 17]), (500, [500, 499]),
                       (501, [501, 2, 0]), (850, [850, 401, 400, 399])):
        feats = torch.randn(len(lengths), 80, T, generator=gen).half()
        ref = O.from_features(state, feats, lengths).numpy()
        out = run(engine, feats, lengths)
        assert np.abs(out - ref).max() < FP32_TOL, (T, lengths)


@pytest.mark.parametrize('causal', [False, True])
@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_attention_tiles_of_two_widths(monkeypatch, precision, causal):
    """A batch whose windows span 30 .. 500 keys: the planner gives the short ones
    half-width query tiles (16 queries per wave) next to the full-width tiles of the
    long ones.  Same result as with one width everywhere and as the oracle."""
    lengths = [1000, 1000, 230, 129, 64, 30]
    gen = torch.Generator().manual_seed(31)
    feats = torch.randn(len(lengths), 80, 1000, generator=gen).half()
    state = W.seeded_state_dict(seed=1234)
    mixed = E.Engine(state, 0, precision, causal)
    monkeypatch.setenv('PPGS_AMD_ATTN_NARROW', '0')
    uniform = E.Engine(state, 0, precision, causal)
    monkeypatch.delenv('PPGS_AMD_ATTN_NARROW')
    a, b = run(mixed, feats, lengths), run(uniform, feats, lengths)
    tol = TOL[precision]
    assert np.isfinite(a).all() and np.abs(a - b).max() < tol
    ref = O.from_features(state, feats, lengths, is_causal=causal).numpy()
    assert np.abs(a - ref).max() < tol


@pytest.mark.parametrize('causal', [False, True])
@pytest.mark.parametrize('precision', ['fp32', 'fp16', 'bf16'])
def test_attention_rebase(monkeypatch, precision, causal):
    """The attention kernel keeps a per-query softmax shift (the first key tile's maximum) and
    re-bases it only when a probability passes 2^40 (2^10 with fp16 operands) -- which the seeded
    weights never reach.  PPGS_AMD_ATTN_REBASE=always lowers that ceiling to 1: every tile with a
    new maximum re-bases (rescale of l and O, the tile's p redone, the next tile's scores, already
    started from the old shift, corrected).  Same posteriors as the oracle either way.  Then, at the
    real ceiling, Q/K weights scaled so that the logits span hundreds: tiles past the first DO
    overflow the ceiling there."""
    lengths = [500, 333, 64, 17]
    gen = torch.Generator().manual_seed(77)
    feats = torch.randn(len(lengths), 80, 500, generator=gen).half()
    state = W.seeded_state_dict(seed=1234)
    engine = E.Engine(state, 0, precision, causal)
    ref = O.from_features(state, feats, lengths, is_causal=causal).numpy()
    tol = TOL[precision]
    lazy = run(engine, feats, lengths)
    monkeypatch.setenv('PPGS_AMD_ATTN_REBASE', 'always')
    eager = run(engine, feats, lengths)
    monkeypatch.delenv('PPGS_AMD_ATTN_REBASE')
    assert np.abs(lazy - ref).max() < tol and np.abs(eager - ref).max() < tol
    if precision == 'fp32':
        hot = {k: v.clone() for k, v in state.items()}
        for name in hot:
            if name.endswith('self_attn.in_proj_weight'):
                hot[name][:512] *= 12.0              # q and k rows: logits x 144
        ref = O.from_features(hot, feats, lengths, is_causal=causal).numpy()
        hot_engine = E.Engine(hot, 0, precision, causal)
        lazy = run(hot_engine, feats, lengths)
        monkeypatch.setenv('PPGS_AMD_ATTN_REBASE', 'always')
        eager = run(hot_engine, feats, lengths)
        monkeypatch.delenv('PPGS_AMD_ATTN_REBASE')
        # near one-hot attention over logits of magnitude 10^2..10^3: an fp32 score accumulator that starts
        # at -shift rounds differently for different shifts (3e-5 absolute on such a logit, i.e. 2e-5
        # relative on p), and five layers of near-argmax attention amplify that
        assert np.isfinite(lazy).all() and np.abs(lazy - eager).max() < 2e-3, np.abs(lazy - eager).max()
        # against the oracle only in distribution: with logits this large two keys tie to within fp32 rounding
        # in a few of the 20 000 softmaxes, and which one wins differs between any two implementations
        err = np.abs(lazy - ref)
        assert err.mean() < 1e-4 and np.quantile(err, 0.995) < 2e-3, (err.mean(), np.quantile(err, 0.995), err.max())


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_outconv_kernel_vs_linear_kernel(monkeypatch, precision):
    """The output convolution with LDS-resident weights (ppg_outconv.hip, 16-bit modes at hidden 256)
    against the generic k-tap kernel it replaces (PPGS_AMD_OUTCONV=0) and the oracle: ragged windows
    (edges inside 16-token blocks, an exhausted item, T just past a window), posteriors and logits."""
    lengths = [1130, 901, 500, 77, 0, 16]
    gen = torch.Generator().manual_seed(123)
    feats = torch.randn(len(lengths), 80, 1130, generator=gen).half()
    state = W.seeded_state_dict(seed=1234)
    fused = E.Engine(state, 0, precision)
    monkeypatch.setenv('PPGS_AMD_OUTCONV', '0')
    generic = E.Engine(state, 0, precision)
    monkeypatch.delenv('PPGS_AMD_OUTCONV')
    ref = O.from_features(state, feats, lengths).numpy()
    for softmax in (True, False):
        a, b = run(fused, feats, lengths, softmax=softmax), run(generic, feats, lengths, softmax=softmax)
        # same operands, another summation order of the 1280 products
        assert np.isfinite(a).all() and np.abs(a - b).max() < (2e-5 if softmax else 2e-4)
    assert np.abs(run(fused, feats, lengths) - ref).max() < TOL[precision]


def test_unfused_ffn_path_agrees(monkeypatch):
    monkeypatch.setenv('PPGS_AMD_FFN_UNFUSED', '1')
    state = W.seeded_state_dict(seed=1234)
    unfused = E.Engine(state, 0, 'fp32')
    monkeypatch.delenv('PPGS_AMD_FFN_UNFUSED')
    fused, _ = eng()
    gen = torch.Generator().manual_seed(8)
    feats = torch.randn(2, 80, 130, generator=gen).half()
    a = run(unfused, feats, [130, 90])
    b = run(fused, feats, [130, 90])
    assert np.abs(a - b).max() < 2e-5


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'fp16'])
def test_fused_layer_kernel_vs_oracle_and_unfused(monkeypatch, precision):
    """Batches above ~6 k token rows run each layer as attention + ONE kernel
    (out-proj + LN1 + FFN + LN2 + next layer's Q/K/V).  Ragged 700-frame items
    give windows of 500 + 300 frames: 19-block windows put the following window
    at an odd 16-token block (unpaired V^T stores), padding blocks and exhausted
    windows are present.  Checked against the CPU oracle and against the same
    engine with the fusions switched off."""
    state = W.seeded_state_dict(seed=99)
    gen = torch.Generator().manual_seed(17)
    batch, frames = 24, 700
    lengths = [700, 700, 655, 700, 513, 700, 402, 700, 700, 311, 700, 700,
               700, 99, 700, 700, 500, 700, 700, 641, 700, 17, 700, 700]
    feats = torch.randn(batch, 80, frames, generator=gen).half()
    _, info = E.plan_windows(batch, frames, lengths)
    assert info.tokens > 6144 and info.skipped_windows > 0
    fused = E.Engine(state, 0, precision)
    out = run(fused, feats, lengths)
    for workspace in fused._workspaces.values():       # nothing may depend on what the scratch held
        workspace.view(torch.int16).fill_(-1)
    assert np.array_equal(out, run(fused, feats, lengths))
    monkeypatch.setenv('PPGS_AMD_OP_FUSED', '0')
    monkeypatch.setenv('PPGS_AMD_QKV_FUSED', '0')
    pieces = E.Engine(state, 0, precision)
    monkeypatch.delenv('PPGS_AMD_OP_FUSED')
    monkeypatch.setenv('PPGS_AMD_QKV_FUSED', '0')
    no_tail = E.Engine(state, 0, precision)
    monkeypatch.delenv('PPGS_AMD_QKV_FUSED')
    # bf16, hidden 256: the default tiling here is the mixed one (160-token workgroups,
    # waves of 3/3/2/2 blocks trading one block's phase A through LDS); also run the
    # plain 48-tokens-per-wave tiling of the same kernel
    monkeypatch.setenv('PPGS_AMD_FFN_MIXED', '0')
    plain = E.Engine(state, 0, precision)
    monkeypatch.delenv('PPGS_AMD_FFN_MIXED')
    ref = O.from_features(state, feats, torch.tensor(lengths)).numpy()
    out_pieces = run(pieces, feats, lengths)
    out_no_tail = run(no_tail, feats, lengths)
    out_plain = run(plain, feats, lengths)
    # the token-split kernels everywhere (the 16-bit default at this size is the feature-split
    # layer32 kernel: 160-token workgroups, here with a partial last tile, windows that start
    # at odd 16-token blocks -> the unaligned V^T store path -- and padding blocks)
    monkeypatch.setenv('PPGS_AMD_LAYER32', '0')
    token_split = E.Engine(state, 0, precision)
    monkeypatch.delenv('PPGS_AMD_LAYER32')
    out_token_split = run(token_split, feats, lengths)
    # tolerances: fp32 = the parity bar; 16-bit = what test_16bit_modes allows
    tol = TOL[precision]
    if precision != 'fp32':
        # these weights (seed 99) on white-noise features are harsher than the fixtures: the bound is
        # what the FORMAT costs here -- the oracle with every MFMA operand rounded where the kernels
        # round it (tools/precision_attribution.py) -- with 60 % headroom for the summation order
        dtype = torch.bfloat16 if precision == 'bf16' else torch.float16
        emulated = O.from_features(state, feats, torch.tensor(lengths),
                                   quant=lambda stage, x: x.to(dtype).float()).numpy()
        tol = max(tol, 1.6 * float(np.abs(emulated - ref).max()))
    same = 2e-5 if precision == 'fp32' else tol       # 16-bit: operands are rounded at different points when kept in registers
    assert np.abs(out - ref).max() < tol
    assert np.abs(out_token_split - ref).max() < tol
    assert np.abs(out - out_pieces).max() < same
    assert np.abs(out - out_no_tail).max() < same
    assert np.abs(out - out_token_split).max() < same
    assert np.abs(out - out_plain).max() < same
    for b, n in enumerate(lengths):
        assert np.allclose(out[b, :, n:], 1 / 40)


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('case', ['touching_windows', 'chunked_fp32_features'])
def test_head_kernel_vs_three_launches(monkeypatch, precision, case):
    """Gather + input convolution + layer 0's Q/K/V in one kernel (ppg_head32.hip; batches of at
    least half a chip of 160-token tiles) against the gather / in-conv / QKV launches it replaces
    and against the oracle.  'touching_windows': 144 windows of exactly 160 rows (10 blocks of 16,
    no padding rows between them), so the convolution's taps at every window edge would reach a
    neighbour's live rows -- the per-lane tap mask -- with ragged valid lengths;
    'chunked_fp32_features': 1200-frame items (windows with replicate padding on the left,
    500 / 500 / 300 frames; 19-block windows leave half-written V^T groups to the kernel's
    housekeeping workgroups) given as fp32 features.  The scratch workspace is poisoned."""
    gen = torch.Generator().manual_seed(23)
    if case == 'touching_windows':
        frames = 160
        lengths = [160] * 8 + torch.randint(1, 161, (136,), generator=gen).tolist()
        feats = torch.randn(len(lengths), 80, frames, generator=gen).half()
    else:
        frames = 1200
        lengths = [1200, 1200, 1111, 1200, 640, 1200, 1200, 77] + [1200] * 9
        feats = torch.randn(len(lengths), 80, frames, generator=gen)
    _, info = E.plan_windows(len(lengths), frames, lengths)
    assert info.tokens >= 128 * 160
    state = W.seeded_state_dict(seed=1234)
    fused = E.Engine(state, 0, precision)
    monkeypatch.setenv('PPGS_AMD_HEAD32', '0')
    three = E.Engine(state, 0, precision)
    monkeypatch.delenv('PPGS_AMD_HEAD32')
    a, b = run(fused, feats, lengths), run(three, feats, lengths)
    assert np.isfinite(a).all() and not np.array_equal(a, b)       # (two different kernels ran)
    assert np.abs(a - b).max() < TOL[precision]
    for workspace in fused._workspaces.values():
        workspace.view(torch.int16).fill_(-1)                       # NaN in every format
    assert np.array_equal(a, run(fused, feats, lengths))
    ref = O.from_features(state, feats[:8].float(), torch.tensor(lengths[:8])).numpy()     # (items are independent)
    assert np.abs(a[:8] - ref).max() < TOL[precision]


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('shape', ['ragged', 'odd_blocks', 'layer32'])
def test_poisoned_workspace(precision, shape):
    """Nothing in the scratch workspace is read before it is written: the
    caller's buffer may hold anything (here: NaN bit patterns).  'ragged':
    windows of 40 and 300 frames; 'odd_blocks': every window has 21 16-token
    blocks (333 frames), so every window's last 32-column V^T group is half
    written by the projections -- round 1 cleared the wrong columns there in the
    16-bit modes (the permuted group's unwritten slots are not its linear tail);
    'layer32': a batch large enough for the feature-split kernel with windows at
    odd 16-token blocks and a partial last 160-token tile."""
    engine, state = eng(seed=5, precision=precision)
    gen = torch.Generator().manual_seed(12)
    if shape == 'ragged':
        frames, lengths = 700, [700, 40, 300, 513]
    elif shape == 'odd_blocks':
        frames, lengths = 333, [333, 333, 333, 333, 333, 333, 111]
    else:
        frames, lengths = 333, [333] * 29 + [111, 17, 333]
    feats = torch.randn(len(lengths), 80, frames, generator=gen).half()
    poison = torch.full((96 << 20,), float('nan'), device='cuda')      # what the allocator hands out next
    del poison
    fresh = E.Engine(state, 0, precision)                              # its workspace comes from the poisoned pool
    first = run(fresh, feats, lengths)
    assert np.isfinite(first).all()
    clean = run(engine, feats, lengths)
    for workspace in engine._workspaces.values():
        workspace.view(torch.int16).fill_(-1)          # 0xffff.. = NaN as bf16, fp16 and fp32
    dirty = run(engine, feats, lengths)
    assert np.isfinite(dirty).all()
    assert np.array_equal(clean, dirty) and np.array_equal(clean, first)
    ref = O.from_features(state, feats, torch.tensor(lengths)).numpy()
    assert np.abs(clean - ref).max() < TOL[precision]


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'fp16'])
def test_fused_layer_kernel_hidden_512(precision):
    """The w2v2fb-shaped network (768 -> 512) through the fused layer kernel
    (32-row weight tiles, 48 Q/K/V tiles, one 16-token block per wave)."""
    state = W.seeded_state_dict(seed=31, input_channels=768, hidden_channels=512)
    gen = torch.Generator().manual_seed(4)
    lengths = [600, 600, 471, 600, 333, 600]
    feats = torch.randn(6, 768, 600, generator=gen).half()
    _, info = E.plan_windows(6, 600, lengths)
    assert info.tokens > 2048                        # above the split-hidden regime
    out = run(E.Engine(state, 0, precision), feats, lengths)
    ref = O.from_features(state, feats, torch.tensor(lengths)).numpy()
    assert np.abs(out - ref).max() < TOL[precision]


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_16bit_modes(golden, precision):
    """The throughput modes against the reference fixtures: seeded AND sharpened
    checkpoint, single window / chunked / causal / hidden 512, and the entry-point
    fixture captured through the reference's genuine glue (g7_glue), next to the
    reference's own shipped (bf16 autocast) result on the same input."""
    tol, sharp_tol = TOL[precision], SHARP_TOL[precision]
    g = golden('g2_single_window')
    engine, _ = eng(precision=precision)
    sharp, _ = eng(seed=4321, sharpen=2.0, precision=precision)
    ppg = run(engine, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg']).max() < tol
    assert np.allclose(ppg.sum(1), 1, atol=1e-5)
    ppg = run(sharp, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg_sharp']).max() < sharp_tol
    assert (ppg.argmax(1) == g['ppg_sharp'].argmax(1)).mean() >= 0.99
    g3 = golden('g3_chunked')
    for tag in 'abc':
        ppg = run(engine, g3[f'features_{tag}'], g3[f'lengths_{tag}'])
        assert np.abs(ppg - g3[f'ppg_{tag}']).max() < tol, tag
    ppg = run(sharp, g3['features_a'], g3['lengths_a'])
    assert np.abs(ppg - g3['ppg_a_sharp']).max() < sharp_tol
    causal, _ = eng(precision=precision, causal=True)
    ppg = run(causal, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg_causal']).max() < tol
    wide, _ = eng(seed=55, cin=768, hidden=512, precision=precision)
    g5 = golden('g5_w2v2fb')
    ppg = run(wide, g5['features'], g5['lengths'])
    assert np.abs(ppg - g5['ppg']).max() < tol
    # entry point (C1) through the reference's own from_audio / from_features glue
    g7 = golden('g7_glue')
    mel = ppgs_amd.preprocess.mel.from_audios(t(g7['audio']).cuda())
    for model, tag, bound in ((engine, '', tol), (sharp, '_sharp', sharp_tol)):
        ppg = model.encode(mel, [100]).cpu().numpy()
        ours = np.abs(ppg - g7[f'ppg_fp32{tag}']).max()
        shipped = np.abs(g7[f'ppg_shipped{tag}'] - g7[f'ppg_fp32{tag}']).max()
        assert ours < bound, (tag, ours)
        # never worse than 1.6x what the reference's own shipped arithmetic loses on this input
        assert ours < 1.6 * shipped, (tag, ours, shipped)
    ppg = run(engine, g7['batch_features'], g7['batch_lengths'])
    assert np.abs(ppg - g7['batch_ppg_fp32']).max() < tol


def test_c2_sharpened_argmax_agreement(golden):
    """Config C2 size on the SHARPENED checkpoint (posteriors up to ~0.5): what a
    downstream user reads off a PPG is the per-frame phoneme ranking.  Fixture
    g6_sharp_stats holds the reference's fp32 argmax track, its top-1/top-2 margins,
    and how its own shipped bf16-autocast arithmetic fares on the same batch
    (99.978 % agreement, 1.9e-2 max-abs)."""
    g = golden('g6_sharp_stats')
    gen = torch.Generator().manual_seed(1234)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=gen)).cuda()
    mel = ppgs_amd.preprocess.mel.from_audios(audio)
    ref_arg = t(g['argmax'].astype(np.int64))
    margin = t(g['margin'].astype(np.float32))
    assert 0.999 < float(g['shipped_argmax_agreement']) < 1.0 and float(g['shipped_max_abs']) > 1e-2
    for precision, floor in (('fp32', 1.0), ('fp16', 0.9999), ('bf16', 0.999)):
        engine, _ = eng(seed=4321, sharpen=2.0, precision=precision)
        ppg = engine.encode(mel, [1000] * 32).cpu()
        agree = ppg.argmax(1) == ref_arg
        assert float(agree.float().mean()) >= floor, precision
        assert bool(agree[margin > 0.02].all()), precision           # only near-ties may flip
        tol = SHARP_TOL[precision]
        assert (ppg[0, :, :64] - t(g['ppg_item0_first64'])).abs().max() < tol
        assert (ppg[31, :, -64:] - t(g['ppg_item31_last64'])).abs().max() < tol
        assert (ppg.amax(-1) - t(g['ppg_max'])).abs().max() < tol
        assert (ppg.mean(-1) - t(g['ppg_mean'])).abs().max() < tol


# ------------------------------------------------- full size (config C2) ----

def test_c2_full_size_statistics(golden):
    """32 x 160000 samples -> 32 x 1000 frames, audio to posteriors, against
    per-item statistics captured from the reference (fixture G6), plus
    size-independent properties."""
    g = golden('g6_c2_stats')
    gen = torch.Generator().manual_seed(1234)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=gen)).cuda()
    mel = ppgs_amd.preprocess.mel.from_audios(audio)
    assert mel.shape == (32, 80, 1000)
    assert np.abs(mel.float().sum(dim=(1, 2)).cpu().numpy() - g['mel_sum']).max() < 2.0
    for precision, tol in (('fp32', FP32_TOL), ('bf16', BF16_TOL), ('fp16', FP16_TOL)):
        engine, _ = eng(precision=precision)
        ppg = engine.encode(mel, [1000] * 32)
        torch.cuda.synchronize()
        assert ppg.shape == (32, 40, 1000)
        assert torch.isfinite(ppg).all()
        assert (ppg.sum(1) - 1).abs().max() < 1e-5          # rows are distributions
        ppg = ppg.cpu()
        assert np.abs(ppg[0, :, :64].numpy() - g['ppg_item0_first64']).max() < tol
        assert np.abs(ppg[31, :, -64:].numpy() - g['ppg_item31_last64']).max() < tol
        assert np.abs(ppg.mean(-1).numpy() - g['ppg_mean']).max() < tol
        assert np.abs(ppg.amax(-1).numpy() - g['ppg_max']).max() < 2 * tol
        if precision == 'fp32':
            # every frame of every item: the per-item histogram of the winning phoneme, against the reference's own
            # (G6 `argmax_hist`).  ONE frame of the 32 000 has its two best posteriors closer than 2e-4 in the
            # reference's result (the fp32 mode is within 6e-5): at most one frame per item may change bins.
            hist = np.stack([np.bincount(ppg[b].argmax(0).numpy(), minlength=40) for b in range(32)])
            moved = np.abs(hist - g['argmax_hist']).sum(1)
            assert moved.max() <= 2 and moved.sum() <= 4, moved
    # batch-composition invariance: an item computed alone (full length, so
    # no halo difference) equals its row in the batch
    engine, _ = eng()
    alone = engine.encode(mel[5:6], [1000]).cpu()
    batch = engine.encode(mel, [1000] * 32).cpu()
    assert (alone[0] - batch[5]).abs().max() < 5e-6
    # permutation of batch rows permutes the output rows
    perm = torch.randperm(32, generator=gen)
    permuted = engine.encode(mel[perm.cuda()], [1000] * 32).cpu()
    # (fp32 rounding only: the FFN walks the hidden chunks in a per-workgroup
    # rotated order, so the summation order depends on the row's position)
    assert (permuted - batch[perm]).abs().max() < 5e-6


def test_files_to_files_roundtrip(tmp_path):
    """from_files_to_files writes torch.load-able (40, samples//160) fp32
    tensors; batched (num_workers>0) and serial paths agree where batch
    composition does not matter (equal lengths)."""
    from scipy.io import wavfile
    gen = torch.Generator().manual_seed(77)
    path = tmp_path / 'seeded.pt'
    torch.save(W.seeded_state_dict(seed=1234), path)
    files, outs_a, outs_b = [], [], []
    for i, n in enumerate((8000, 8000, 12345)):
        audio = (0.1 * torch.randn(n, generator=gen)).numpy()
        f = tmp_path / f'a{i}.wav'
        wavfile.write(f, 16000, audio.astype(np.float32))
        files.append(f)
        outs_a.append(tmp_path / f'a{i}-serial.pt')
        outs_b.append(tmp_path / f'a{i}-batched.pt')
    old = ppgs_amd.core.PRECISION
    ppgs_amd.core.PRECISION = 'fp32'
    try:
        ppgs_amd.from_files_to_files(files, outs_a, checkpoint=str(path), gpu=0)
        ppgs_amd.from_files_to_files(
            files[:2], outs_b[:2], checkpoint=str(path), num_workers=2, gpu=0,
            max_frames=1000)
    finally:
        ppgs_amd.core.PRECISION = old
    for f, n in zip(outs_a, (8000, 8000, 12345)):
        ppg = torch.load(f)
        assert ppg.shape == (40, n // 160) and ppg.dtype == torch.float32
    for a, b in zip(outs_a[:2], outs_b[:2]):
        assert (torch.load(a) - torch.load(b)).abs().max() < 1e-5


def test_file_pipeline_overlapped_batches_equal_one_batch_at_a_time(tmp_path):
    """from_files_to_files overlaps batch i + 1's H2D copy and frontend with batch i's encoder (two HIP streams) and
    batch i - 1's writes: every file's posteriors equal, bit for bit, those of the same batches run one at a time with
    a synchronize in between."""
    from scipy.io import wavfile
    rng = np.random.default_rng(5)
    path = tmp_path / 'seeded.pt'
    torch.save(W.seeded_state_dict(seed=1234), path)
    files, outs = [], []
    for i in range(200):
        samples = int(16000 * rng.uniform(3.0, 9.0))
        f = tmp_path / f'{i:03d}.wav'
        wavfile.write(f, 16000, (0.1 * rng.standard_normal(samples)).astype(np.float32))
        files.append(str(f))
        outs.append(str(tmp_path / f'{i:03d}.pt'))
    old = ppgs_amd.core.PRECISION
    ppgs_amd.core.PRECISION = 'bf16'
    try:
        ppgs_amd.from_files_to_files(files, outs, checkpoint=str(path), num_workers=8, gpu=0, max_frames=32000)
        torch.cuda.synchronize()
        model = ppgs_amd.core.engine_for('mel', str(path), 0)
        checked = 0
        for audios, lengths, names in ppgs_amd.core.loader(files, num_workers=4, max_frames=32000, gpu=0):
            mel = ppgs_amd.preprocess.mel.from_audios(audios.cuda())
            frames = (lengths // 160).tolist()
            ref = model.encode(mel, frames)
            torch.cuda.synchronize()
            for row, name in enumerate(names):
                got = torch.load(outs[files.index(name)])
                assert got.shape == (40, frames[row])
                assert torch.equal(got, ref[row, :, :frames[row]].cpu()), name
                checked += 1
        assert checked == len(files)
    finally:
        ppgs_amd.core.PRECISION = old


# ----------------------------------------- configs C4 / C5 as parity cases ---

def test_c4_bucketed_ragged_utterances_vs_oracle():
    """configs[3] at test scale: ragged utterances, length-bucketed and packed
    under max_frames, run through the sharding entry point (world size 1);
    every utterance is checked against the oracle run on the same padded
    batch (batch composition sets the halo frames)."""
    from ppgs_amd import data, distributed
    gen = torch.Generator().manual_seed(1234)
    frames = torch.randint(50, 1300, (24,), generator=gen).tolist()
    audios = [0.1 * torch.randn(1, f * 160, generator=gen) for f in frames]
    state = W.seeded_state_dict(seed=1234)
    engine, _ = eng()

    def compute(padded, lengths):
        mel = ppgs_amd.preprocess.mel.from_audios(padded.cuda())
        return engine.encode(mel, lengths // 160).cpu()

    out = distributed.from_audios_sharded(audios, compute=compute, max_frames=4000)
    assert [o.shape for o in out] == [(40, f) for f in frames]
    batches = data.pack_batches(frames, 4000)
    assert all(len(b) * max(frames[i] for i in b) <= 4000 for b in batches)
    for batch in batches[:3] + batches[-2:]:
        padded, _ = data.collate([audios[i] for i in batch])
        ref = O.from_audio(state, padded)
        # oracle treats every row as full length; compare with its own masks
        mel = O.mel_from_audios(padded)
        ref = O.from_features(state, mel.float(), [frames[i] for i in batch]).numpy()
        for row, index in enumerate(batch):
            err = np.abs(out[index].numpy() - ref[row, :, :frames[index]]).max()
            assert err < FP32_TOL, (index, err)


def test_c5_causal_streaming_chunks_and_graph_replay():
    """configs[4]: causal model, 64 independent 160-frame chunks; the encode
    is captured into a HIP graph (no allocation / sync inside ppg_encode once
    the plan is cached) and replayed on new data."""
    engine, state = eng(causal=True)
    gen = torch.Generator().manual_seed(17)
    feats = torch.randn(64, 80, 160, generator=gen).half()
    lengths = [160] * 64
    ref = O.from_features(state, feats[:8], lengths[:8], is_causal=True).numpy()
    static_in = feats.cuda()
    eager = engine.encode(static_in, lengths)
    torch.cuda.synchronize()
    assert np.abs(eager[:8].cpu().numpy() - ref).max() < FP32_TOL
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out = engine.encode(static_in, lengths)
    fresh = torch.randn(64, 80, 160, generator=gen).half()
    static_in.copy_(fresh.cuda())
    graph.replay()
    torch.cuda.synchronize()
    ref = O.from_features(state, fresh[:8], lengths[:8], is_causal=True).numpy()
    assert np.abs(static_out[:8].cpu().numpy() - ref).max() < FP32_TOL
    again = engine.encode(static_in, lengths)
    assert (again - static_out).abs().max() == 0


@pytest.mark.parametrize('precision', ['fp32', 'fp16', 'bf16'])
def test_kv_cached_stream_equals_causal_forward(precision):
    """Streaming causal mode (ppg_stream_*): an utterance pushed in ragged chunks (1 .. 100
    frames, not multiples of the kernels' 16-row blocks) with the K / V^T and residual rows
    cached on the device yields, frame by frame, the causal forward of the whole utterance --
    oracle (reference Transformer.forward with is_causal, one window) and the engine's own
    one-shot encode.  Only frames < received - 4 may be emitted before the flush (two 5-tap
    convolutions look two frames ahead each)."""
    engine, state = eng(precision=precision, causal=True)
    gen = torch.Generator().manual_seed(41)
    total = 437
    feats = torch.randn(80, total, generator=gen).half()
    ref = O.from_features(state, feats[None].float(), torch.tensor([total]), is_causal=True).numpy()[0]
    one_shot = run(engine, feats[None], [total])[0]
    stream = engine.stream(500)
    assert stream.rows == 512
    pieces, received = [], 0
    for n in (16, 48, 7, 100, 1, 64, 33, 90, 2, 76):
        out = stream.push(feats[:, received:received + n].cuda())
        received += n
        pieces.append(out)
        assert sum(p.shape[1] for p in pieces) == max(received - 4, 0)
    assert received == total and stream.received == total
    pieces.append(stream.push(None, flush=True))
    torch.cuda.synchronize()
    streamed = torch.cat(pieces, dim=1).cpu().numpy()
    assert streamed.shape == (40, total) and np.isfinite(streamed).all()
    assert np.abs(streamed - ref).max() < TOL[precision]
    # same kernels on the same rows as the one-shot forward (token-split path at this size): the 16-bit modes
    # agree with it much closer than with the fp32 oracle
    assert np.abs(streamed - one_shot).max() < (FP32_TOL if precision == 'fp32' else 2e-3)
    with pytest.raises(ValueError):
        stream.push(feats[:, :1].cuda())               # flushed
    with pytest.raises(ValueError):
        eng(precision=precision)[0].stream(100)        # not a causal engine
    with pytest.raises((ValueError, E.PpgError)):
        engine.stream(501)                             # more than one window


def test_large_ragged_batch_and_legacy_mode():
    """Many windows (B=96 x T=2600 -> 7 windows per item, ~270k token rows),
    and legacy (unchunked) mode on a 1200-frame item: finite, normalised, and
    spot-checked against the oracle."""
    engine, state = eng(precision='bf16')
    gen = torch.Generator().manual_seed(3)
    B, T = 96, 2600
    lengths = torch.randint(1, T + 1, (B,), generator=gen).tolist()
    lengths[0] = T
    feats = torch.randn(B, 80, T, generator=gen).half()
    out = engine.encode(feats.cuda(), lengths)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and (out.sum(1) - 1).abs().max() < 1e-5
    rows = [0, 41, 95]
    ref = O.from_features(state, feats[rows], [lengths[i] for i in rows]).numpy()
    # the oracle batch has the same T, so window plans agree row by row
    assert np.abs(out[rows].cpu().numpy() - ref).max() < BF16_TOL
    fp32, _ = eng()
    f = torch.randn(1, 80, 1200, generator=gen).half()
    legacy = fp32.encode(f.cuda(), [1200], legacy_mode=True).cpu().numpy()
    ref = O.from_features(state, f, [1200], legacy_mode=True).numpy()
    assert np.abs(legacy - ref).max() < FP32_TOL
    chunked = fp32.encode(f.cuda(), [1200]).cpu().numpy()
    assert np.abs(legacy - chunked).max() > 1e-4          # the two modes really differ
    with pytest.raises(ValueError):
        fp32.encode(torch.zeros(1, 80, 5000).half().cuda(), [5000], legacy_mode=True)


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
def test_two_pipelines_equal_one(monkeypatch, precision):
    """A batch of >= 128 token rows per CU runs as two half-batches on two HIP streams of the engine (the default,
    ppg_engine_pipelines); PPGS_AMD_STREAMS=1 is one pipeline.  Windows are independent: a uniform batch gives the
    same bits in the 16-bit modes; in a ragged one the planner may give a short window the other attention tile width (it ranks the
    windows of a pipeline), which moves the result by rounding of the operand format only."""
    state = W.seeded_state_dict(seed=1234)
    gen = torch.Generator().manual_seed(21)
    feats = torch.randn(32, 80, 1000, generator=gen).half().cuda()
    double_engine = E.Engine(state, 0, precision)
    monkeypatch.setenv('PPGS_AMD_STREAMS', '1')
    single_engine = E.Engine(state, 0, precision)
    monkeypatch.delenv('PPGS_AMD_STREAMS')
    for lengths, exact in (([1000] * 32, True), ([1000] * 30 + [730, 129], False)):
        _, info = E.plan_windows(32, 1000, lengths, engine=double_engine)
        assert double_engine.pipelines(info.tokens) == 2 and double_engine.pipelines(10240) == 1
        assert single_engine.pipelines(info.tokens) == 1
        double = double_engine.encode(feats, lengths)
        single = single_engine.encode(feats, lengths)
        torch.cuda.synchronize()
        if exact and precision != 'fp32':
            assert torch.equal(single, double)
        else:
            # (fp32 mode = the token-split kernels: a workgroup walks the hidden chunks from an offset of its own
            # index, so the fp32 sums of a row depend on where its tile lands -- last-bit differences)
            assert (single - double).abs().max() < (5e-6 if precision == 'fp32' else TOL[precision])
        # ... and again on a side stream of the caller (the fork and the join are relative to the caller's stream)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            again = double_engine.encode(feats, lengths)
        side.synchronize()
        assert torch.equal(again, double)


def test_c3_w2v2fb_frontend_and_engine(monkeypatch):
    """configs[2]: w2v2fb representation.  The wav2vec2 model is the
    reference's third-party HF architecture (seeded random init offline); its
    feature encoder AND its transformer body run on the HIP engine (ppg_w2v2_*,
    ppg_w2v2_body_*): the latents match the same HF model run by PyTorch on the
    CPU, and the 768-channel / hidden-512 PPG network on top runs in the HIP
    engine and matches the oracle on the same latents."""
    monkeypatch.setenv('PPGS_AMD_W2V2_RANDOM_INIT', '7')
    import transformers
    from ppgs_amd.preprocess import w2v2fb
    w2v2fb.clear()
    gen = torch.Generator().manual_seed(2)
    audio = 0.1 * torch.randn(2, 1, 16000, generator=gen)
    lengths = torch.tensor([16000, 12000])
    audio[1, :, 12000:] = 0
    feats = w2v2fb.from_audios(audio, lengths, gpu=0)
    assert feats.shape == (2, 768, 100) and feats.dtype == torch.float16 and feats.is_cuda
    # the same third-party model on the CPU (what the reference would run)
    torch.manual_seed(7)
    cpu_model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config()).eval()
    with torch.no_grad():
        padded = torch.nn.functional.pad(audio, (40, 40)).squeeze(1)
        positions = torch.arange(16000 + 80) - 80
        mask = (positions[None] < lengths[:, None]).long()
        ref = cpu_model(padded, mask).last_hidden_state.transpose(1, 2)
        ref = torch.nn.functional.interpolate(ref, size=100, mode='nearest').half()
    assert (feats.cpu().float() - ref.float()).abs().max() < 2e-2
    state = W.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
    engine = E.Engine(state, 0, 'fp32')
    frames = lengths // 160
    ppg = engine.encode(feats, frames).cpu().numpy()
    oracle = O.from_features(state, feats.cpu(), frames).numpy()
    assert np.abs(ppg - oracle).max() < FP32_TOL
    w2v2fb.clear()


@pytest.mark.parametrize('poison', ['nan', '-inf', 'inf'])
def test_non_finite_audio_through_w2v2fb_stays_non_finite(monkeypatch, poison):
    """the erf-form GELU of round 4 swallowed non-finite inputs (fminf / fmaxf return their non-NaN operand:
    GELU(NaN) = GELU(-Inf) = -1e-9), and the wav2vec2 feature encoder's first GELU has no residual around it -- NaN
    audio came out as finite garbage.  torch's gelu gives NaN for NaN, -Inf and +Inf... the feature rows the bad
    sample reaches must be non-finite here too."""
    monkeypatch.setenv('PPGS_AMD_W2V2_RANDOM_INIT', '7')
    from ppgs_amd.preprocess import w2v2fb
    w2v2fb.clear()
    gen = torch.Generator().manual_seed(6)
    audio = 0.1 * torch.randn(2, 1, 16000, generator=gen)
    audio[1, 0, 8000] = float(poison)
    old = ppgs_amd.core.PRECISION
    try:
        for precision in ('fp32', 'fp16', 'bf16'):
            ppgs_amd.core.PRECISION = precision
            feats = w2v2fb.from_audios(audio, torch.tensor([16000, 16000]), gpu=0).float()
            assert not bool(torch.isfinite(feats[1]).all()), (precision, poison)
            # (the clean item stays finite in the fp32 mode.  In the 16-bit modes the body's attention reads key tiles
            # of 64 rows: the masked keys behind an item's last frame are the NEXT item's rows, p = 0 times NaN is NaN in
            # the matrix pipe -- a non-finite item takes its batch neighbours with it, and the file pipeline's
            # per-batch check then skips the batch with a warning instead of saving anything made from it)
            if precision == 'fp32':
                assert bool(torch.isfinite(feats[0]).all()), precision
    finally:
        ppgs_amd.core.PRECISION = old
    w2v2fb.clear()
    # the activation itself, through the body's GELU GEMM epilogue and the token-split one: covered by the above for the
    # encoder; the device function is one: gelu_erf / gelu_erf_pair (ppg_device.h)


def test_c3_w2v2fb_fp16x2_route(monkeypatch):
    """configs[2] with PRECISION = 'fp16x2': the wav2vec2 engines run on fp16 hi + lo operand pairs too (feature
    encoder layers 1-6, every projection of the body, its attention; the positional convolution on f32-input MFMAs).
    The latents stay within 1e-4 of the fp32 mode's BEFORE their rounding to fp16 (the representation is fp16: a
    1e-4 difference may move a value across a rounding boundary, one fp16 ulp), and the posteriors of the hidden-512
    PPG network within 1e-4 of the oracle on the same latents.  PPGS_AMD_W2V2_FP32=1 (the route until round 5: the
    wav2vec2 engines in fp32) gives the fp32 mode's latents bit for bit."""
    monkeypatch.setenv('PPGS_AMD_W2V2_RANDOM_INIT', '7')
    from ppgs_amd.preprocess import w2v2fb
    w2v2fb.clear()
    gen = torch.Generator().manual_seed(5)
    audio = 0.1 * torch.randn(2, 1, 16000, generator=gen)
    lengths = torch.tensor([16000, 11200])
    audio[1, :, 11200:] = 0
    old = ppgs_amd.core.PRECISION
    try:
        ppgs_amd.core.PRECISION = 'fp32'
        reference_feats = w2v2fb.from_audios(audio, lengths, gpu=0).clone()
        ppgs_amd.core.PRECISION = 'fp16x2'
        assert w2v2fb.w2v2_precision() == 'fp16x2'
        feats = w2v2fb.from_audios(audio, lengths, gpu=0).clone()
        monkeypatch.setenv('PPGS_AMD_W2V2_FP32', '1')
        assert w2v2fb.w2v2_precision() == 'fp32'
        assert torch.equal(w2v2fb.from_audios(audio, lengths, gpu=0), reference_feats)
        monkeypatch.delenv('PPGS_AMD_W2V2_FP32')
        # the wav2vec2 engines themselves, before the representation's fp16 rounding
        model = w2v2fb.model_for(torch.device('cuda', 0))
        padded = torch.nn.functional.pad(audio[:, 0], (40, 40)).cuda()
        mask = (torch.arange(padded.shape[1])[None] < (lengths + 80)[:, None]).to(torch.long).cuda()
        hidden = {}
        for precision in ('fp32', 'fp16x2'):
            ppgs_amd.core.PRECISION = precision
            hidden[precision] = w2v2fb.last_hidden_state(model, padded, mask).clone()
    finally:
        ppgs_amd.core.PRECISION = old
    valid = model._get_feat_extract_output_lengths(lengths + 80)
    for item, count in enumerate(valid.tolist()):
        assert (hidden['fp16x2'][item, :count] - hidden['fp32'][item, :count]).abs().max() < 1e-4
    assert feats.shape == (2, 768, 100) and feats.dtype == torch.float16
    # one fp16 ulp at the latents' magnitude (< 8: 2^-8 = 3.9e-3), on a handful of values
    differ = feats != reference_feats
    assert (feats.float() - reference_feats.float()).abs().max() <= 2.0 ** -7 and differ.float().mean() < 0.02
    state = W.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
    engine = E.Engine(state, 0, 'fp16x2')
    frames = lengths // 160
    ppg = engine.encode(feats, frames).cpu().numpy()
    oracle = O.from_features(state, feats.cpu(), frames).numpy()
    assert np.abs(ppg - oracle).max() < FP32_TOL
    w2v2fb.clear()


def test_graphed_encode_at_the_benchmark_size_two_pipelines():
    """Engine.graphed at 32 x 1000 frames: the capture holds BOTH pipelines of the batch (fork / join across two HIP
    streams inside ppg_encode); replays equal the eager launches bit for bit, also with new input in the static buffer."""
    engine, _ = eng(precision='bf16')
    lengths = [1000] * 32
    _, info = E.plan_windows(32, 1000, lengths, engine=engine)
    assert engine.pipelines(info.tokens) == 2
    gen = torch.Generator().manual_seed(41)
    run = engine.graphed(32, 1000, lengths)
    for _ in range(3):
        feats = torch.randn(32, 80, 1000, generator=gen).half().cuda()
        reference = engine.encode(feats, lengths).clone()
        out = run(feats)
        torch.cuda.synchronize()
        assert torch.equal(out, reference)


def test_graphed_encode_helper():
    engine, state = eng(precision='bf16')
    gen = torch.Generator().manual_seed(4)
    run = engine.graphed(4, 100)
    for _ in range(2):
        feats = torch.randn(4, 80, 100, generator=gen).half()
        out = run(feats.cuda()).clone()
        torch.cuda.synchronize()
        ref = engine.encode(feats.cuda(), [100] * 4)
        assert (out - ref).abs().max() == 0


def test_frontend_and_steps_beside_the_encoder_on_other_streams():
    """Callers on several HIP streams (the file pipeline runs batch i + 1's frontend beside batch i's encoder): every
    result equals the one-stream result.  Round 2's frontend (256-thread workgroups that shared a CU with the
    attention kernel's) got a frame pair wrong about once per 300 pairs in exactly this situation."""
    engine, state = eng(precision='bf16')
    gen = torch.Generator().manual_seed(1234)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=gen)).cuda()
    lengths = [1000] * 32
    spec_ref, mel_ref = E.frontend(audio, spectrogram=True, mel=True)
    ref = engine.encode(mel_ref, lengths)
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(12):
        with torch.cuda.stream(a):
            for _ in range(2):
                engine.encode(mel_ref, lengths)
        with torch.cuda.stream(b):
            outs = [E.frontend(audio, spectrogram=(rep % 2 == 0), mel=True) for _ in range(4)]
        torch.cuda.synchronize()
        for spec, mel in outs:
            assert torch.equal(mel, mel_ref)
            assert spec is None or torch.equal(spec, spec_ref)
    for rep in range(4):
        outs = []
        for i in range(16):
            with torch.cuda.stream((a, b)[i % 2]):
                outs.append(engine.encode(ppgs_amd.preprocess.mel.from_audios(audio), lengths))
        torch.cuda.synchronize()
        assert all(torch.equal(out, ref) for out in outs)


def test_graph_capture_spans_both_pipelines():
    """A batch that runs as two pipelines, captured: the fork to the engine's second stream and the join are
    part of the HIP graph, and a replay equals the eager call."""
    engine, state = eng(precision='bf16')
    _, info = E.plan_windows(32, 1000, [1000] * 32, engine=engine)
    assert engine.pipelines(info.tokens) == 2
    gen = torch.Generator().manual_seed(9)
    run = engine.graphed(32, 1000)
    for _ in range(2):
        feats = torch.randn(32, 80, 1000, generator=gen).half().cuda()
        out = run(feats).clone()
        torch.cuda.synchronize()
        assert torch.equal(out, engine.encode(feats, [1000] * 32))


def test_plan_cache_eviction_and_graph_replay():
    """More than 64 distinct batch shapes pass through one engine (every ragged
    batch of a file job is a new shape): evictions must not touch the plan a
    captured HIP graph refers to, uploads are asynchronous (no device-wide
    sync), and capturing a shape whose plan is not cached is refused."""
    engine, state = eng(seed=21, precision='fp16', causal=True)
    gen = torch.Generator().manual_seed(2)
    feats = torch.randn(8, 80, 96, generator=gen).half().cuda()
    replay = engine.graphed(8, 96)
    second = engine.graphed(8, 96)          # its own scratch: two graphs of one engine do not share one
    assert replay.scratch.data_ptr() != second.scratch.data_ptr()
    before = replay(feats).clone()
    ref = O.from_features(state, feats.cpu(), [96] * 8, is_causal=True).numpy()
    assert np.abs(before.cpu().numpy() - ref).max() < FP16_TOL
    for i in range(90):                      # 90 new shapes -> the 64-entry cache turns over
        frames = 40 + i
        x = torch.randn(2, 80, frames, generator=gen).half().cuda()
        out = engine.encode(x, [frames, 17 + (i % 5)])
        if i % 30 == 0:
            check = O.from_features(state, x.cpu(), [frames, 17 + (i % 5)], is_causal=True).numpy()
            assert np.abs(out.cpu().numpy() - check).max() < FP16_TOL
    after = replay(feats)
    torch.cuda.synchronize()
    assert torch.equal(before, after)
    assert torch.equal(second(feats), after)
    with pytest.raises((E.PpgError, ValueError, RuntimeError)):
        graph = torch.cuda.CUDAGraph()
        fresh = torch.zeros(3, 80, 77, dtype=torch.float16, device='cuda')
        scratch = torch.empty(engine.workspace_bytes(3, 77, [77, 77, 77]), dtype=torch.uint8, device='cuda')
        with torch.cuda.graph(graph):
            engine.encode(fresh, [77, 77, 77], workspace=scratch)


# ----------------------------------------- wav2vec2 feature encoder (f1) ----

def _seeded_w2v2(golden):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    from oracle import make_golden_w2v2 as M
    g = golden('g11_w2v2_features')
    model = M.seeded_model(int(g['seed']))
    if abs(M.weight_checksum(model.feature_extractor) - float(g['checksum'])) > 1e-6 * float(g['checksum']):
        pytest.skip('this torch / transformers build seeds the HF model differently from the fixture')
    return g, model


def test_w2v2_feature_encoder_vs_hf_fixture(golden):
    """ppg_w2v2_features (conv0 + GroupNorm-from-moments + GELU, then six strided convolutions as
    MFMA GEMMs) against the output of HF's own Wav2Vec2FeatureEncoder (fixture G11): fp32 mode and
    fp16x2 (hi + lo operand pairs) 1e-4 on activations of magnitude ~3; fp16 operands 2e-2; ragged zero-padded row included."""
    g, model = _seeded_w2v2(golden)
    state = model.feature_extractor.state_dict()
    audio = t(g['audio']).cuda()
    for precision, tol in (('fp32', 1e-4), ('fp16x2', 1e-4), ('fp16', 2e-2), ('bf16', 1.5e-1)):
        encoder = E.W2v2FeatureEncoder(state, 0, precision)
        assert encoder.frames(6000) == 18
        out = encoder(audio)
        torch.cuda.synchronize()
        assert out.shape == (3, 18, 512)
        assert np.abs(out.cpu().numpy() - g['features']).max() < tol, precision
        # workspace contents must not matter
        for workspace in encoder._workspaces.values():
            workspace.fill_(255)
        assert torch.equal(out, encoder(audio))
    with pytest.raises(ValueError):
        encoder(torch.zeros(1, 300).cuda())


def test_w2v2_body_vs_hf_fixture(golden):
    """ppg_w2v2_body_forward (feature projection, grouped positional convolution, 12 post-norm layers
    with attention at head dimension 64) against the output of HF's own modules (fixture G12) on the
    rows inside the frame-level mask: fp32 mode and fp16x2 (every projection and the attention on fp16 hi + lo
    operand pairs) 1e-4 on activations of magnitude ~4.5; fp16 operands 1e-2; bf16 6e-2.  Ragged valid lengths (70, 47, 9 of 70 frames); workspace contents must not matter."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    from oracle import make_golden_w2v2 as M
    from oracle import make_golden_w2v2_body as MB
    g = golden('g12_w2v2_body')
    model = M.seeded_model(int(g['seed']))
    if abs(MB.body_checksum(model) - float(g['checksum'])) > 1e-6 * float(g['checksum']):
        pytest.skip('this torch / transformers build seeds the HF model differently from the fixture')
    features, valid, ref = t(g['features']).cuda(), g['valid'].tolist(), g['last_hidden_state']
    for precision, tol in (('fp32', 1e-4), ('fp16x2', 1e-4), ('fp16', 1e-2), ('bf16', 6e-2)):
        body = E.W2v2Body(model, 0, precision)
        out = body(features, valid)
        torch.cuda.synchronize()
        assert out.shape == (3, 70, 768) and torch.isfinite(out).all()
        for item, frames in enumerate(valid):
            assert np.abs(out[item, :frames].cpu().numpy() - ref[item, :frames]).max() < tol, (precision, item)
        for workspace in body._workspaces.values():
            workspace.fill_(255)
        assert torch.equal(out, body(features, valid))
    with pytest.raises(ValueError):
        body(features, [71, 1, 1])
    with pytest.raises(ValueError):
        body(features[:, :, :100], valid)


def test_w2v2_body_shapes_vs_hf_modules():
    """The HIP body against the HF modules on the same GPU (2-layer model of the base width, seeded)
    at the sizes the fixture does not reach: 30 s of audio (1499 frames, 24 key tiles), single-frame
    items, a batch of one-frame items, ragged masks, row counts that are not whole tiles -- in the fp32 mode and in
    fp16x2 (both <= 1e-4)."""
    import transformers
    transformers.utils.logging.set_verbosity_error()
    torch.manual_seed(5)
    model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=2)).eval().cuda()
    bodies = {precision: E.W2v2Body(model, 0, precision) for precision in ('fp32', 'fp16x2')}
    gen = torch.Generator().manual_seed(3)
    for shape, valid in (((1, 1499, 512), [1499]), ((2, 33, 512), [33, 1]), ((5, 1, 512), [1] * 5),
                         ((3, 257, 512), [257, 200, 129]), ((9, 77, 512), [77, 1, 40, 77, 76, 33, 64, 65, 77])):
        x = torch.randn(*shape, generator=gen).cuda()
        frames = shape[1]
        mask = torch.arange(frames, device='cuda')[None] < torch.tensor(valid, device='cuda')[:, None]
        with torch.no_grad():
            hidden, _ = model.feature_projection(x)
            ref = model.encoder(hidden, attention_mask=mask).last_hidden_state
        for precision, body in bodies.items():
            out = body(x, valid)
            assert torch.isfinite(out).all()
            for item, count in enumerate(valid):
                assert (out[item, :count] - ref[item, :count]).abs().max() < 1e-4, (precision, shape, item)


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_w2v2_body_16bit_kernels_at_odd_shapes(monkeypatch, precision):
    """The 16-bit modes' own kernels of the body (ppg_gemm32.hip for every projection, ppg_posconv.hip) at the sizes
    the fixture does not reach -- row counts that are not whole tiles (9 x 77 frames: 864 rows), one-frame items, 1499
    frames, ragged masks, a two-pipeline batch: against the HF modules on the same GPU (the fixture test's tolerances)
    and against the same mode on the token-split kernels (PPGS_AMD_W2V2_GEMM32=0, _POSCONV=0: same operand roundings,
    another summation order; measured 6e-4 / 4.5e-3)."""
    import transformers
    transformers.utils.logging.set_verbosity_error()
    torch.manual_seed(5)
    model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=2)).eval().cuda()
    gen = torch.Generator().manual_seed(3)
    tol_ref, tol_old = {'fp16': (1e-2, 2e-3), 'bf16': (6e-2, 1.5e-2)}[precision]
    cases = (((1, 1499, 512), [1499]), ((2, 33, 512), [33, 1]), ((5, 1, 512), [1] * 5), ((3, 257, 512), [257, 200, 129]),
             ((9, 77, 512), [77, 1, 40, 77, 76, 33, 64, 65, 77]), ((1, 160, 512), [160]), ((16, 499, 512), [499] * 15 + [250]))
    new_body = E.W2v2Body(model, 0, precision)
    monkeypatch.setenv('PPGS_AMD_W2V2_GEMM32', '0')
    monkeypatch.setenv('PPGS_AMD_W2V2_POSCONV', '0')
    old_body = E.W2v2Body(model, 0, precision)
    for shape, valid in cases:
        x = torch.randn(*shape, generator=gen).cuda()
        mask = torch.arange(shape[1], device='cuda')[None] < torch.tensor(valid, device='cuda')[:, None]
        with torch.no_grad():
            hidden, _ = model.feature_projection(x)
            ref = model.encoder(hidden, attention_mask=mask).last_hidden_state
        new, old = new_body(x, valid).clone(), old_body(x, valid).clone()
        assert torch.isfinite(new).all()
        for item, count in enumerate(valid):
            assert (new[item, :count] - ref[item, :count]).abs().max() < tol_ref, (shape, item)
            assert (new[item, :count] - old[item, :count]).abs().max() < tol_old, (shape, item)


def test_w2v2_body_two_pipelines_equal_one(monkeypatch):
    """A batch of >= 8 items (>= 4096 rows) runs as two half-batches on two HIP streams of the body
    (PPGS_AMD_W2V2_STREAMS=1: one): the same bits, odd batch sizes and ragged masks included."""
    import transformers
    transformers.utils.logging.set_verbosity_error()
    torch.manual_seed(5)
    model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=2)).eval().cuda()
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(9, 499, 512, generator=gen).cuda()
    valid = [499, 400, 499, 1, 257, 499, 33, 480, 499]
    double = E.W2v2Body(model, 0, 'bf16')(x, valid)
    monkeypatch.setenv('PPGS_AMD_W2V2_STREAMS', '1')
    single = E.W2v2Body(model, 0, 'bf16')(x, valid)
    torch.cuda.synchronize()
    for item, count in enumerate(valid):
        assert torch.equal(single[item, :count], double[item, :count]), item


@pytest.mark.parametrize('total', [1130, 780, 501, 2470])
def test_long_stream_equals_chunked_causal_forward(total):
    """An utterance longer than a window, streamed (engine.long_stream: every frame goes to the one
    or two 500-row windows of the reference's chunk rule it belongs to, each a KV-cached stream)
    equals the reference's chunked causal forward (oracle), frame by frame.  1130 frames = three
    windows; 780 = two, with a third that starts inside the utterance but past its kept range;
    501 = the shortest chunked case; 2470 = seven windows (the live-window bound below is what
    keeps a long-running stream's device memory constant)."""
    engine, state = eng(causal=True)
    gen = torch.Generator().manual_seed(55)
    feats = torch.randn(80, total, generator=gen).half()
    ref = O.from_features(state, feats[None].float(), torch.tensor([total]), is_causal=True).numpy()[0]
    stream = engine.long_stream()
    pieces, received = [], 0
    sizes = [16, 48, 7, 100, 1, 64, 33, 90, 2, 76, 130, 49]
    index = 0
    while received < total:
        n = min(sizes[index % len(sizes)], total - received)
        pieces.append(stream.push(feats[:, received:received + n].cuda()))
        received += n
        index += 1
        assert sum(p.shape[1] for p in pieces) <= max(received - 4, 0)
        # a finished window is never fed (and so never re-created) again: at most two are ever alive
        assert len(stream._windows) <= 2
    pieces.append(stream.push(None, flush=True))
    out = torch.cat(pieces, dim=1).cpu().numpy()
    assert out.shape == (40, total)
    assert np.abs(out - ref).max() < FP32_TOL


@pytest.mark.parametrize('capacity,total,step', [(40, 40, 1), (17, 9, 2), (500, 500, 125), (64, 3, 3)])
def test_kv_cached_stream_edges(capacity, total, step):
    """Streams at the edges: one-frame pushes, fewer frames than the convolutions' look-ahead before
    the flush, a full 500-frame window, fp32 features."""
    engine, state = eng(causal=True)
    gen = torch.Generator().manual_seed(77)
    feats = torch.randn(80, total, generator=gen)
    ref = O.from_features(state, feats[None], torch.tensor([total]), is_causal=True).numpy()[0]
    stream = engine.stream(capacity, torch.float32)
    pieces = [stream.push(feats[:, i:i + step].cuda()) for i in range(0, total, step)]
    pieces.append(stream.push(None, flush=True))
    out = torch.cat(pieces, dim=1).cpu().numpy()
    assert out.shape == (40, total) and np.abs(out - ref).max() < FP32_TOL
    with pytest.raises(ValueError):
        engine.stream(capacity).push(torch.zeros(80, capacity + 1).cuda())


def test_w2v2fb_representation_native_vs_pytorch(golden, monkeypatch):
    """The w2v2fb representation end to end (reference ppgs/preprocess/w2v2fb/core.py:32-75: pad 40,
    sample mask, Wav2Vec2Model, nearest upsampling, fp16) with the HIP feature encoder AND the HIP
    transformer body against the same HF model run entirely by PyTorch-ROCm (also with only the
    body on PyTorch: PPGS_AMD_W2V2_BODY=0), and against the fixture's last_hidden_state."""
    g, model = _seeded_w2v2(golden)
    from ppgs_amd.preprocess import w2v2fb
    device = torch.device('cuda', 0)
    model = model.to(device)
    monkeypatch.setattr(w2v2fb, '_models', {str(device): model})
    monkeypatch.setattr(ppgs_amd.core, 'PRECISION', 'fp32')
    audio = t(g['audio'])[:, None].cuda()
    lengths = torch.tensor([6000, 4100, 6000])
    native = w2v2fb.from_audios(audio, lengths, gpu=0)
    monkeypatch.setenv('PPGS_AMD_W2V2_NATIVE', '0')
    stock = w2v2fb.from_audios(audio, lengths, gpu=0)
    assert native.shape == stock.shape == (3, 768, 37) and native.dtype == torch.float16
    assert (native.float() - stock.float()).abs().max() < 5e-3
    monkeypatch.delenv('PPGS_AMD_W2V2_NATIVE')
    monkeypatch.setenv('PPGS_AMD_W2V2_BODY', '0')
    hybrid = w2v2fb.from_audios(audio, lengths, gpu=0)
    monkeypatch.delenv('PPGS_AMD_W2V2_BODY')
    monkeypatch.setenv('PPGS_AMD_W2V2_NATIVE', '0')
    assert (hybrid.float() - stock.float()).abs().max() < 5e-3 and not torch.equal(hybrid, native)
    # unpadded, unmasked input through the model = the fixture's last_hidden_state
    monkeypatch.delenv('PPGS_AMD_W2V2_NATIVE')
    with torch.no_grad():
        extract = w2v2fb.feature_encoder_for(device, model)(t(g['audio']).cuda())
        projected, _ = model.feature_projection(extract)
        hidden = model.encoder(projected).last_hidden_state
    assert np.abs(hidden.cpu().numpy() - g['last_hidden_state']).max() < 2e-3


def test_w2v2_feature_encoder_c3_size(golden):
    """configs[2] size: 16 x 160080 samples (10 s + the reference's 40 + 40 pad) -> 16 x 500 frames,
    HIP feature encoder against HF's on the same GPU (PyTorch-ROCm fp32) on spot rows."""
    g, model = _seeded_w2v2(golden)
    gen = torch.Generator().manual_seed(3)
    audio = (0.1 * torch.randn(16, 160080, generator=gen)).cuda()
    encoder = E.W2v2FeatureEncoder(model.feature_extractor.state_dict(), 0, 'fp32')
    out = encoder(audio)
    assert out.shape == (16, 500, 512) and bool(torch.isfinite(out).all())
    extractor = model.feature_extractor.cuda()
    with torch.no_grad():
        ref = extractor(audio[[0, 7, 15]]).transpose(1, 2)
    assert (out[[0, 7, 15]] - ref).abs().max() < 2e-4
    fast = E.W2v2FeatureEncoder(model.feature_extractor.state_dict(), 0, 'fp16')(audio)
    assert (fast[[0, 7, 15]] - ref).abs().max() < 3e-2


def test_c3_full_size_body_and_ppg_network(monkeypatch):
    """configs[2] at its full size, 16 x 160000 samples -> (16, 768, 1000) -> (16, 40, 1000), all on the
    HIP engine in fp32 mode: the w2v2fb latents (feature encoder + 12-layer body + upsampling) against the
    same seeded HF model run by PyTorch-ROCm on spot items (fp16 latents within 2e-4 + one fp16 rounding,
    different in < 5 % of the values; the body alone is held to 1e-4 by the G12 fixture), and the hidden-512 PPG network on those
    latents against the CPU oracle on spot items, 1e-4."""
    monkeypatch.setenv('PPGS_AMD_W2V2_RANDOM_INIT', '7')
    from ppgs_amd.preprocess import w2v2fb
    w2v2fb.clear()
    old = ppgs_amd.core.PRECISION
    ppgs_amd.core.PRECISION = 'fp32'
    try:
        gen = torch.Generator().manual_seed(12)
        audio = 0.1 * torch.randn(16, 1, 160000, generator=gen)
        lengths = torch.full((16,), 160000)
        lengths[5], lengths[11] = 123456, 40000
        for row in (5, 11):
            audio[row, :, int(lengths[row]):] = 0
        feats = w2v2fb.from_audios(audio, lengths, gpu=0)
        assert feats.shape == (16, 768, 1000) and feats.dtype == torch.float16
        model = w2v2fb.model_for(torch.device('cuda', 0))
        spots = [0, 5, 11]
        with torch.no_grad():
            padded = torch.nn.functional.pad(audio[spots].cuda(), (40, 40)).squeeze(1)
            positions = torch.arange(160080, device='cuda') - 80
            mask = (positions[None] < lengths[spots].cuda()[:, None]).long()
            ref = model(padded, mask).last_hidden_state.transpose(1, 2)
            ref = torch.nn.functional.interpolate(ref, size=1000, mode='nearest').half()
        for row, item in enumerate(spots):
            valid = int(lengths[item]) // 160
            # fp16 latents of magnitude ~4: equal up to one fp16 rounding flip (HIP engine fp32 vs PyTorch-ROCm fp32)
            got, want = feats[item, :, :valid].cpu().numpy(), ref[row, :, :valid].cpu().numpy()
            diff = np.abs(got.astype(np.float32) - want.astype(np.float32))
            ulp = np.spacing(np.maximum(np.abs(got), np.abs(want))).astype(np.float32)
            # (the fp32 latents agree to ~1e-4 -- the bar G11 / G12 hold the two stages to; then one fp16 rounding)
            assert (diff <= 2e-4 + ulp).all() and (diff > 0).mean() < 0.05, item
        state = W.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
        engine = E.Engine(state, 0, 'fp32')
        frames = (lengths // 160).tolist()
        ppg = engine.encode(feats, frames)
        assert ppg.shape == (16, 40, 1000) and bool(torch.isfinite(ppg).all())
        oracle = O.from_features(state, feats[spots].cpu(), [frames[i] for i in spots]).numpy()
        for row, item in enumerate(spots):
            assert np.abs(ppg[item, :, :frames[item]].cpu().numpy() - oracle[row, :, :frames[item]]).max() < FP32_TOL, item
        # the 16-bit modes on the same latents
        for precision in ('fp16', 'bf16'):
            fast = E.Engine(state, 0, precision).encode(feats, frames)
            for row, item in enumerate(spots):
                assert np.abs(fast[item, :, :frames[item]].cpu().numpy() - oracle[row, :, :frames[item]]).max() < TOL[precision], (precision, item)
    finally:
        ppgs_amd.core.PRECISION = old
        w2v2fb.clear()


def test_c4_thousand_utterances_slice():
    """configs[3], a 1000-utterance slice of the benchmark corpus (frame counts randint(50, 3001), seed 1234,
    the first 1000 of the 10 000): packed under max_frames = 32000 with the row budget the benchmark uses,
    run through the sharded entry point; every output has its utterance's frame count, rows are distributions,
    and utterances spot-checked against the oracle on their own padded batch (first / middle / last batch:
    the longest, a middle and the shortest bucket) are within 1e-4 in fp32 mode."""
    from ppgs_amd import data, distributed
    gen = torch.Generator().manual_seed(1234)
    frames = torch.randint(50, 3001, (10000,), generator=gen).tolist()[:1000]
    agen = torch.Generator().manual_seed(99)
    audios = [0.1 * torch.randn(1, f * 160, generator=agen) for f in frames]
    state = W.seeded_state_dict(seed=1234)
    engine, _ = eng()

    def compute(padded, lengths):
        mel = ppgs_amd.preprocess.mel.from_audios(padded.cuda())
        return engine.encode(mel, lengths // 160).cpu()

    out = distributed.from_audios_sharded(audios, compute=compute, max_frames=32000)
    assert [o.shape for o in out] == [(40, f) for f in frames]
    total = torch.cat([o.sum(0) for o in out])
    assert (total - 1).abs().max() < 1e-5
    # the batches exactly as from_audios_sharded formed them (one rank: the LPT order, frame AND row budget)
    mine = distributed.shard_lpt([data.flops(f) for f in frames], 1)[0]
    packed = data.pack_batches([frames[i] for i in mine], 32000, max_rows=data.row_budget(32000, gpu=None))
    batches = [[mine[j] for j in batch] for batch in packed]
    assert all(len(b) * max(frames[i] for i in b) <= 32000 for b in batches)
    assert sorted(i for b in batches for i in b) == list(range(1000))
    for batch in (batches[0], batches[len(batches) // 2], batches[-1]):
        keep = batch[:2] + batch[-1:]
        padded, _ = data.collate([audios[i] for i in batch])
        mel = O.mel_from_audios(padded)
        rows = [batch.index(i) for i in keep]
        # (the oracle on the spot rows only, at the batch's padded length: batch composition sets the halo frames)
        ref = O.from_features(state, mel[rows].float(), [frames[i] for i in keep]).numpy()
        for row, index in enumerate(keep):
            err = np.abs(out[index].numpy() - ref[row, :, :frames[index]]).max()
            assert err < FP32_TOL, (index, err)


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_batched_kv_cached_streams_equal_causal_forward(precision, fused, monkeypatch):
    """Several utterances advanced together by ONE launch sequence per step (engine.batched_stream: row-mapped
    launches of the token-split kernels, one attention launch over all items' query tiles): every item equals
    the oracle's causal forward of its own utterance, with ragged, unaligned pushes per item, items that sit
    steps out, items that end at different steps and an item of one frame.  `fused`: every step's layers as ONE
    fused launch each (out-projection + LayerNorm-1 + FFN + LayerNorm-2 + the next layer's Q/K/V under the row map:
    PPGS_AMD_STREAM_FUSED=2 when the stream is created, for streams pushed in whole chunks) or as four launches (the
    default).  The form is fixed for the life of a stream, so a row recomputed by a later step gets the same bits."""
    monkeypatch.setenv('PPGS_AMD_STREAM_FUSED', '2' if fused else '0')
    engine, state = eng(precision=precision, causal=True)
    gen = torch.Generator().manual_seed(91)
    totals = [437, 160, 1, 500, 33, 275, 96]
    batch = len(totals)
    feats = [torch.randn(80, n, generator=gen).half() for n in totals]
    ref = [O.from_features(state, f[None].float(), torch.tensor([n]), is_causal=True).numpy()[0] for f, n in zip(feats, totals)]
    stream = engine.batched_stream(batch, 500)
    sent, pieces, done = [0] * batch, [[] for _ in range(batch)], [False] * batch
    sizes = [16, 48, 7, 100, 1, 64, 33, 90, 2, 76, 130, 49, 5]
    step = 0
    while not all(done):
        counts, flush = [], []
        for b in range(batch):
            n = 0 if done[b] or (step + b) % 5 == 4 else min(sizes[(step + 3 * b) % len(sizes)], totals[b] - sent[b])
            counts.append(n)
            flush.append(not done[b] and sent[b] + n == totals[b] and (n > 0 or sent[b] == totals[b]))
        nmax = max(counts)
        chunk = torch.zeros(batch, 80, nmax, dtype=torch.float16)
        for b in range(batch):
            chunk[b, :, :counts[b]] = feats[b][:, sent[b]:sent[b] + counts[b]]
        out = stream.push(chunk.cuda(), counts, flush)
        for b in range(batch):
            sent[b] += counts[b]
            pieces[b].append(out[b])
            done[b] = done[b] or flush[b]
            assert sum(p.shape[1] for p in pieces[b]) <= (sent[b] if done[b] else max(sent[b] - 4, 0))
        step += 1
        assert step < 400
    tol = FP32_TOL if precision == 'fp32' else 2e-3
    for b in range(batch):
        got = torch.cat(pieces[b], dim=1).cpu().numpy()
        assert got.shape == (40, totals[b])
        assert np.abs(got - ref[b]).max() < tol, b
    with pytest.raises(ValueError):
        stream.push(torch.zeros(batch, 80, 4).cuda(), [4] * batch)        # every item was flushed


@pytest.mark.parametrize('batch', [1, 3])
@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_stream_near_full_push_after_small_push(precision, batch):
    """with batch * rows % 64 == 32 (capacity 150 -> 160 rows, odd batch) a step that touches nearly
    all 16-row blocks gives the split-hidden FFN a partial-row stride of ceil(blocks / 4) * 64 > batch * rows; the
    partial buffer was sized for batch * rows and the last splits wrote into the K | Q / V^T caches of the layers
    behind it -- cached rows the step does not recompute.  A small push (rows 0..15 cached), then a near-full one."""
    engine, state = eng(precision=precision, causal=True)
    gen = torch.Generator().manual_seed(150)
    totals = [150, 150, 139][:batch]
    feats = [torch.randn(80, n, generator=gen).half() for n in totals]
    ref = [O.from_features(state, f[None].float(), torch.tensor([n]), is_causal=True).numpy()[0] for f, n in zip(feats, totals)]
    stream = engine.batched_stream(batch, 150)
    assert stream.rows == 160
    pieces = [[] for _ in range(batch)]
    sent = [0] * batch
    for size in (20, 130):
        counts = [min(size, totals[b] - sent[b]) for b in range(batch)]
        chunk = torch.zeros(batch, 80, max(counts), dtype=torch.float16)
        for b in range(batch):
            chunk[b, :, :counts[b]] = feats[b][:, sent[b]:sent[b] + counts[b]]
        out = stream.push(chunk.cuda(), counts, flush=(size == 130))
        for b in range(batch):
            sent[b] += counts[b]
            pieces[b].append(out[b])
    tol = FP32_TOL if precision == 'fp32' else 2e-3
    for b in range(batch):
        got = torch.cat(pieces[b], dim=1).cpu().numpy()
        assert got.shape == (40, totals[b])
        assert np.abs(got - ref[b]).max() < tol, (b, np.abs(got - ref[b]).max())


def test_fp16_overflow_sets_the_sticky_flag():
    """fp16 operands end at 65504.  A checkpoint whose FFN activations pass that (layer 2's linear1 scaled by 3e4)
    turns into NaN posteriors in the fp16 mode: the engine's sticky flag reports it (Engine.nonfinite /
    check_finite, which the file pipelines consult before saving), the bf16 mode (fp32's exponent range) stays
    finite on the same weights, and an ordinary checkpoint never sets the flag."""
    gen = torch.Generator().manual_seed(8)
    feats = torch.randn(2, 80, 120, generator=gen).half().cuda()
    engine, _ = eng(precision='fp16')
    engine.encode(feats, [120, 77])
    assert engine.nonfinite() is False
    state = W.seeded_state_dict(seed=1234)
    state = {k: v.clone() for k, v in state.items()}
    state['model.layers.2.linear1.weight'] *= 3e4
    wild = E.Engine(state, 0, 'fp16')
    out = wild.encode(feats, [120, 77])
    assert not bool(torch.isfinite(out[0]).all())
    assert wild.nonfinite(clear=False) is True
    with pytest.raises(E.PpgError):
        wild.check_finite()
    assert wild.nonfinite() is False                       # cleared by check_finite
    wide = E.Engine(state, 0, 'bf16')
    out = wide.encode(feats, [120, 77])
    assert bool(torch.isfinite(out).all()) and wide.nonfinite() is False


def test_fp16x2_mode_meets_the_fp32_bar(golden):
    """The compensated 16-bit mode (PPG_PRECISION_FP16X2: operands as fp16 hi + lo planes, three fp16 MFMAs per
    product, fp32 accumulation): posteriors within 1e-4 -- the bar of the fp32 mode -- of the reference fixtures on
    the seeded AND the sharpened checkpoint (where plain fp16 operands are 3e-3 off), single window, chunked,
    causal, through the reference's own glue (g7_glue, fp32 route), on ragged batches against the oracle, and at
    the benchmark size (statistics fixture G6 and spot utterances)."""
    engine, state = eng(precision='fp16x2')
    sharp, _ = eng(seed=4321, sharpen=2.0, precision='fp16x2')
    g = golden('g2_single_window')
    ppg = run(engine, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg']).max() < FP32_TOL
    assert np.allclose(ppg.sum(1), 1, atol=1e-5)
    ppg = run(sharp, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg_sharp']).max() < FP32_TOL
    g3 = golden('g3_chunked')
    for tag in 'abc':
        ppg = run(engine, g3[f'features_{tag}'], g3[f'lengths_{tag}'])
        assert np.abs(ppg - g3[f'ppg_{tag}']).max() < FP32_TOL, tag
    ppg = run(sharp, g3['features_a'], g3['lengths_a'])
    assert np.abs(ppg - g3['ppg_a_sharp']).max() < FP32_TOL
    causal, _ = eng(precision='fp16x2', causal=True)
    ppg = run(causal, g['features'], g['lengths'])
    assert np.abs(ppg - g['ppg_causal']).max() < FP32_TOL
    g7 = golden('g7_glue')
    mel = ppgs_amd.preprocess.mel.from_audios(t(g7['audio']).cuda())
    for model, tag in ((engine, ''), (sharp, '_sharp')):
        ppg = model.encode(mel, [100]).cpu().numpy()
        assert np.abs(ppg - g7[f'ppg_fp32{tag}']).max() < FP32_TOL, tag
    ppg = run(engine, g7['batch_features'], g7['batch_lengths'])
    assert np.abs(ppg - g7['batch_ppg_fp32']).max() < FP32_TOL
    # ragged batches (0-length items, lengths around the chunk rule's edges) against the oracle
    gen = torch.Generator().manual_seed(17)
    for T, lengths in ((501, [501, 0, 33]), (850, [850, 400, 17, 849]), (64, [64, 1])):
        feats = torch.randn(len(lengths), 80, T, generator=gen).half()
        ref = O.from_features(state, feats.float(), torch.tensor(lengths)).numpy()
        out = run(engine, feats.numpy(), lengths)
        assert np.abs(out - ref).max() < FP32_TOL, (T, lengths)
    # the benchmark size: statistics of the reference (G6) and the small-batch path (hidden splits)
    g6 = golden('g6_c2_stats')
    agen = torch.Generator().manual_seed(1234)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=agen)).cuda()
    mel = ppgs_amd.preprocess.mel.from_audios(audio)
    ppg = engine.encode(mel, [1000] * 32).cpu()
    assert bool(torch.isfinite(ppg).all())
    assert np.abs(ppg[0, :, :64].numpy() - g6['ppg_item0_first64']).max() < FP32_TOL
    assert np.abs(ppg[31, :, -64:].numpy() - g6['ppg_item31_last64']).max() < FP32_TOL
    assert np.abs(ppg.mean(-1).numpy() - g6['ppg_mean']).max() < FP32_TOL


def test_fp16x2_mode_at_hidden_512(golden):
    """fp16x2 at the w2v2fb geometry (768 input channels, hidden 512, two heads of 256): attention on 32-key tiles of
    32 KiB (a tile holds a whole [32 hi | 32 lo] group), the FFN as two GEMMs through [32 hi | 32 lo] rows.  The
    reference fixture G5 and ragged / chunked / causal batches against the oracle, all at the fp32 bar; the
    KV-cached stream is refused in this combination, and so are geometries outside the two the mode covers."""
    wide, state = eng(seed=55, cin=768, hidden=512, precision='fp16x2')
    g5 = golden('g5_w2v2fb')
    ppg = run(wide, g5['features'], g5['lengths'])
    assert np.abs(ppg - g5['ppg']).max() < FP32_TOL
    assert np.allclose(ppg.sum(1), 1, atol=1e-5)
    gen = torch.Generator().manual_seed(29)
    for T, lengths in ((501, [501, 0, 33]), (850, [850, 400, 17, 849]), (64, [64, 1]), (499, [499] * 16)):
        feats = torch.randn(len(lengths), 768, T, generator=gen).half()
        ref = O.from_features(state, feats.float(), torch.tensor(lengths)).numpy()
        out = run(wide, feats.numpy(), lengths)
        assert np.isfinite(out).all()
        assert np.abs(out - ref).max() < FP32_TOL, (T, lengths)
    # sharpened posteriors (up to 0.5+), where plain fp16 operands are 3e-3 off
    sharp, sharp_state = eng(seed=56, sharpen=2.0, cin=768, hidden=512, precision='fp16x2')
    feats = torch.randn(3, 768, 300, generator=gen).half()
    ref = O.from_features(sharp_state, feats.float(), torch.tensor([300, 150, 299])).numpy()
    assert np.abs(run(sharp, feats.numpy(), [300, 150, 299]) - ref).max() < FP32_TOL
    causal, causal_state = eng(seed=55, cin=768, hidden=512, precision='fp16x2', causal=True)
    feats = torch.randn(2, 768, 200, generator=gen).half()
    ref = O.from_features(causal_state, feats.float(), torch.tensor([200, 77]), is_causal=True).numpy()
    assert np.abs(run(causal, feats.numpy(), [200, 77]) - ref).max() < FP32_TOL
    with pytest.raises((ValueError, E.PpgError)):
        causal.stream(100)
    with pytest.raises((ValueError, E.PpgError)):       # hidden 512 as four heads of 128: not covered
        E.Engine(W.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512), 0, 'fp16x2', heads=4)


def test_fp16x2_feature_split_ffn_vs_token_split_and_oracle(monkeypatch):
    """fp16x2, batches that fill the chip: the FFN block runs on ppg_ffn32x2.hip (96-token workgroups of the
    feature-split machinery, hi + lo fragment images, three MFMAs per product) instead of the token-split
    ffn_kernel<PrecX2> + reduce pass (PPGS_AMD_FFN32X2=0).  Both forms against each other on a uniform and on a
    ragged batch (tile tails, windows of every length), seeded and sharpened checkpoints, and against the fp32
    oracle on spot utterances: the fp32 bar, 1e-4."""
    gen = torch.Generator().manual_seed(321)
    feats = torch.randn(32, 80, 1000, generator=gen).half()
    cases = [[1000] * 32, [1000, 997, 730, 501, 500, 499, 129, 33] * 4]
    for seed, sharpen in ((1234, 1.0), (4321, 2.0)):
        state = W.seeded_state_dict(seed=seed, sharpen=sharpen)
        fused = E.Engine(state, 0, 'fp16x2')            # out-proj + LN1 + FFN + LN2 + the next layer's Q/K/V in one launch per layer
        monkeypatch.setenv('PPGS_AMD_FFN32X2', '2')
        no_tail = E.Engine(state, 0, 'fp16x2')          # ... without the Q/K/V tail
        monkeypatch.setenv('PPGS_AMD_FFN32X2', '1')
        ffn_only = E.Engine(state, 0, 'fp16x2')         # the FFN block only
        monkeypatch.setenv('PPGS_AMD_FFN32X2', '0')
        split = E.Engine(state, 0, 'fp16x2')
        monkeypatch.delenv('PPGS_AMD_FFN32X2')
        for lengths in cases:
            a = fused.encode(feats.cuda(), lengths)
            b = split.encode(feats.cuda(), lengths)
            c = ffn_only.encode(feats.cuda(), lengths)
            d = no_tail.encode(feats.cuda(), lengths)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(a).all())
            assert float((a - b).abs().max()) < 2e-5, (seed, lengths[:3])
            assert float((c - b).abs().max()) < 2e-5, (seed, lengths[:3])
            assert float((d - b).abs().max()) < 2e-5, (seed, lengths[:3])
            picks = [1, 30]
            ref = O.from_features(state, feats[picks].float(), torch.tensor([lengths[i] for i in picks])).numpy()
            assert np.abs(a[picks].cpu().numpy() - ref).max() < FP32_TOL, (seed, lengths[:3])
