"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES.

Runs only in the build container (needs /root/reference; it does not exist on
the GPU box, and nothing in tests/, smoke() or bench.py imports this file).
The reference cannot be imported as a package here (yapecs, torchaudio,
torchutil, librosa, pypar are absent), so -- SURVEY.md 8(c) recipe -- a stub
``ppgs`` namespace is registered, the config constants are executed into it,
and the hot-path files are loaded unmodified from where they lie:

    ppgs/config/defaults.py, ppgs/config/static.py   constants
    ppgs/model/transformer.py                        Transformer (the network)
    ppgs/preprocess/spectrogram.py, mel.py           frontend
    ppgs/data/sampler.py                             Sampler.batch (packing)

``librosa.filters.mel`` (absent) is served by
``transformers.audio_utils.mel_filter_bank`` (HF's librosa-compatible
implementation) so that the reference's mel.py runs unmodified; the oracle's
own restatement of the filterbank is checked against it too.

Each fixture stores inputs (or the seed that regenerates them), the reference
outputs, the torch version and the autocast flag (always OFF: fp32 parity
definition).

Usage:  python oracle/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

from ppgs_amd import weights as W      # seeded synthetic checkpoints  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    sys.modules[name] = module
    spec.loader.exec_module(module)
    return module


def import_reference():
    """Stub-namespace import of the reference hot-path modules."""
    ppgs = types.ModuleType('ppgs')
    ppgs.__path__ = [os.path.join(REF, 'ppgs')]
    sys.modules['ppgs'] = ppgs
    for cfg in ('defaults', 'static'):
        module = _load(
            f'ppgs.config.{cfg}', os.path.join(REF, 'ppgs', 'config', f'{cfg}.py'))
        for key in dir(module):
            if key.isupper():
                setattr(ppgs, key, getattr(module, key))

    # third-party stub: librosa.filters.mel served by HF's implementation
    from transformers.audio_utils import mel_filter_bank
    librosa = types.ModuleType('librosa')
    librosa.filters = types.ModuleType('librosa.filters')

    def mel(sr, n_fft, n_mels):
        return mel_filter_bank(
            n_fft // 2 + 1, n_mels, 0.0, sr / 2.0, sr,
            norm='slaney', mel_scale='slaney').T.astype(np.float32)
    librosa.filters.mel = mel
    sys.modules['librosa'] = librosa
    sys.modules['librosa.filters'] = librosa.filters

    ppgs.preprocess = types.ModuleType('ppgs.preprocess')
    sys.modules['ppgs.preprocess'] = ppgs.preprocess
    ppgs.preprocess.spectrogram = _load(
        'ppgs.preprocess.spectrogram',
        os.path.join(REF, 'ppgs', 'preprocess', 'spectrogram.py'))
    ppgs.preprocess.mel = _load(
        'ppgs.preprocess.mel', os.path.join(REF, 'ppgs', 'preprocess', 'mel.py'))
    ppgs.model = types.ModuleType('ppgs.model')
    sys.modules['ppgs.model'] = ppgs.model
    transformer = _load(
        'ppgs.model.transformer',
        os.path.join(REF, 'ppgs', 'model', 'transformer.py'))
    ppgs.model.Transformer = transformer.Transformer
    ppgs.data = types.ModuleType('ppgs.data')
    sys.modules['ppgs.data'] = ppgs.data
    ppgs.data.sampler = _load(
        'ppgs.data.sampler', os.path.join(REF, 'ppgs', 'data', 'sampler.py'))
    return ppgs


def reference_model(ppgs, state, **kwargs):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ppgs.model.Transformer(**kwargs)
    model.load_state_dict(state)
    return model.eval()


def reference_forward(model, features, lengths, softmax=True):
    """reference infer() tail (core.py:586-596) with autocast OFF."""
    with torch.inference_mode():
        logits = model(features.float(), lengths.clone())
        return torch.softmax(logits, dim=1) if softmax else logits


def randn(seed, *shape):
    generator = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=generator)


def save(name, **arrays):
    meta = dict(torch_version=torch.__version__, autocast=False)
    out = {}
    for key, value in arrays.items():
        if isinstance(value, torch.Tensor):
            value = value.detach().cpu().numpy()
        out[key] = value
    np.savez_compressed(os.path.join(OUT, name + '.npz'), meta=str(meta), **out)
    size = os.path.getsize(os.path.join(OUT, name + '.npz'))
    print(f'{name}: {size / 1024:.1f} KiB')


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ppgs = import_reference()
    spectrogram = ppgs.preprocess.spectrogram
    mel = ppgs.preprocess.mel

    # G0: mel filterbank as the reference's mel.py sees it
    basis = sys.modules['librosa'].filters.mel(sr=16000, n_fft=1024, n_mels=80)
    save('g0_mel_basis', basis=basis)

    # G1: frontend. (a) two equal-length rows, (b) zero-padded ragged batch
    # (batch-edge reflect semantics: collate zero-extends, then reflect-pad)
    audio = 0.1 * randn(11, 2, 1, 4000)
    spec16 = spectrogram.from_audios(audio, audio.shape[-1])
    mel16 = mel.from_audios(audio, audio.shape[-1])
    stft = torch.stft(
        torch.nn.functional.pad(audio, (432, 432), mode='reflect').squeeze(1),
        1024, hop_length=160, window=torch.hann_window(1024), center=False,
        onesided=True, return_complex=True)
    spec32 = torch.sqrt(torch.view_as_real(stft).pow(2).sum(-1) + 1e-6)
    ragged = 0.1 * randn(12, 2, 1, 4800)
    ragged[1, :, 3000:] = 0.
    save('g1_frontend',
         audio=audio, spec32=spec32, spec16=spec16, mel16=mel16,
         ragged_audio=ragged, ragged_lengths=np.array([4800, 3000]),
         ragged_mel16=mel.from_audios(ragged, torch.tensor([4800, 3000])))
    # silence + a very short input (N = 1600 -> 10 frames)
    quiet = torch.zeros(1, 1, 1600)
    save('g1_frontend_silence', audio=quiet,
         mel16=mel.from_audios(quiet, 1600))

    # Models: seeded + sharpened, non-causal & causal
    state = W.seeded_state_dict(seed=1234)
    sharp = W.seeded_state_dict(seed=4321, sharpen=2.0)
    model = reference_model(ppgs, state)
    model_sharp = reference_model(ppgs, sharp)
    model_causal = reference_model(ppgs, state, is_causal=True)

    # G2: single window, ragged lengths
    feats = randn(21, 3, 80, 160).half()
    lengths = torch.tensor([160, 100, 37])
    save('g2_single_window',
         features=feats, lengths=lengths,
         logits=reference_forward(model, feats, lengths, False),
         ppg=reference_forward(model, feats, lengths, True),
         ppg_sharp=reference_forward(model_sharp, feats, lengths, True),
         logits_causal=reference_forward(model_causal, feats, lengths, False),
         ppg_causal=reference_forward(model_causal, feats, lengths, True))

    # G3: chunked. (2,80,900) with an item exhausted after the first window,
    # (1,80,501) and (1,80,1201): stride/overlap/last-window rules
    feats = randn(31, 2, 80, 900).half()
    lengths = torch.tensor([900, 300])
    g3 = dict(features_a=feats, lengths_a=lengths,
              ppg_a=reference_forward(model, feats, lengths),
              ppg_a_sharp=reference_forward(model_sharp, feats, lengths))
    for tag, T, seed in (('b', 501, 32), ('c', 1201, 33)):
        f = randn(seed, 1, 80, T).half()
        l = torch.tensor([T])
        g3[f'features_{tag}'] = f
        g3[f'lengths_{tag}'] = l
        g3[f'ppg_{tag}'] = reference_forward(model, f, l)
    save('g3_chunked', **g3)

    # G4: halo rule. item of length 60 in T=100 and T=62 batches
    feats = randn(41, 2, 80, 100).half()
    lengths = torch.tensor([100, 60])
    out100 = reference_forward(model, feats, lengths)
    feats62 = feats[:, :, :62].clone()
    lengths62 = torch.tensor([62, 60])
    out62 = reference_forward(model, feats62, lengths62)
    alone = reference_forward(model, feats[1:, :, :60], torch.tensor([60]))
    save('g4_halo', features=feats, lengths=lengths, ppg_T100=out100,
         ppg_T62=out62, ppg_alone=alone)

    # G5: w2v2fb-shaped model (Cin 768, H 512)
    state5 = W.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
    model5 = reference_model(
        ppgs, state5, input_channels=768, hidden_channels=512)
    feats = randn(51, 2, 768, 120).half()
    lengths = torch.tensor([120, 77])
    save('g5_w2v2fb', features=feats, lengths=lengths,
         ppg=reference_forward(model5, feats, lengths))

    # G6: big-shape statistics only (32 x 1000, config C2), audio -> ppg
    audio = 0.1 * randn(1234, 32, 1, 160000)
    feats = mel.from_audios(audio, 160000)
    lengths = torch.full((32,), 1000, dtype=torch.long)
    ppg = reference_forward(model, feats, lengths)
    save('g6_c2_stats',
         ppg_mean=ppg.mean(dim=-1), ppg_max=ppg.amax(dim=-1),
         argmax_hist=torch.stack([
             torch.bincount(ppg[b].argmax(0), minlength=40) for b in range(32)]),
         mel_sum=feats.float().sum(dim=(1, 2)),
         ppg_item0_first64=ppg[0, :, :64], ppg_item31_last64=ppg[31, :, -64:])

    # G7: entry point, config C1: audio (1,1,16000) -> (1,40,100), fp32 oracle
    # route from_features(mel.from_audios(audio, n).float(), frames)
    audio = 0.1 * randn(71, 1, 1, 16000)
    feats = mel.from_audios(audio, 16000)
    save('g7_c1_entry', audio=audio, mel16=feats,
         ppg=reference_forward(model, feats, torch.tensor([100])))

    # G8: packing. reference Sampler.batch() over 200 lengths
    gen = torch.Generator().manual_seed(1234)
    lens = torch.randint(50, 3001, (200,), generator=gen).numpy()

    class FakeDataset:
        def buckets(self):
            indices = np.argsort(lens)
            return [np.stack((indices, np.sort(lens))).T]
    g8 = dict(lengths=lens)
    for tag, max_frames in (('32000', 32000), ('inf', float('inf'))):
        sampler = ppgs.data.sampler.Sampler(FakeDataset(), max_frames=max_frames)
        batches = sampler.batch()
        g8[f'batches_{tag}_flat'] = np.concatenate(
            [np.asarray(b) for b in batches])
        g8[f'batches_{tag}_sizes'] = np.asarray([len(b) for b in batches])
    save('g8_packing', **g8)


if __name__ == '__main__':
    main()
