"""Native ingest / output stage (host only): WAV batch decode against the
scipy-based loader, .pt writer against torch.load."""
import numpy as np
import pytest
import torch
from scipy.io import wavfile

from ppgs_amd import engine as E
from ppgs_amd import load


@pytest.fixture()
def wavs(tmp_path):
    rng = np.random.default_rng(0)
    x = (0.1 * rng.standard_normal(4321)).astype(np.float32)
    files = {
        'f32': x, 'f64': x.astype(np.float64),
        'i16': (x * 32768).astype(np.int16), 'i32': (x * 2 ** 31).astype(np.int32),
        'u8': ((x * 128) + 128).astype(np.uint8), 'stereo': np.stack([x, -x], 1),
    }
    paths = []
    for name, samples in files.items():
        path = tmp_path / f'{name}.wav'
        wavfile.write(path, 16000, samples)
        paths.append(path)
    return paths


def test_wav_batch_matches_python_loader(wavs):
    batch, lengths, rates = E.wav_read_batch(wavs, 5000, threads=3, pin_memory=False)
    assert batch.shape == (len(wavs), 1, 5000)
    assert lengths.tolist() == [4321] * len(wavs) and rates == [16000] * len(wavs)
    for row, path in zip(batch, wavs):
        reference = load.audio(path)[0]
        assert torch.equal(row[0, :4321], reference), path        # bit-identical decode
        assert row[0, 4321:].abs().max() == 0                      # collate zero padding
    assert E.wav_info(wavs[-1]) == (4321, 16000, 2)
    # truncation to max_samples
    batch, lengths, _ = E.wav_read_batch(wavs[:1], 1000, threads=1, pin_memory=False)
    assert torch.equal(batch[0, 0], load.audio(wavs[0])[0, :1000]) and lengths.tolist() == [4321]


def test_wav_errors(tmp_path):
    bad = tmp_path / 'bad.wav'
    bad.write_bytes(b'not a wave file at all')
    with pytest.raises(ValueError, match='RIFF'):
        E.wav_info(bad)
    with pytest.raises(ValueError, match='cannot open'):
        E.wav_read_batch([tmp_path / 'missing.wav'], 100, pin_memory=False)


def test_pt_writer_is_torch_loadable(tmp_path):
    gen = torch.Generator().manual_seed(1)
    tensor = torch.randn(6, 40, 257, generator=gen)
    lengths = [257, 256, 1, 0, 100, 17]
    paths = [tmp_path / f'{i}.pt' for i in range(6)]
    E.pt_write_batch(paths, tensor, lengths, threads=4)
    for path, row, length in zip(paths, tensor, lengths):
        loaded = torch.load(path, weights_only=True)
        assert loaded.dtype == torch.float32 and loaded.shape == (40, length)
        assert loaded.is_contiguous() and torch.equal(loaded, row[:, :length])
        # same content as the reference's save_masked (preprocess/core.py:219-221)
        torch.save(row[..., :length].clone(), tmp_path / 'ref.pt')
        assert torch.equal(torch.load(tmp_path / 'ref.pt'), loaded)
    with pytest.raises(ValueError):
        E.pt_write_batch(paths[:1], tensor[:1], [300])


def _loader_files(tmp_path, specs):
    rng = np.random.default_rng(3)
    files = []
    for i, (n, rate) in enumerate(specs):
        path = tmp_path / f'{i}.wav'
        wavfile.write(path, rate, (0.1 * rng.standard_normal(n)).astype(np.float32))
        files.append(path)
    return files


def _check_loader(files):
    import ppgs_amd
    batches = list(ppgs_amd.core.loader(files, num_workers=2, max_frames=1000))
    seen = {f for _, _, names in batches for f in names}
    assert seen == set(files)
    for padded, lengths, names in batches:
        assert padded.shape[0] == len(names) and padded.shape[2] == int(lengths.max())
        for row, length, name in zip(padded, lengths, names):
            reference = load.audio(name)[0]
            assert torch.allclose(row[0, :length], reference[:length])
            assert row[0, length:].abs().sum() == 0


def test_loader_batches_native(tmp_path):
    """The loader decodes 16 kHz files natively with the collate semantics; a
    file at another rate needs the resampler, which is the HIP kernel: without a
    device that must fail loudly, not fall back to a host filter."""
    import ppgs_amd
    if torch.cuda.is_available():
        pytest.skip('CPU-only check of the loader (pinned memory not needed)')
    _check_loader(_loader_files(tmp_path, ((3200, 16000), (1600, 16000), (2000, 16000))))
    other = _loader_files(tmp_path, ((2400, 8000),))
    with pytest.raises(ppgs_amd.engine.PpgError):
        list(ppgs_amd.core.loader(other, num_workers=1, max_frames=1000))


@pytest.mark.gpu
def test_loader_batches_native_and_resampled(tmp_path):
    """... and with a device the 8 kHz file goes through load.audio + ppg_resample."""
    _check_loader(_loader_files(tmp_path, ((3200, 16000), (1600, 16000), (2400, 8000))))


def _wav_bytes(rate=16000, channels=1, bits=16, block_align=None, data=b'', data_size=None, fmt_tag=1):
    import struct
    block_align = channels * bits // 8 if block_align is None else block_align
    fmt = struct.pack('<HHIIHH', fmt_tag, channels, rate, rate * block_align, block_align, bits)
    size = len(data) if data_size is None else data_size
    body = b'WAVE' + b'fmt ' + struct.pack('<I', len(fmt)) + fmt + b'data' + struct.pack('<I', size) + data
    return b'RIFF' + struct.pack('<I', len(body) & 0xffffffff) + body


def test_wav_header_is_not_trusted(tmp_path):
    """Crafted / damaged RIFF headers: inconsistent block_align is rejected (the decoder
    would read past its buffer), streaming and truncated data chunks decode what the
    file really holds, and an unreadable file is skipped by the loader, not fatal."""
    import ppgs_amd
    samples = (np.arange(1000) - 500).astype('<i2')
    good = tmp_path / 'good.wav'
    good.write_bytes(_wav_bytes(data=samples.tobytes()))
    assert E.wav_info(good)[:2] == (1000, 16000)
    # block_align smaller than a frame: 64-bit samples at a 1-byte stride
    evil = tmp_path / 'evil.wav'
    evil.write_bytes(_wav_bytes(bits=64, block_align=1, fmt_tag=3, data=b'\\0' * 64))
    with pytest.raises(ValueError):
        E.wav_info(evil)
    odd = tmp_path / 'odd_bits.wav'
    odd.write_bytes(_wav_bytes(bits=12, block_align=2, data=b'\\0' * 64))
    with pytest.raises(ValueError):
        E.wav_info(odd)
    # streaming writer: size field 0xFFFFFFFF (or 0) -> what follows in the file
    for tag, size in (('ffff', 0xFFFFFFFF), ('zero', 0)):
        stream = tmp_path / f'stream_{tag}.wav'
        stream.write_bytes(_wav_bytes(data=samples.tobytes(), data_size=size))
        assert E.wav_info(stream)[0] == 1000
    # truncated: the header promises 1000 samples, 300 are there
    cut = tmp_path / 'cut.wav'
    cut.write_bytes(_wav_bytes(data=samples[:300].tobytes(), data_size=2000))
    assert E.wav_info(cut)[0] == 300
    padded, lengths, _ = E.wav_read_batch([str(good), str(cut)], 1000, threads=2, pin_memory=False)
    assert lengths.tolist() == [1000, 300]
    assert torch.equal(padded[1, 0, :300], torch.from_numpy(samples[:300].astype(np.float32) / 32768))
    assert padded[1, 0, 300:].abs().sum() == 0
    # the loader skips what it cannot read and keeps going
    junk = tmp_path / 'junk.wav'
    junk.write_bytes(b'not audio at all')
    with pytest.warns(UserWarning):
        batches = list(ppgs_amd.core.loader([good, junk, evil, cut], num_workers=1, max_frames=1000))
    seen = sorted(str(f) for _, _, names in batches for f in names)
    assert seen == sorted([str(good), str(cut)])
    # the .pt writer replaces atomically and leaves no temporary behind
    out = tmp_path / 'x.pt'
    E.pt_write_batch([str(out)], torch.zeros(1, 40, 5), [5], 1)
    assert torch.load(out).shape == (40, 5) and not list(tmp_path.glob('*.tmp~'))
    with pytest.raises(ValueError):
        E.pt_write_batch([str(tmp_path / 'no_such_dir' / 'y.pt')], torch.zeros(1, 40, 5), [5], 1)


def test_loader_reference_packing_mode(tmp_path, golden):
    """PACKING_MODE 'reference': the file loader packs exactly like the reference
    Sampler (fixture G8 sizes at max_frames 32000; one batch at infinity)."""
    import ppgs_amd
    if torch.cuda.is_available():
        pytest.skip('CPU-only check of the loader')
    g = golden('g8_packing')
    lengths = g['lengths'][:40]
    files = []
    for i, frames in enumerate(lengths):
        path = tmp_path / f'{i}.wav'
        wavfile.write(path, 16000, np.zeros(int(frames) * 160, np.int16))
        files.append(path)
    from oracle import ppg_oracle as O
    ref = O.sampler_batches(lengths, 32000)
    loader = ppgs_amd.core.loader(files, num_workers=1, max_frames=32000, mode='reference')
    assert [len(b) for b in loader.batches] == [len(b) for b in ref]
    assert [list(b) for b in loader.batches] == [list(map(int, b)) for b in ref]
    everything = ppgs_amd.core.loader(files, num_workers=1, max_frames=float('inf'), mode='reference')
    assert len(everything.batches) == 1 and len(everything.batches[0]) == 40
