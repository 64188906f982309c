"""Public inference API: same names and argument meaning as the reference's
``ppgs/core.py`` (from_audio :22, from_features :72, from_file :131,
from_file_to_file :171, from_files_to_files :207, from_dataloader :280,
infer :551, resample :599), executed by the HIP engine.

Differences that a drop-in user should know (also in INTEGRATION.md):
* ``gpu=None`` computes on the current HIP device (the reference runs on the
  CPU there); this engine has no CPU path and raises if no GPU is visible.  What
  the CALLER sees is the reference's: host tensors in + ``gpu=None`` -> a host
  tensor out (reference core.py:106: device 'cpu'); a tensor that is already on
  a HIP device, or an explicit ``gpu=``, -> the result stays on that device.
* Outputs are fp32 posteriors, layout (batch, 40, frames).
* ``from_audio`` accepts batch > 1 (all rows full length); the reference
  raises there (core.py:60).
"""
import contextlib
import math
import os
import warnings
from concurrent.futures import ThreadPoolExecutor

import torch

from . import config, data, engine, load, preprocess

# Arithmetic of the encoder GEMMs.  The reference's shipped inference runs under
# autocast (ppgs/core.py:586: fp16 on a GPU, bf16 on the CPU).  Here:
#   'fp16'  fp16 MFMA operands, fp32 accumulation / residual / LayerNorm / softmax (default:
#           the MFMA rate of bf16, posteriors ~3e-4 from the fp32 reference; operands must
#           stay below 65504, which the reference's own GPU autocast assumes of its checkpoints)
#   'bf16'  bf16 operands (the benchmark configuration of BASELINE.json; ~2-3e-3, which is what
#           the reference's own bf16 autocast loses, fixture g7_glue)
#   'fp32'  f32-input MFMA, the parity mode (<= 1e-4)
#   'fp16x2' every operand as an fp16 hi + lo pair, three fp16 MFMAs per product: ~3e-6 from the fp32
#           reference at a third of the fp32 mode's time
CHECK_FINITE = os.environ.get('PPGS_AMD_CHECK_FINITE', '0') != '0'
PRECISION = os.environ.get('PPGS_AMD_PRECISION', 'fp16')

# Frame budget of one padded batch when the caller leaves max_frames at the
# reference default of infinity (which would put every file in one batch).
DEFAULT_BATCH_FRAMES = 262144

# How from_files_to_files packs files into padded batches: 'sorted' (default:
# length-homogeneous batches under the same B * maxlen <= max_frames rule) or
# 'reference' (the reference Sampler's seeded shuffle and greedy packing, bit
# for bit -- fixture G8 -- including ONE batch of everything when max_frames is
# infinite: the longest item of a batch sets every other item's halo frames, so
# this is the switch for reproducing the reference's per-file numbers exactly).
PACKING_MODE = os.environ.get('PPGS_AMD_PACKING', 'sorted')

_engines = {}


###############################################################################
# Device / engine selection
###############################################################################


def device_for(gpu=None, tensor=None):
    if not torch.cuda.is_available():
        raise engine.PpgError(
            'ppgs_amd: no HIP device visible; the engine has no CPU path')
    if gpu is not None:
        return torch.device('cuda', int(gpu))
    if tensor is not None and tensor.is_cuda:
        return tensor.device
    return torch.device('cuda', torch.cuda.current_device())


def engine_for(representation=None, checkpoint=None, gpu=None, precision=None,
               is_causal=None):
    """Cached engine per (representation, checkpoint, device, precision,
    causal) -- the model cache of reference ppgs.infer (core.py:565-583)."""
    device = device_for(gpu)
    precision = precision or PRECISION
    is_causal = config.IS_CAUSAL if is_causal is None else bool(is_causal)
    ckpt_key = id(checkpoint) if isinstance(checkpoint, dict) else str(checkpoint)
    key = (str(representation), ckpt_key, device.index, precision, is_causal)
    if key not in _engines:
        state = load.state_dict(checkpoint, representation)
        _engines[key] = (engine.Engine(
            state, device=device.index, precision=precision,
            is_causal=is_causal), checkpoint)
    return _engines[key][0]


def clear_cache():
    _engines.clear()


###############################################################################
# Application programming interface
###############################################################################


def from_audio(audio, sample_rate, representation=config.REPRESENTATION,
               checkpoint=None, gpu=None, legacy_mode=False):
    """Infer ppgs from audio (reference ppgs/core.py:22-69).

    audio (batch, 1, samples) -> (batch, 40, samples // 160)
    """
    to_host = gpu is None and not audio.is_cuda      # (the reference's gpu=None returns a CPU tensor)
    features = preprocess.from_audio(
        audio=audio, sample_rate=sample_rate, representation=representation,
        gpu=gpu)
    lengths = torch.full(
        (features.shape[0],), features.shape[-1], dtype=torch.long)
    result = from_features(
        features=features, lengths=lengths, representation=representation,
        checkpoint=checkpoint, gpu=gpu, legacy_mode=legacy_mode)
    return result.cpu() if to_host else result


def from_features(features, lengths, representation=config.REPRESENTATION,
                  checkpoint=None, gpu=None, softmax=True, legacy_mode=False):
    """Infer ppgs from input features (reference ppgs/core.py:72-128).

    features (batch, channels, frames), lengths (batch,) ->
    (batch, 40, frames) posteriors (logits when softmax=False)
    """
    to_host = gpu is None and not features.is_cuda
    device = device_for(gpu, features)
    result = infer(
        features=features.to(device), lengths=lengths,
        representation=representation, checkpoint=checkpoint,
        softmax=softmax, legacy_mode=legacy_mode)
    return result.cpu() if to_host else result


def from_file(file, representation=config.REPRESENTATION, checkpoint=None,
              gpu=None, legacy_mode=False):
    """Infer ppgs from an audio file -> (40, frames)
    (reference ppgs/core.py:131-168)."""
    audio = load.audio(file, gpu=gpu)
    return from_audio(
        audio=audio[None] if audio.dim() == 2 else audio,
        sample_rate=config.SAMPLE_RATE, representation=representation,
        checkpoint=checkpoint, gpu=gpu, legacy_mode=legacy_mode).squeeze(0)


def from_file_to_file(audio_file, output_file,
                      representation=config.REPRESENTATION, checkpoint=None,
                      gpu=None, legacy_mode=False):
    """Infer ppg from an audio file and save a torch tensor file
    (reference ppgs/core.py:171-204)."""
    result = from_file(
        file=audio_file, checkpoint=checkpoint, representation=representation,
        gpu=gpu, legacy_mode=legacy_mode)
    host = result.detach().cpu()
    engine_for(representation, checkpoint, result.device.index).check_finite()     # (raises instead of saving NaNs)
    torch.save(host, output_file)


def from_files_to_files(audio_files, output_files,
                        representation=config.REPRESENTATION, checkpoint=None,
                        num_workers=0, gpu=None,
                        max_frames=config.MAX_INFERENCE_FRAMES,
                        legacy_mode=False):
    """Infer ppgs from audio files and save to torch tensor files
    (reference ppgs/core.py:207-272).

    num_workers == 0: one file at a time (batch 1), as the reference.
    num_workers > 0: files are length-sorted and packed into padded batches
    with ``len(batch) * longest <= max_frames``; ``num_workers // 2`` threads
    decode audio and ``num_workers // 2`` threads write results.
    """
    if len(audio_files) != len(output_files):
        raise ValueError('audio_files and output_files must pair one-to-one')
    if num_workers == 0:
        for audio_file, output_file in zip(audio_files, output_files):
            from_file_to_file(
                audio_file, output_file, representation=representation,
                checkpoint=checkpoint, gpu=gpu, legacy_mode=legacy_mode)
        return
    dataloader = loader(
        audio_files, num_workers=max(num_workers // 2, 1),
        max_frames=max_frames, mode=PACKING_MODE, gpu=gpu)
    mapping = dict(zip(audio_files, output_files))
    from_dataloader(
        dataloader=dataloader, output_files=mapping,
        representation=representation, checkpoint=checkpoint,
        save_workers=num_workers // 2, gpu=gpu, legacy_mode=legacy_mode)


###############################################################################
# Batched file pipeline
###############################################################################


class loader:
    """Iterable of (audios (B,1,maxlen), lengths (B,), audio_files) batches;
    counterpart of reference ppgs.data.loader with
    features=['audio','length','audio_file'] (ppgs/data/loader.py:20-43)."""

    def __init__(self, audio_files, num_workers=1,
                 max_frames=config.MAX_INFERENCE_FRAMES, mode='sorted', gpu=None):
        self.files = list(audio_files)
        self.gpu = gpu                     # the device that resamples files that are not at 16 kHz
        frames, self.samples, self.rates = [], [], []
        readable = []
        for file in self.files:
            # one unreadable file must not abort the job: warn and skip it
            try:
                samples, rate = load.info(file)
            except (ValueError, OSError, engine.PpgError) as error:
                warnings.warn(f'Skipping {file}: {error}')
                continue
            readable.append(file)
            frames.append(data.frames_of(samples, rate))
            self.samples.append(samples)
            self.rates.append(rate)
        self.files = readable
        budget = max_frames
        keep = data.filter_lengths(frames, budget, self.files)
        self.files = [self.files[i] for i in keep]
        self.frames = [frames[i] for i in keep]
        self.samples = [self.samples[i] for i in keep]
        self.rates = [self.rates[i] for i in keep]
        if math.isinf(budget) and mode != 'reference':
            budget = max(DEFAULT_BATCH_FRAMES, max(self.frames, default=0))
        rows = data.row_budget(budget) if mode == 'sorted' else None
        self.batches = data.pack_batches(self.frames, budget, mode=mode, max_rows=rows)
        self.num_workers = max(int(num_workers), 1)
        self.dataset = self.files          # len(dataloader.dataset) as in the reference

    def __len__(self):
        return len(self.batches)

    def _load(self, indices):
        """One padded batch.  16 kHz WAV files are decoded by the native
        multi-threaded reader straight into a pinned (B, 1, maxlen) buffer;
        anything else goes through load.audio (resampling) + collate."""
        files = [self.files[i] for i in indices]
        if all(self.rates[i] == config.SAMPLE_RATE for i in indices):
            longest = max(self.samples[i] for i in indices)
            padded, lengths, _ = engine.wav_read_batch(
                files, longest, threads=self.num_workers)
            return padded, lengths, tuple(files)
        audios = [load.audio(file, gpu=self.gpu)[:1] for file in files]
        padded, lengths = data.collate(audios)
        return padded, lengths, tuple(files)

    def __iter__(self):
        # one batch decoded ahead of the consumer (the native reader and the
        # Python fallback both release the GIL while reading)
        with ThreadPoolExecutor(1) as pool:
            pending = None
            for batch in self.batches + [None]:
                upcoming = pool.submit(self._load, batch) if batch is not None else None
                if pending is not None:
                    yield pending.result()
                pending = upcoming


def from_dataloader(dataloader, output_files,
                    representation=config.REPRESENTATION, checkpoint=None,
                    save_workers=1, gpu=None, legacy_mode=False):
    """Infer ppgs from a dataloader yielding (audio, length, filename) batches
    (reference ppgs/core.py:280-391): frontend on the padded batch, forward,
    then each item truncated to length // 160 frames and saved.

    Batches are software-pipelined over two HIP streams: while batch i runs
    its kernels, batch i+1's audio goes host -> device from pinned memory and
    batch i-1's posteriors come back into pinned memory and are handed to the
    writer threads (the reference does the three steps back to back and
    blocks on ``result.cpu()``, core.py:363).
    """
    if representation not in ('mel', 'w2v2fb'):
        raise ValueError(
            f'from_dataloader supports the mel and w2v2fb representations, '
            f'got {representation!r}')
    device = device_for(gpu)
    frontend = getattr(preprocess, representation)
    pool = ThreadPoolExecutor(2) if save_workers > 0 else None
    pending = []

    class Slot:
        def __init__(self):
            self.stream = torch.cuda.Stream(device)
            self.done = torch.cuda.Event()
            self.job = None

    slots = [Slot(), Slot()]

    def retire(slot):
        """Wait for the slot's batch, hand its rows to the writers."""
        if slot.job is None:
            return
        host, filenames, frame_lengths, keep = slot.job
        slot.done.synchronize()
        slot.job = None
        # The check is on THIS batch's own posteriors (pinned host memory, complete): the engine's sticky flag is
        # engine-wide, and the other slot's batch is already in flight on the same engine -- its NaN would be
        # charged to this batch and cleared before its own retirement.  The flag is cleared here only so that
        # from_file_to_file / check_finite callers after the job do not inherit it.
        if not bool(torch.isfinite(host).all()):
            engine_for(representation, checkpoint, device.index).nonfinite(clear=True)
            warnings.warn(
                f'Skipping a batch of {len(filenames)} files ({filenames[0]} ...): non-finite posteriors '
                f'({PRECISION} operands out of range, or non-finite audio); PPGS_AMD_PRECISION=bf16 / fp32 '
                'has the range of fp32')
            return
        # the native writer emits torch.load-able (40, length) files, several
        # threads per batch, off the main thread
        if pool is not None:
            pending.append(pool.submit(
                engine.pt_write_batch, filenames, host, frame_lengths,
                max(save_workers, 1)))
        else:
            engine.pt_write_batch(filenames, host, frame_lengths, 1)
        # back-pressure on the save queue (reference core.py:364-365)
        while len(pending) > 8:
            pending.pop(0).result()

    try:
        for index, (audios, lengths, audio_files) in enumerate(dataloader):
            slot = slots[index % 2]
            retire(slot)                      # its buffers are free again
            frame_lengths = lengths // config.HOPSIZE
            filenames = [output_files[file] for file in audio_files]
            staged = audios if audios.is_pinned() else audios.pin_memory()
            with torch.cuda.stream(slot.stream):
                on_device = staged.to(device, non_blocking=True)
                features = frontend.from_audios(on_device, lengths, gpu=device.index)
                result = from_features(
                    features=features, lengths=frame_lengths,
                    representation=representation, checkpoint=checkpoint,
                    gpu=device.index, legacy_mode=legacy_mode)
                host = torch.empty(
                    result.shape, dtype=result.dtype, pin_memory=True)
                host.copy_(result, non_blocking=True)
                slot.done.record(slot.stream)
            # keep the device tensors alive until the stream has used them
            slot.job = (host, filenames, frame_lengths.tolist(),
                        (staged, on_device, features, result))
        for slot in slots:
            retire(slot)
    finally:
        for slot in slots:
            if slot.job is not None:
                slot.done.synchronize()
        for future in pending:
            future.result()
        if pool is not None:
            pool.shutdown()


###############################################################################
# Utilities
###############################################################################


def infer(features, lengths, representation='mel', checkpoint=None,
          softmax=True, legacy_mode=False):
    """Perform model inference (reference ppgs/core.py:551-596)."""
    model = engine_for(representation, checkpoint, features.device.index)
    out = model.encode(
        features, lengths, softmax=softmax, legacy_mode=legacy_mode)
    if CHECK_FINITE:                   # (synchronises: off by default, the call is asynchronous like the reference's)
        model.check_finite()
    return out


def resample(audio, sample_rate, target_rate=config.SAMPLE_RATE, gpu=None):
    """Perform audio resampling (reference ppgs/core.py:599-608; `gpu` is this package's
    addition: which HIP device converts a HOST tensor -- default the current one).

    Identity at 16 kHz.  Otherwise torchaudio.transforms.Resample's default
    windowed-sinc polyphase filter (Hann window, lowpass_filter_width 6,
    rolloff 0.99), executed by the HIP kernel ``ppg_resample`` -- the ONE
    implementation in this package (torchaudio is absent in the build image:
    the kernel follows its published algorithm and is tested against a fixture
    computed from the closed-form filter in float64, tests/golden/g10).  Host
    tensors (file loading) make the round trip through HIP device `gpu`
    and come back on the host, like every other entry point of an engine
    without a CPU path.
    """
    if sample_rate == target_rate:
        return audio
    if audio.is_cuda:
        return engine.resample(audio, sample_rate, target_rate)
    device = device_for(gpu)
    return engine.resample(audio.to(device), sample_rate, target_rate).cpu()


###############################################################################
# PPG post-ops (reference ppgs/core.py:399-543), per-frame arithmetic on the GPU
###############################################################################


_similarity_cache = {}


def similarity_matrix(device=None):
    """The reference's 40 x 40 phoneme similarity matrix
    (ppgs.SIMILARITY_MATRIX_PATH, a data asset of the reference package, not of
    this one): read from the .pt file PPGS_AMD_SIMILARITY_MATRIX names (a bare
    tensor, loaded with weights_only=True) and cached per path.  Without it,
    pass ``similarity=`` to :func:`distance`."""
    path = os.environ.get('PPGS_AMD_SIMILARITY_MATRIX')
    if path is None:
        raise ValueError(
            'ppgs_amd.distance(normalize=True) needs the phoneme similarity '
            'matrix: pass similarity=<(40,40) tensor> or set '
            'PPGS_AMD_SIMILARITY_MATRIX to the reference\'s '
            'balanced_similarity.pt')
    if path not in _similarity_cache:
        matrix = torch.load(path, map_location='cpu', weights_only=True)
        if not torch.is_tensor(matrix) or tuple(matrix.shape) != (
                config.OUTPUT_CHANNELS, config.OUTPUT_CHANNELS):
            raise ValueError(
                f'{path}: expected a ({config.OUTPUT_CHANNELS}, '
                f'{config.OUTPUT_CHANNELS}) tensor')
        _similarity_cache[path] = matrix
    matrix = _similarity_cache[path]
    return matrix if device is None else matrix.to(device)


_mix_cache = {}


def _similarity_mix(similarity, exponent, device):
    """(S.T ** exponent) on the device, cached per (matrix, device, exponent)
    as the reference caches it per device (ppgs/core.py:432-441)."""
    key = (id(similarity), similarity._version, str(device), float(exponent))
    hit = _mix_cache.get(key)
    if hit is None or hit[0] is not similarity:
        if len(_mix_cache) > 16:
            _mix_cache.clear()
        mix = (similarity.to(device=device, dtype=torch.float32).T
               ** exponent).contiguous()
        _mix_cache[key] = hit = (similarity, mix)
    return hit[1]


def distance(ppgX, ppgY, reduction='mean', normalize=True,
             exponent=config.SIMILARITY_EXPONENT, similarity=None):
    """Pronunciation distance between two aligned (40, frames) PPGs: the
    similarity-normalised Jensen-Shannon distance of reference
    ppgs/core.py:399-472.  Extra keyword `similarity`: the matrix to use
    instead of the reference's asset (see similarity_matrix)."""
    if reduction not in ('mean', 'sum', 'none', None):
        raise ValueError(f'Reduction method {reduction} not defined')
    device = device_for(None, ppgX)
    mix = None
    if normalize:
        if similarity is None:
            similarity = similarity_matrix()
        mix = _similarity_mix(similarity, exponent, device)
    jsd = engine.distance_frames(ppgX.to(device), ppgY.to(device), mix)
    if reduction == 'mean':
        return jsd.mean(dim=0)
    if reduction == 'sum':
        return jsd.sum(dim=0)
    return jsd


def interpolate(ppgX, ppgY, interp):
    """Linear interpolation (reference ppgs/core.py:480-502)."""
    return (1. - interp) * ppgX + interp * ppgY


def sparsify(ppg, method='percentile', threshold=torch.Tensor([0.85])):
    """Make posteriorgrams sparse and renormalise (reference
    ppgs/core.py:510-543).  (batch, 40, frames) -> same; like the reference,
    'percentile' with a one-element threshold tensor (the default) returns the
    result with an extra leading dimension of 1 (torch.quantile's q axis)."""
    methods = {'constant': 0, 'percentile': 1, 'topk': 2}
    if method not in methods:
        raise ValueError(f'Sparsification method {method} not defined')
    device = device_for(None, ppg)
    value = float(threshold.reshape(-1)[0]) if torch.is_tensor(threshold) else float(threshold)
    if torch.is_tensor(threshold) and threshold.numel() != 1:
        raise ValueError('ppgs_amd.sparsify takes one threshold')
    out = engine.sparsify(ppg.to(device), methods[method], value)
    if method == 'percentile' and torch.is_tensor(threshold) and threshold.dim() == 1:
        out = out[None]
    if method == 'topk' and out.shape[0] == 1:
        pass    # the reference's batch-1 semantics; larger batches mix items there (core.py:537-538)
    return out


def representation_file_extension():
    """reference ppgs/core.py:611-621 for REPRESENTATION_KIND == 'ppg'."""
    if config.REPRESENTATION == config.BEST_REPRESENTATION:
        return '-ppg.pt'
    return f'-{config.REPRESENTATION}-ppg.pt'
