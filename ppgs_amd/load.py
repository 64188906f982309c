"""Checkpoint and audio loading (mirrors reference ppgs/load.py:17-81)."""
import os

import numpy as np
import torch

from . import config, weights

# Set to a path to use a local checkpoint when none is given
# (reference ppgs/config/defaults.py:124)
LOCAL_CHECKPOINT = os.environ.get('PPGS_AMD_CHECKPOINT')

_HUB_REPO = 'CameronChurchwell/ppgs'
_HUB_FILES = {'mel': 'mel-800k.pt', 'w2v2fb': 'w2v2fb-425k.pt'}


def _decode_other(file):
    """Containers the native RIFF/WAV reader does not handle (the reference
    decodes mp3 / flac / ogg through torchaudio.load, ppgs/load.py:17-30): use
    soundfile, then torchaudio, whichever is installed AND can decode the file
    (an old libsndfile raises LibsndfileError -- a RuntimeError -- on mp3).
    A file nobody can decode is a ValueError, which the file pipelines turn
    into "skipped with a warning" instead of the end of the job."""
    failures = []
    try:
        import soundfile
        try:
            data, sample_rate = soundfile.read(os.fspath(file), dtype='float32', always_2d=True)
            return torch.from_numpy(np.ascontiguousarray(data.T)), int(sample_rate)
        except (RuntimeError, OSError, ValueError) as error:
            failures.append(f'soundfile: {error}')
    except ImportError:
        failures.append('soundfile: not installed')
    try:
        import torchaudio
        try:
            samples, sample_rate = torchaudio.load(os.fspath(file))
            return samples.to(torch.float32), int(sample_rate)
        except (RuntimeError, OSError, ValueError) as error:
            failures.append(f'torchaudio: {error}')
    except ImportError:
        failures.append('torchaudio: not installed')
    raise ValueError(f'{file}: not a RIFF/WAV file and no decoder for it ({"; ".join(failures)})')


def _is_riff(file):
    with open(os.fspath(file), 'rb') as handle:
        head = handle.read(12)
    # (big-endian RIFX is not something the native reader parses: it goes to the other decoders)
    return len(head) == 12 and head[:4] == b'RIFF' and head[8:12] == b'WAVE'


def audio(file, gpu=None):
    """Load an audio file as (channels, samples) fp32 at 16 kHz.

    A file at another sample rate is converted by the HIP resampler on device
    `gpu` (default: the current HIP device) -- it needs one; 16 kHz files do not.

    The reference goes through torchaudio.load + Resample
    (ppgs/load.py:17-30); here PCM/float WAV is decoded directly, other
    containers by soundfile / torchaudio if present, and other sample rates
    are converted by :func:`ppgs_amd.core.resample`.
    """
    from . import core
    if not _is_riff(file):
        samples, sample_rate = _decode_other(file)
        return core.resample(samples, sample_rate, gpu=gpu)
    from scipy.io import wavfile
    sample_rate, data = wavfile.read(os.fspath(file))
    if data.dtype == np.uint8:
        samples = (data.astype(np.float32) - 128.) / 128.
    elif np.issubdtype(data.dtype, np.integer):
        samples = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    else:
        samples = data.astype(np.float32)
    samples = torch.from_numpy(np.ascontiguousarray(samples))
    if samples.dim() == 1:
        samples = samples[None]
    else:
        samples = samples.T.contiguous()       # (channels, samples)
    return core.resample(samples, sample_rate, gpu=gpu)


def info(file):
    """(num_samples, sample_rate) from the RIFF header, without decoding
    (the reference uses torchaudio.info, ppgs/data/dataset.py:187); other
    containers are decoded to find out."""
    from . import engine
    if not _is_riff(file):
        samples, sample_rate = _decode_other(file)
        return int(samples.shape[-1]), sample_rate
    samples, rate, _ = engine.wav_info(file)
    return samples, rate


def state_dict(checkpoint=None, representation=None):
    """The model parameters as a reference-layout state dict.

    Resolution order follows reference ppgs/load.py:33-81: explicit
    ``checkpoint`` (a ``.pt`` path holding either a bare state_dict or
    ``{'model': state_dict}``; a dict is accepted as-is), else
    ``LOCAL_CHECKPOINT``, else the HF-hub file of the representation.
    """
    if representation not in (None, 'mel', 'w2v2fb'):
        raise ValueError(
            'Supplying representation directly only supported '
            'for w2v2fb and mel')
    if isinstance(checkpoint, dict):
        state = checkpoint
    else:
        if checkpoint is None:
            checkpoint = LOCAL_CHECKPOINT
        if checkpoint is None:
            import huggingface_hub
            name = _HUB_FILES.get(representation or config.REPRESENTATION)
            if name is None:
                raise ValueError(
                    f'No default checkpoints exist for '
                    f'representation {representation}')
            checkpoint = huggingface_hub.hf_hub_download(_HUB_REPO, name)
        state = torch.load(checkpoint, map_location='cpu', weights_only=True)
    if 'model' in state and not torch.is_tensor(state['model']):
        state = state['model']
    expected = weights.state_dict_shapes(*_geometry_args(state))
    for key, shape in expected.items():
        if key not in state:
            raise KeyError(f'checkpoint is missing {key}')
        if tuple(state[key].shape) != tuple(shape):
            raise ValueError(
                f'checkpoint {key} has shape {tuple(state[key].shape)}, '
                f'expected {tuple(shape)}')
    if representation is not None:
        want = config.MODEL_GEOMETRY[representation]
        cin, hidden, _ = weights.geometry(state)
        if (cin, hidden) != (want['input_channels'], want['hidden_channels']):
            raise ValueError(
                f'checkpoint geometry (Cin={cin}, H={hidden}) does not match '
                f'representation {representation}')
    return state


def _geometry_args(state):
    cin, hidden, layers = weights.geometry(state)
    return (cin, hidden, layers, state['output_layer.weight'].shape[0],
            state['input_layer.weight'].shape[2],
            (state['model.layers.0.linear1.weight'].shape[0]
             if layers else config.FFN_CHANNELS),
            state['position.encoding'].shape[0])


def model(checkpoint=None, representation=None, gpu=None, precision=None,
          is_causal=None):
    """Load a model onto a GPU -> :class:`ppgs_amd.engine.Engine`
    (counterpart of reference ppgs.load.model, ppgs/load.py:33-81)."""
    from . import core
    return core.engine_for(representation, checkpoint, gpu, precision, is_causal)
