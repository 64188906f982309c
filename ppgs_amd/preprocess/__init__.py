"""Feature frontends (mirrors reference ppgs/preprocess/core.py:194-221)."""
import torch

from .. import config
from . import mel, spectrogram, w2v2fb


def from_audio(audio, representation=config.REPRESENTATION,
               sample_rate=config.SAMPLE_RATE, gpu=None):
    """Preprocess audio -> (batch, channels, frames) features
    (reference ppgs/preprocess/core.py:194-216)."""
    from .. import core
    audio = core.resample(audio, sample_rate)
    if representation is None:
        representation = config.REPRESENTATION
    if representation not in ('mel', 'w2v2fb'):
        raise ValueError(
            f'representation {representation!r} has no audio frontend here; '
            "compute the features yourself and call from_features")
    frontend = mel if representation == 'mel' else w2v2fb
    features = frontend.from_audio(audio, sample_rate=config.SAMPLE_RATE, gpu=gpu)
    if features.dim() == 2:
        features = features[None]
    return features


def save_masked(tensor, file, length):
    """Save the first `length` frames (reference preprocess/core.py:219-221)."""
    torch.save(tensor[..., :length].clone(), file)
