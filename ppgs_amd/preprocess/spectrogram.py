"""Magnitude spectrogram on the GPU (reference ppgs/preprocess/spectrogram.py)."""
from .. import engine


def from_audios(audio, lengths=None, gpu=None):
    """(batch, 1, samples) fp32 -> (batch, 513, samples // 160) fp16.

    Same contract as reference spectrogram.from_audios (spectrogram.py:14-50);
    `lengths` is accepted and unused there too.  Runs on cuda:{gpu}, or on
    the tensor's own GPU when gpu is None.
    """
    from .. import core
    audio = audio.to(core.device_for(gpu, audio))
    spec, _ = engine.frontend(audio, spectrogram=True, mel=False)
    return spec


def from_audio(audio, sample_rate=None, gpu=None):
    if audio.dim() == 2:
        audio = audio.unsqueeze(dim=0)
    return from_audios(audio, audio.shape[-1], gpu=gpu)
