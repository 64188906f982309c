"""Log-mel spectrogram on the GPU (reference ppgs/preprocess/mel.py)."""
from .. import config, engine


def from_audios(audio, lengths=None, sample_rate=config.SAMPLE_RATE, gpu=None):
    """(batch, 1, samples) fp32 -> (batch, 80, samples // 160) fp16.

    Same contract as reference mel.from_audios (mel.py:14-19): rows are
    already zero-extended to the batch's sample count; one fused kernel does
    reflect-pad, STFT, magnitude, fp16 round, Slaney mel, log, fp16 round.
    """
    from .. import core
    audio = audio.to(core.device_for(gpu, audio))
    _, mel = engine.frontend(audio, spectrogram=False, mel=True)
    return mel


def from_audio(audio, sample_rate=config.SAMPLE_RATE, gpu=None):
    """reference mel.from_audio (mel.py:22-30); the reference's autocast
    context there only affects its own matmul precision and has no
    counterpart here (the mel product is fp32)."""
    if audio.dim() == 2:
        audio = audio.unsqueeze(dim=0)
    return from_audios(audio, audio.shape[-1], sample_rate=sample_rate, gpu=gpu)
