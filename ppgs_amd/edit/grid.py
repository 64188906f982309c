"""Grid-based PPG interpolation: the time-stretching part of the reference's
ppgs.edit (ppgs/edit/grid.py).  `sample` runs on the GPU (ppg_grid_sample);
the grid constructors are host arithmetic.
"""
import torch

from .. import config, core, engine


def sample(ppg, grid):
    """PPG (..., frames) at the float-valued frame indices `grid` (length,)
    -> (..., length), linear interpolation between neighbouring frames with the
    final frame replicated (reference ppgs/edit/grid.py:13-45)."""
    device = core.device_for(None, ppg)
    return engine.grid_sample(ppg.to(device), grid.to(device))


def constant(ppg, ratio):
    """Grid for constant-ratio time-stretching; lower ratio is slower
    (reference ppgs/edit/grid.py:53-65)."""
    return of_length(ppg, round(ppg.shape[-1] / ratio + 1e-4))


def of_length(ppg, length):
    """Grid resampling the PPG to `length` frames (reference
    ppgs/edit/grid.py:109-126)."""
    return torch.linspace(0., ppg.shape[-1] - 1., length, dtype=torch.float, device=ppg.device)


def from_alignments(source, target, sample_rate=config.SAMPLE_RATE, hopsize=config.HOPSIZE):
    """Grid converting the source forced alignment to the target's timing
    (reference ppgs/edit/grid.py:68-106).  Alignments are pypar.Alignment
    objects; pypar does the per-frame rate comparison, as in the reference."""
    import pypar
    source_frames = int((source.duration() * sample_rate) / hopsize)
    target_frames = int((target.duration() * sample_rate) / hopsize)
    rates = pypar.compare.per_frame_rate(target, source, sample_rate, hopsize, target_frames)
    indices = torch.cumsum(torch.tensor(rates), 0)
    indices -= indices[0].clone()
    indices *= (source_frames - 1) / indices[-1]
    return indices
