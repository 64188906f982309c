"""ppgs_amd -- MI355X-native engine for the PPG inference path of
interactiveaudiolab/ppgs (``ppgs.from_audio`` / ``from_features`` /
``from_files_to_files`` forward).  Same ``from_*`` API and (batch, 40 phonemes,
frames) output layout; the arithmetic runs in hand-written gfx950 HIP kernels
behind the C ABI of ``include/ppgs_amd.h``.
"""
from .config import *                 # noqa: F401,F403
from .phonemes import PHONEMES, PHONEME_TO_INDEX_MAPPING   # noqa: F401
from . import config, data, engine, load, preprocess, weights   # noqa: F401
from .core import (                   # noqa: F401
    from_audio, from_features, from_file, from_file_to_file,
    from_files_to_files, from_dataloader, infer, resample,
    distance, interpolate, sparsify,
    representation_file_extension, engine_for, clear_cache)
from . import core, distributed, edit      # noqa: F401

__version__ = '0.1.0'
