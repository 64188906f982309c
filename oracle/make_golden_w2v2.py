"""Fixture G11: the wav2vec 2.0 feature encoder of HF transformers itself.

The reference's w2v2fb representation (ppgs/preprocess/w2v2fb/core.py:45,66) runs
``transformers.Wav2Vec2Model.from_pretrained('facebook/wav2vec2-base')``; the
pretrained weights are not reachable here (no network), so -- SURVEY.md 8(c)(5) --
the same architecture is built with seeded random weights:

    torch.manual_seed(1234); model = Wav2Vec2Model(Wav2Vec2Config())

and its OWN ``feature_extractor`` is run on seeded audio.  The fixture stores the
audio, the module's output (``extract_features`` before the projection), the whole
model's ``last_hidden_state`` for the same input, and a checksum of the seeded
feature-encoder weights (so that a test rebuilding the model on another box can
tell whether it got the same weights).  4.2 M weights are not committed: every
consumer rebuilds them from the seed with the same torch / transformers image.

    python oracle/make_golden_w2v2.py        # writes tests/golden/g11_w2v2_features.npz
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as G          # noqa: E402

SEED = 1234


def seeded_model(seed=SEED):
    import transformers
    transformers.utils.logging.set_verbosity_error()
    torch.manual_seed(seed)
    return transformers.Wav2Vec2Model(transformers.Wav2Vec2Config()).eval()


def weight_checksum(module):
    total = torch.zeros((), dtype=torch.float64)
    for index, (_, tensor) in enumerate(sorted(module.state_dict().items())):
        total += (index + 1) * tensor.double().abs().sum()
    return float(total)


def main():
    import transformers
    model = seeded_model()
    audio = 0.1 * G.randn(111, 3, 6000)
    audio[1, 4100:] = 0.                        # a zero-padded (shorter) row: GroupNorm still runs over the whole row
    with torch.no_grad():
        features = model.feature_extractor(audio).transpose(1, 2)
        hidden = model(audio).last_hidden_state
    print('extract_features', tuple(features.shape), float(features.abs().max()), 'hidden', tuple(hidden.shape))
    G.save('g11_w2v2_features', audio=audio, features=features, last_hidden_state=hidden,
           seed=SEED, checksum=weight_checksum(model.feature_extractor),
           transformers_version=transformers.__version__)


if __name__ == '__main__':
    main()
