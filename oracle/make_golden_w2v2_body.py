"""Fixture G12: the wav2vec 2.0 transformer body of HF transformers itself.

As G11 (oracle/make_golden_w2v2.py): the pretrained 'facebook/wav2vec2-base' weights are not
reachable here, so the same architecture is built with seeded random weights and its OWN
``feature_projection`` + ``encoder`` modules are run, with a frame-level attention mask, on
seeded ``extract_features``.  The fixture stores the input, the valid lengths, the modules'
output and a checksum of the seeded body weights; every consumer rebuilds the 90 M weights
from the seed with the same torch / transformers image.

    python oracle/make_golden_w2v2_body.py        # writes tests/golden/g12_w2v2_body.npz
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as G                      # noqa: E402
from oracle import make_golden_w2v2 as M                 # noqa: E402

VALID = (70, 47, 9)


def hf_body(model, features, valid):
    """feature_projection + encoder as Wav2Vec2Model.forward calls them (modeling_wav2vec2.py)"""
    frames = features.shape[1]
    mask = torch.arange(frames)[None] < torch.as_tensor(valid)[:, None]
    hidden, _ = model.feature_projection(features)
    return model.encoder(hidden, attention_mask=mask.to(features.device)).last_hidden_state


def body_checksum(model):
    return M.weight_checksum(model.feature_projection) + M.weight_checksum(model.encoder)


def main():
    import transformers
    model = M.seeded_model()
    features = G.randn(212, len(VALID), max(VALID), 512)
    with torch.no_grad():
        hidden = hf_body(model, features, VALID)
    print('last_hidden_state', tuple(hidden.shape), float(hidden.abs().max()))
    G.save('g12_w2v2_body', features=features, valid=torch.tensor(VALID), last_hidden_state=hidden,
           seed=M.SEED, checksum=body_checksum(model), transformers_version=transformers.__version__)


if __name__ == '__main__':
    main()
