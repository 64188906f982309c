"""Golden vectors for the PPG post-ops: the reference's own
ppgs.distance / ppgs.sparsify / ppgs.interpolate (ppgs/core.py:399-543) and
ppgs.edit.grid.sample / constant / of_length (ppgs/edit/grid.py), imported
here with third-party stubs and run on seeded posteriorgrams.

    python oracle/make_golden_postops.py        # writes tests/golden/g9_postops.npz

The reference's similarity matrix (a 40 x 40 data asset, ppgs/assets/
balanced_similarity.pt) is an INPUT of ppgs.distance; it is stored in the fixture
as data so that the tests can hand it to the implementation under test.
"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as G          # noqa: E402


def main():
    ppgs = G.import_reference()
    sys.modules['torchaudio'] = types.ModuleType('torchaudio')
    torchutil = types.ModuleType('torchutil')
    torchutil.notify = lambda *a, **k: (lambda f: f)
    sys.modules['torchutil'] = torchutil
    core = G._load('ppgs.core', os.path.join(G.REF, 'ppgs', 'core.py'))
    similarity = torch.load(ppgs.SIMILARITY_MATRIX_PATH)

    def posteriors(seed, *shape, sharp=3.0):
        return torch.softmax(sharp * G.randn(seed, *shape), dim=-2)

    x, y = posteriors(1, 40, 57), posteriors(2, 40, 57)
    out = dict(similarity=similarity, exponent=ppgs.SIMILARITY_EXPONENT, x=x, y=y)
    for normalize in (True, False):
        for reduction in ('mean', 'sum', 'none'):
            out[f'distance_{int(normalize)}_{reduction}'] = core.distance(
                x.clone(), y.clone(), reduction=reduction, normalize=normalize)
    out['distance_same'] = core.distance(x.clone(), x.clone())
    interp = torch.linspace(0, 1, 57)
    out['interp'] = interp
    out['interpolate_scalar'] = core.interpolate(x, y, 0.3)
    out['interpolate_vector'] = core.interpolate(x, y, interp)
    batch = posteriors(3, 2, 40, 33)
    out['batch'] = batch
    out['sparsify_percentile'] = core.sparsify(batch.clone(), 'percentile', torch.tensor([0.85]))
    out['sparsify_percentile_50'] = core.sparsify(batch.clone(), 'percentile', torch.tensor([0.5]))
    out['sparsify_constant'] = core.sparsify(batch.clone(), 'constant', torch.tensor([0.1]))
    single = batch[:1].clone()
    out['sparsify_topk3'] = core.sparsify(single.clone(), 'topk', 3)
    # time-stretching (ppgs/edit/grid.py); pypar is only used by from_alignments
    pypar = types.ModuleType('pypar')
    pypar.Alignment = object            # annotation only
    sys.modules['pypar'] = pypar
    ppgs.interpolate = core.interpolate
    grid = G._load('ppgs.edit.grid', os.path.join(G.REF, 'ppgs', 'edit', 'grid.py'))
    out['grid_slow'] = grid.constant(x, 0.7)
    out['grid_fast'] = grid.of_length(x, 23)
    out['grid_edges'] = torch.tensor([0., 0.25, 1., 13.5, 55.999, 56., 56.4, 57., 80., -0.25, -3.5])
    for name in ('slow', 'fast', 'edges'):
        out[f'sample_{name}'] = grid.sample(x, out[f'grid_{name}'])
    out['sample_batch'] = grid.sample(batch, grid.of_length(batch, 50))
    out['sample_half'] = grid.sample(x.half(), out['grid_slow']).float()
    G.save('g9_postops', **out)
    for key, value in out.items():
        if isinstance(value, torch.Tensor):
            print(key, tuple(value.shape))


if __name__ == '__main__':
    main()
