"""The C-ABI library: loads without a GPU, exports every symbol the header
declares, and its host-only planner agrees with the oracle's restatement of
the reference chunking (ppgs/model/transformer.py:49-64).  No compute calls."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import ppg_oracle as O
from ppgs_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'ppgs_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ppg_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = E.library()
    names = declared_functions()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), name
    # the Python binding covers the same set
    assert sorted(E.SYMBOLS) == names
    assert lib.ppg_abi_version() == 1


def test_struct_layouts_match_header():
    assert ctypes.sizeof(E.PpgConfig) == 12 * 4
    assert ctypes.sizeof(E.PpgWindow) == 12 * 4
    assert ctypes.sizeof(E.PpgWeights) == (3 + 12 * 16 + 2) * 8
    assert ctypes.sizeof(E.PpgPlanInfo) == 40


def reference_plan(T, lengths):
    """(item, start, frames, valid, keep_lo, keep_hi) per window, item-major."""
    rows = []
    windows = O.plan_windows(T, lengths)
    for b in range(len(lengths)):
        for w in windows:
            rows.append((b, w['start'], w['Tc'], min(w['clens'][b], w['Tc']),
                         w['keep_lo'], w['keep_hi']))
    return rows


@pytest.mark.parametrize('T,lengths', [
    (100, [100]), (500, [500, 1]), (501, [501]), (1000, [1000, 420, 30]),
    (1201, [1201, 800, 799, 401, 400, 0]), (4999, [4999, 17]),
])
def test_planner_matches_reference_chunking(T, lengths):
    windows, info = E.plan_windows(len(lengths), T, lengths)
    got = [(w.item, w.start, w.frames, w.valid, w.keep_lo, w.keep_hi)
           for w in windows]
    assert got == reference_plan(T, lengths)
    # kept columns tile the output exactly once per item
    for b in range(len(lengths)):
        frames = sorted(
            f for w in windows if w.item == b
            for f in range(w.out_frame, w.out_frame + w.keep_hi - w.keep_lo))
        assert frames == list(range(T))
    computed = [w for w in windows if w.valid > 0]
    assert info.num_windows == len(computed)
    assert info.skipped_windows == len(windows) - len(computed)
    assert info.processed_frames == sum(w.frames for w in computed)
    assert info.attention_pairs == sum(w.frames ** 2 for w in computed)
    # token rows: 16-aligned, disjoint, in order
    offset = 0
    for w in computed:
        assert w.tok_off == offset and w.tok_off % 16 == 0 and w.vt_off % 32 == 0
        offset += -(-w.frames // 16) * 16
    assert info.tokens == offset


def test_planner_random_lengths_property():
    rng = np.random.default_rng(0)
    for _ in range(25):
        T = int(rng.integers(1, 3000))
        lengths = [int(v) for v in rng.integers(0, T + 1, size=rng.integers(1, 6))]
        lengths[0] = T
        windows, _ = E.plan_windows(len(lengths), T, lengths)
        got = [(w.item, w.start, w.frames, w.valid, w.keep_lo, w.keep_hi)
               for w in windows]
        assert got == reference_plan(T, lengths)


@pytest.mark.parametrize('T,lengths,heads', [
    (1000, [1000] * 32, 2),                                   # C2: 64 long windows + 32 short ones
    (700, [700, 40, 300, 513, 700, 99, 17, 655], 2),          # ragged, exhausted windows
    (160, [160, 3, 77, 160, 64, 65], 2),                      # every window fits a narrow tile or two
    (1201, [1201, 800, 799, 401, 400, 0], 1),
])
def test_attention_items_cover_every_query_once_longest_first_xcd_affine(T, lengths, heads):
    """The attention launch order (host planner): every query row of every computed window is in
    exactly one tile; half-width tiles exactly for windows with <= half the longest window's keys
    or <= 64 rows; longest first inside each of the 8 / gcd(8, heads) interleaved lanes; all tiles
    of a window in ONE lane (they share an XCD's L2 for a given head)."""
    windows, _ = E.plan_windows(len(lengths), T, lengths)
    computed = [w for w in windows if w.tok_off >= 0]
    items = E.plan_attention_items(len(lengths), T, lengths, heads=heads)
    covered = {i: np.zeros(w.frames, dtype=int) for i, w in enumerate(computed)}
    longest = max(w.valid for w in computed)
    for it in items:
        w = computed[it.window]
        assert (it.frames, it.valid) == (w.frames, w.valid)
        assert it.narrow == int(2 * w.valid <= longest or w.frames <= 64)
        assert it.queries == (64 if it.narrow else 128) and it.q0 % it.queries == 0
        covered[it.window][it.q0:it.q0 + it.queries] += 1
    assert all((c == 1).all() for c in covered.values())
    lanes = 8 // math.gcd(8, heads)
    lane_of = {}
    for position, it in enumerate(items):
        lane_of.setdefault(it.window, set()).add(position % lanes)
    full = len(items) - len(items) % lanes           # (the interleave runs ragged once a lane is exhausted)
    multi = [w for w, ls in lane_of.items() if len(ls) > 1]
    assert all(any(p >= full - lanes * 8 for p, it in enumerate(items) if it.window == w) for w in multi)
    for lane in range(lanes):
        valids = [it.valid for p, it in enumerate(items) if p % lanes == lane and p < full // 2]
        assert valids == sorted(valids, reverse=True)


def test_planner_legacy_mode_and_errors():
    windows, _ = E.plan_windows(1, 1200, [1200], legacy_mode=True)
    assert len(windows) == 1 and windows[0].frames == 1200 and not windows[0].chunked
    with pytest.raises(ValueError):          # reference asserts T < 5000 (transformer.py:47-48)
        E.plan_windows(1, 5000, [5000], legacy_mode=True)
    with pytest.raises(ValueError):
        E.plan_windows(1, 100, [101])
    with pytest.raises(ValueError):
        E.plan_windows(2, 100, [100])


def test_compute_entry_points_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    lib = E.library()
    with pytest.raises(E.PpgError):
        E.Engine({})
    rc = lib.ppg_frontend(0, ctypes.c_void_p(16), 1, 16000, ctypes.c_void_p(16),
                          None, None)
    assert rc == -2 and b'no HIP device' in lib.ppg_last_error()
    import ppgs_amd
    with pytest.raises(E.PpgError):
        ppgs_amd.from_audio(torch.zeros(1, 1, 16000), 16000)
    # the satellites of the path have no CPU route either
    dummy = ctypes.c_void_p(16)
    assert lib.ppg_resample(0, dummy, 1, 48000, 48000, 16000, dummy, None) == -2
    assert b'no HIP device' in lib.ppg_last_error()
    assert lib.ppg_distance(0, dummy, dummy, 10, None, dummy, None) == -2
    assert lib.ppg_sparsify(0, dummy, 1, 10, 1, 0.85, dummy, None) == -2
    assert lib.ppg_grid_sample(0, dummy, 40, 10, dummy, 5, dummy, None) == -2
    with pytest.raises(E.PpgError):
        ppgs_amd.distance(torch.rand(40, 5), torch.rand(40, 5), normalize=False)


def test_resample_length_host_helper():
    lib = E.library()
    import math
    for samples, rate in ((48000, 48000), (44100, 44100), (12345, 22050), (1, 8000), (160001, 16001)):
        assert lib.ppg_resample_length(samples, rate, 16000) == math.ceil(16000 * samples / rate)
    assert lib.ppg_resample_length(-1, 48000, 16000) == -1
