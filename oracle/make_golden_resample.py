"""Fixture G10 for the resampler: torchaudio.transforms.Resample's defaults
(sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99 -- what reference
ppgs/core.py:599-608 applies off 16 kHz) evaluated from the CLOSED FORM of the
published filter, per output sample, in float64:

    out[j] = scale * sum_i x[i] * w((i/orig - j/new) * base_freq)
    w(t)   = sinc(t) * cos^2(pi t / (2 * 6))  for |t| < 6, else 0
    base_freq = 0.99 * min(orig, new),  scale = base_freq / orig,
    orig, new = rates / gcd,  len(out) = ceil(new * len(x) / orig)

torchaudio itself is absent in the build image (parity against the package is
unpinned); this is an independent route to the same numbers -- no polyphase
kernel bank, no strided convolution, exact rational sample times -- against
which BOTH the oracle's restatement of torchaudio's kernel-bank formulation and
the HIP kernel are tested (tests/test_oracle_golden.py, tests/test_gpu_parity.py).

    python oracle/make_golden_resample.py      # writes tests/golden/g10_resample.npz
"""
import math
import os
import sys
from fractions import Fraction

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as G          # noqa: E402

RATES = (48000, 44100, 22050, 8000, 16001)


def closed_form(x, rate, target=16000):
    gcd = math.gcd(rate, target)
    orig, new = rate // gcd, target // gcd
    base = 0.99 * min(orig, new)
    scale = base / orig
    n = x.shape[-1]
    length = math.ceil(new * n / orig)
    out = np.zeros(x.shape[:-1] + (length,), dtype=np.float64)
    reach = 6.0 / base * orig                  # |i - j*orig/new| < reach
    for j in range(length):
        centre = Fraction(j * orig, new)
        lo = max(int(math.floor(float(centre) - reach)) - 1, 0)
        hi = min(int(math.ceil(float(centre) + reach)) + 1, n - 1)
        i = np.arange(lo, hi + 1)
        # (i/orig - j/new) * base, the difference taken exactly in integers first
        t = (i * new - j * orig).astype(np.float64) / (orig * new) * base
        w = np.where(np.abs(t) < 6.0,
                     np.sinc(t) * np.cos(np.pi * t / 12.0) ** 2, 0.0)
        out[..., j] = scale * (x[..., lo:hi + 1].astype(np.float64) * w).sum(-1)
    return out


def main():
    arrays = {}
    for rate in RATES:
        audio = (0.1 * G.randn(rate, 2, 1, rate // 40 + 17)).numpy()
        arrays[f'audio_{rate}'] = audio
        arrays[f'out_{rate}'] = closed_form(audio, rate)
        print(rate, audio.shape, '->', arrays[f'out_{rate}'].shape)
    G.save('g10_resample', **arrays)


if __name__ == '__main__':
    main()
