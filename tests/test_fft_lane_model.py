"""The index algebra of the MFCC kernel's FFT (csrc/mfcc_device.h: 16 x 16 x 4, one LDS exchange, radix 4 across
the lanes of a quad, paired untangle with the partner value fetched from another lane) as tools/fft_lane_model.py
states it lane by lane: against numpy.fft, on the CPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import fft_lane_model as m  # noqa: E402


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_lane_model_matches_numpy_fft(seed):
    rng = np.random.default_rng(seed)
    x = np.zeros(2048)
    x[:1920] = rng.normal(size=1920) * 3000
    z = x[0::2] + 1j * x[1::2]
    v = m.fft1024_lanes(z)
    Z = np.zeros(1024, complex)
    for lane in range(64):
        for q in range(16):
            Z[(lane >> 2) + 16 * q + 256 * m.REV[lane & 3]] = v[lane, q]
    ref = np.fft.fft(z)
    assert np.abs(Z - ref).max() <= 1e-12 * np.abs(ref).max()
    P, cnt = m.power_from_lanes(v)
    assert (cnt == 1).all()                                   # every bin 0..1024 written exactly once
    pref = np.abs(np.fft.rfft(x)) ** 2
    assert np.abs(P - pref).max() <= 1e-12 * pref.max()


def test_untangle_steps_cover_every_pair_once():
    seen = set()
    for lane in range(64):
        for u in range(8):
            k = int(m.fft_bin(lane, u))
            assert 0 <= k < 1024
            pair = frozenset((k, 1024 - k))
            assert pair not in seen
            seen.add(pair)
    bins = set()
    for p in seen:
        bins |= set(p)
    assert bins | {512} == set(range(1025))
