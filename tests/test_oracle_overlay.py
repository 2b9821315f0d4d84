"""Known-answer tests of the overlay restatement (oracle/overlay_ref.py).  PARITY UNPINNED: no pydub here."""
import numpy as np

from oracle import overlay_ref


def _mix(noise, speech, gain=None):
    return overlay_ref.overlay(np.array(noise, np.int16), np.array(speech, np.int16), gain).reshape(-1)


def test_plain_saturating_add():
    out = _mix([0, 100, 32767, -32768, 30000, -30000], [5, -100, 1, -1, 10000, -10000])
    np.testing.assert_array_equal(out, [5, 0, 32767, -32768, 32767, -32768])


def test_falsy_gain_leaves_noise_unscaled():
    a = _mix([1000, -1000], [1, 1], None)
    b = _mix([1000, -1000], [1, 1], 0)
    np.testing.assert_array_equal(a, [1001, -999])
    np.testing.assert_array_equal(a, b)


def test_gain_floors_towards_minus_infinity():
    f = 10.0 ** (-3 / 20.0)                         # 0.7079...
    out = _mix([10, -10, 3, -3, 1, -1], [0, 0, 0, 0, 0, 0], -3)
    expect = [int(np.floor(v * f)) for v in (10, -10, 3, -3, 1, -1)]
    np.testing.assert_array_equal(out, expect)      # 7, -8, 2, -3, 0, -1
    assert expect == [7, -8, 2, -3, 0, -1]


def test_positive_gain_saturates_before_the_add():
    out = _mix([30000, -30000, 30000], [0, 0, -100], 6)      # x1.995 -> clipped to 32767 / -32768 first
    np.testing.assert_array_equal(out, [32767, -32768, 32667])


def test_noise_longer_than_speech_passes_through_unscaled():
    out = _mix([1000] * 6, [1, 1], -15)
    f = 10.0 ** (-15 / 20.0)
    np.testing.assert_array_equal(out, [int(np.floor(1000 * f)) + 1] * 2 + [1000] * 4)


def test_speech_longer_than_noise_is_cut():
    out = _mix([1, 2], [10, 20, 30, 40])
    np.testing.assert_array_equal(out, [11, 22])
