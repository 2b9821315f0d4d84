"""Known-answer tests of the overlay restatement (oracle/overlay_ref.py), and its arithmetic held against the
standard library's audioop, the C module pydub's overlay calls (pydub itself is not importable here)."""
import warnings

import numpy as np
import pytest

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


# ---- against audioop.mul + audioop.add, the two calls pydub's overlay makes per overlaid stretch ----

def _audioop():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        return pytest.importorskip('audioop')


def _audioop_overlay(noise, speech, gain):
    """pydub.AudioSegment.overlay(position=0, loop=False) on raw int16 data: the overlaid stretch of the base is
    scaled with audioop.mul when the gain is truthy, mixed with audioop.add, and the rest of the base is kept."""
    audioop = _audioop()
    n = min(noise.size, speech.size)
    seg = noise[:n].tobytes()
    if gain:
        seg = audioop.mul(seg, 2, 10.0 ** (gain / 20.0))
    mixed = np.frombuffer(audioop.add(seg, speech[:n].tobytes(), 2), np.int16)
    return np.concatenate((mixed, noise[n:]))


@pytest.mark.parametrize('gain', [None, -15, -3, 0, 6, 20, -60])      # core/common.py:25 uses None, -15, -3
def test_matches_audioop_on_random_and_edge_samples(gain):
    rng = np.random.default_rng(7)
    edges = np.array([-32768, -32767, -32766, -1, 0, 1, 32766, 32767, 16384, -16384, 23170, -23170], np.int16)
    noise = np.concatenate((rng.integers(-32768, 32768, 48000).astype(np.int16), np.repeat(edges, len(edges))))
    speech = np.concatenate((rng.integers(-32768, 32768, 48000).astype(np.int16), np.tile(edges, len(edges))))
    np.testing.assert_array_equal(overlay_ref.overlay(noise, speech, gain).reshape(-1),
                                  _audioop_overlay(noise, speech, gain))


@pytest.mark.parametrize('n_noise,n_speech', [(480, 480), (960, 481), (481, 960), (7, 3), (1, 1)])
def test_matches_audioop_on_ragged_lengths(n_noise, n_speech):
    rng = np.random.default_rng(n_noise * 1000 + n_speech)
    noise = rng.integers(-32768, 32768, n_noise).astype(np.int16)
    speech = rng.integers(-32768, 32768, n_speech).astype(np.int16)
    for gain in (None, -15, -3):
        np.testing.assert_array_equal(overlay_ref.overlay(noise, speech, gain).reshape(-1),
                                      _audioop_overlay(noise, speech, gain))


def test_every_int16_value_through_the_reference_gains():
    noise = np.arange(-32768, 32768, dtype=np.int64).astype(np.int16)
    for gain in (-15, -3):
        for s in (0, 1, -1, 32767, -32768):
            speech = np.full(noise.shape, s, np.int16)
            np.testing.assert_array_equal(overlay_ref.overlay(noise, speech, gain).reshape(-1),
                                          _audioop_overlay(noise, speech, gain))
