"""Known-answer tests for oracle/mfcc_ref.py.

The reference holds no golden vectors for python_speech_features (PARITY
UNPINNED), so the restatement is anchored on facts that follow from the
published algorithm and from the exact call at core/process_data.py:270-277.
"""
import numpy as np
import pytest

from oracle import mfcc_ref, pipeline_ref, synth_ref


def test_mel_bin_edges_and_filterbank_shape():
    edges = mfcc_ref.mel_bin_edges().astype(int).tolist()
    assert edges == [0, 8, 18, 28, 40, 53, 67, 82, 99, 118, 138, 160, 185, 211, 241, 273, 309, 348,
                     391, 438, 490, 546, 609, 677, 752, 834, 925, 1024]
    fb = mfcc_ref.mel_filterbank()
    assert fb.shape == (26, 1025)
    assert np.count_nonzero(fb) == 1915
    assert np.all(fb[:, 1024] == 0) and np.all(fb[:, 0] == 0)
    # each filter peaks at exactly 1 on its centre bin
    for j in range(26):
        assert fb[j, edges[j + 1]] == 1.0
    # neighbouring triangles are complementary between their shared edges
    for j in range(25):
        seg = slice(edges[j + 1], edges[j + 2])
        np.testing.assert_allclose(fb[j, seg] + fb[j + 1, seg], 1.0, atol=1e-15)


def test_lifter_constants():
    w = mfcc_ref.lifter_weights()[1:]
    np.testing.assert_allclose(w, [2.565463, 4.099058, 5.569565, 6.947049, 8.203468, 9.313245, 10.253789,
                                   11.005952, 11.554423, 11.888036, 12.0, 11.888036], atol=1e-6)


def test_all_zero_slice_hits_eps_clamp():
    z = np.zeros((13, 480), np.int16)
    _, m, d = mfcc_ref.process_test_frames(z[3:], z[:3])
    assert m.shape == (10, 12) and d.shape == (10, 12)
    assert np.abs(m).max() < 1e-12 and np.abs(d).max() < 1e-12
    le = mfcc_ref.log_mel_energies(z)
    np.testing.assert_allclose(le, np.log(np.finfo(float).eps))


def test_preemphasis_has_lag_of_one_row():
    rng = np.random.default_rng(1)
    a = rng.integers(-2000, 2000, size=(6, 480)).astype(np.int16)
    y = mfcc_ref.preemphasis_2d(a)
    assert y.shape == (6 * 480,) and y.dtype == np.float64
    np.testing.assert_array_equal(y[:480], a[0])
    np.testing.assert_allclose(y[480:], (a[1:] - 0.97 * a[:-1]).ravel())


def test_comb_cancellation():
    """Rows that decay by 0.97 are annihilated by the lag-480 pre-emphasis (up to int16 rounding)."""
    base = np.full(480, 20000.0)
    rows = [np.round(base * 0.97 ** k) for k in range(8)]
    a = np.array(rows).astype(np.int16)
    y = mfcc_ref.preemphasis_2d(a).reshape(8, 480)
    assert np.abs(y[1:]).max() <= 0.97 * 0.5 + 0.5      # only rounding residue survives


def test_frame_count_and_support():
    rng = np.random.default_rng(2)
    a = rng.integers(-3000, 3000, size=(12, 480)).astype(np.int16)
    sig = mfcc_ref.preemphasis_2d(a)
    fr = mfcc_ref.frame_signal(sig, 0.12 * 16000, 0.03 * 16000)
    assert fr.shape == (9, 1920)
    for t in range(9):
        np.testing.assert_array_equal(fr[t], sig[480 * t: 480 * t + 1920])
    # frame t depends only on input rows t-1 .. t+3 of the aligned array
    m0 = mfcc_ref.mfcc_slice(a)
    b = a.copy()
    b[8:] = rng.integers(-3000, 3000, size=(4, 480))
    m1 = mfcc_ref.mfcc_slice(b)
    np.testing.assert_array_equal(m0[:5], m1[:5])
    assert np.abs(m0[5:] - m1[5:]).max() > 1e-3


def test_sinusoid_on_filter_centre():
    """A bin-centred sinusoid puts an analytically known energy into its filter."""
    k0 = 160                      # centre bin of filter 10 (edges 138,160,185)
    n = np.arange(1920)
    amp = 1000.0
    frame = amp * np.cos(2 * np.pi * k0 * n / 2048)
    p = mfcc_ref.power_spectrum(frame[None, :])[0]
    fb = mfcc_ref.mel_filterbank()
    e = fb @ p
    # Dirichlet-kernel main lobe: |X[k0]| ~= amp*1920/2, so P[k0] = (amp*960)^2/2048 to within leakage
    assert abs(p[k0] / ((amp * 960) ** 2 / 2048) - 1) < 1e-2
    assert np.argmax(e) == 10
    assert e[10] > 20 * max(e[8], e[12])
    # the centre bin carries weight 1, so the filter energy is at least the peak-bin power
    assert e[10] >= p[k0]


def test_delta_ramp_and_edges():
    f = np.outer(np.arange(10, dtype=float), np.ones(12)) * 3.0
    d = mfcc_ref.delta(f, 2)
    np.testing.assert_allclose(d[2:-2], 3.0)
    # replicate padding: d[0] = (f1 + 2 f2 - 3 f0)/10, d[1] = (-2 f0 - f0 + f2 + 2 f3)/10
    np.testing.assert_allclose(d[0], (3 + 2 * 6 - 0) / 10)
    np.testing.assert_allclose(d[1], (-0 - 0 + 6 + 2 * 9) / 10)
    np.testing.assert_allclose(d[-1], (-2 * 21 - 24 + 3 * 27) / 10)


def test_slice_boundary_effect():
    """Row 0 of every slice is not pre-emphasised, and deltas replicate per slice."""
    fr = synth_ref.synth_frames(40, 400, seed=7)       # start past the leading segments
    whole = pipeline_ref.features_sliced(fr, slice_len=0)
    sliced = pipeline_ref.features_sliced(fr, slice_len=166)
    # first slice identical except for its last two delta rows
    np.testing.assert_array_equal(whole[:166, :12], sliced[:166, :12])
    np.testing.assert_array_equal(whole[:164, 12:], sliced[:164, 12:])
    # the first frame of the second slice sees an un-pre-emphasised first row
    assert np.abs(whole[166, :12] - sliced[166, :12]).max() > 1e-3
    np.testing.assert_array_equal(whole[167:330, :12], sliced[167:330, :12])
    assert np.abs(whole[166:168, 12:] - sliced[166:168, 12:]).max() > 1e-4


def test_process_test_frames_returns_input_frames():
    fr = synth_ref.synth_frames(0, 20, seed=3)
    frames, m, d = mfcc_ref.process_test_frames(fr[3:], fr[:3])
    np.testing.assert_array_equal(frames, fr[3:])
    assert frames.dtype == np.int16 and m.dtype == np.float64 and d.dtype == np.float64


def test_synth_is_position_independent_and_mixed():
    a = synth_ref.synth_audio(0, 48000, seed=1337)
    b = synth_ref.synth_audio(12345, 1000, seed=1337)
    np.testing.assert_array_equal(a[12345:13345], b)
    assert a.dtype == np.int16
    big = synth_ref.synth_audio(0, 16000 * 30, seed=1337).reshape(-1, 4000)
    silent = (np.abs(big).max(1) == 0).mean()
    assert 0.05 < silent < 0.6
    assert np.abs(big).max() > 8000
    # 64-bit sample index: no wrap at 2**32
    c = synth_ref.synth_audio(2 ** 32 - 8, 16, seed=1)
    d = synth_ref.synth_audio(2 ** 32, 8, seed=1)
    np.testing.assert_array_equal(c[8:], d)


def test_windows_layout():
    feat = np.arange(40 * 24, dtype=np.float32).reshape(40, 24)
    w = pipeline_ref.windows(feat)
    assert w.shape == (11, 30, 24)
    for i in (0, 5, 10):
        np.testing.assert_array_equal(w[i], np.hstack((feat[i:i + 30, :12], feat[i:i + 30, 12:])))
    assert pipeline_ref.windows(feat[:29]).shape == (0, 30, 24)
    assert pipeline_ref.windows(feat, stride=6).shape == (2, 30, 24)


def test_decide():
    p = np.array([[0.5, 0.5], [0.4, 0.6], [0.7, 0.3]], np.float32)
    np.testing.assert_array_equal(pipeline_ref.decide(p), [0, 1, 0])
    np.testing.assert_array_equal(pipeline_ref.decide(p, 0.5), [1, 1, 0])
    np.testing.assert_array_equal(pipeline_ref.decide(p, 0.3), [1, 1, 1])


def test_golden_slices_reproduce():
    """The committed vectors (tests/golden/make_mfcc_fixtures.py) pin the restatement against regressions."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mfcc_slices.npz'))
    for s in range(g['rows'].shape[0]):
        _, m, d = mfcc_ref.process_test_frames(g['rows'][s, 3:], g['rows'][s, :3])
        np.testing.assert_allclose(m, g['mfcc'][s], rtol=0, atol=1e-9)
        np.testing.assert_allclose(d, g['delta'][s], rtol=0, atol=1e-9)
    # the zeroed frames of slice 1 put whole analysis windows into the eps clamp: cepstra vanish there
    assert np.abs(g['mfcc'][1][23:28]).max() < 1e-9


def test_gain_invariance_of_c1_to_c12():
    """Doubling the input adds log 4 to every log filter energy; DCT rows k >= 1 annihilate a constant, so
    c1..c12 and their deltas do not change (away from the zero-energy clamp)."""
    rng = np.random.default_rng(8)
    rows = rng.integers(-8000, 8000, size=(43, 480)).astype(np.int16)
    _, m1, d1 = mfcc_ref.process_test_frames(rows[3:], rows[:3])
    _, m2, d2 = mfcc_ref.process_test_frames((2 * rows[3:]).astype(np.int16), (2 * rows[:3]).astype(np.int16))
    np.testing.assert_allclose(m2, m1, rtol=0, atol=1e-9)
    np.testing.assert_allclose(d2, d1, rtol=0, atol=1e-9)
