"""GPU parity of the threshold sweep / FAR@FRR / ROC helpers: exact integer counts against numpy,
the reference's sklearn-based loop, and sklearn.metrics itself."""
import numpy as np
import pytest
import torch
from sklearn import metrics as skm

from oracle import metrics_ref

pytestmark = pytest.mark.gpu


def _scores(n, seed, kind='vad'):
    rng = np.random.default_rng(seed)
    y = (rng.random(n) < 0.4).astype(np.int64)
    if kind == 'vad':                                   # posteriors pile up near 0 and 1, as a softmax's do
        z = rng.normal(0, 3, n) + np.where(y == 1, 4.0, -6.0)
        s = 1.0 / (1.0 + np.exp(-z))
    else:
        s = rng.random(n)
    return y, s.astype(np.float32)


def _brute(y, s, thr):
    s64 = s.astype(np.float64)
    tp = np.array([np.sum((s64 >= t) & (y == 1)) for t in thr], np.int64)
    fp = np.array([np.sum((s64 >= t) & (y == 0)) for t in thr], np.int64)
    return tp, fp


@pytest.mark.parametrize('n,kind', [(1, 'vad'), (63, 'vad'), (5000, 'vad'), (100003, 'uniform'), (100003, 'vad')])
def test_threshold_sweep_exact(vad, n, kind):
    y, s = _scores(n, n, kind)
    thr = np.sort(np.r_[vad.metrics.frr_thresholds('-3'), s[:: max(n // 50, 1)].astype(np.float64), 0.0, 1.0, 2.0])
    tp, fp, pos, neg = vad.threshold_sweep(y, s, thr)
    btp, bfp = _brute(y, s, thr)
    np.testing.assert_array_equal(tp, btp)
    np.testing.assert_array_equal(fp, bfp)
    assert pos == int(y.sum()) and neg == n - int(y.sum())


def test_threshold_sweep_special_values_and_strided(vad):
    s = np.array([0.0, 1.0, np.nan, 0.5, np.float32(0.1), -0.0, 1e-45, np.inf, 0.5, 0.5], np.float32)
    y = np.array([1, 1, 1, 0, 1, 0, 1, 0, 1, 0])
    thr = np.array([0.0, 1e-45, 0.1, float(np.float32(0.1)), 0.5, 1.0])   # 0.1 < float32(0.1) as doubles
    tp, fp, pos, neg = vad.threshold_sweep(y, s, thr)
    btp, bfp = _brute(y, s, thr)                          # NaN >= t is False: below every threshold
    np.testing.assert_array_equal(tp, btp)
    np.testing.assert_array_equal(fp, bfp)
    # column 1 of an (n,2) posterior array, read in place
    p = torch.tensor(np.stack([1 - np.nan_to_num(s), s], 1)).cuda()
    tp2, fp2, _, _ = vad.threshold_sweep(torch.tensor(y).cuda(), p, thr)
    np.testing.assert_array_equal(tp2, btp)
    np.testing.assert_array_equal(fp2, bfp)


def test_threshold_sweep_long_table_and_errors(vad):
    y, s = _scores(20000, 3)
    thr = np.linspace(0, 1, 5000)                          # longer than one kernel table: chunked
    tp, fp, _, _ = vad.threshold_sweep(y, s, thr)
    btp, bfp = _brute(y, s, thr)
    np.testing.assert_array_equal(tp, btp)
    np.testing.assert_array_equal(fp, bfp)
    with pytest.raises(ValueError):
        vad.threshold_sweep(y, s, [0.5, 0.4])
    with pytest.raises(ValueError):
        vad.threshold_sweep(y[:-1], s, [0.5])


@pytest.mark.parametrize('lvl', ['None', '-15', '-3'])
def test_fix_frr_matches_reference_loop(vad, lvl):
    y, s = _scores(4000, 17)
    got = vad.fix_frr(y, s, 1, lvl)
    ref = metrics_ref.fix_frr(list(y), list(s), 1, lvl)
    assert got == (float(ref[0]), float(ref[1]))


def test_fix_frr_score_equal_to_a_rounded_threshold(vad):
    """float32(1e-4) < 1e-4 as doubles: a score of exactly float32(1e-4) reaches the first candidate threshold
    under the float32 comparison of numpy >= 2 (and of the kernels' own decision), not under a double one."""
    t0 = np.float32(1e-4)
    assert float(t0) < 1e-4
    y = np.array([1] * 50 + [0] * 50)
    s = np.r_[t0, np.full(49, 0.9), np.full(25, 1.05e-4), np.full(25, 0.9)].astype(np.float32)
    got = vad.fix_frr(y, s, 1, 'None')
    ref = metrics_ref.fix_frr(list(y), list(s), 1, 'None')
    assert got == (float(ref[0]), float(ref[1]))
    # at t = 1e-4 the tied speech score is still accepted (FRR 0, FAR 100 %); the search stops at the next
    # candidate, 1.1e-4, with FRR 2 % and FAR 50 %.  A double comparison would have stopped at 1e-4: (100, 2).
    assert got == (50.0, 2.0)
    tp, fp, _, _ = vad.threshold_sweep(y, s, [float(t0), 1e-4])
    assert tp[0] == 50 and tp[1] == 49                   # the sweep itself compares exactly, in double


def test_fix_frr_target_not_reached(vad):
    y = np.array([1, 1, 0, 0])
    s = np.array([1.0, 1.0, 0.0, 0.0], np.float32)
    assert vad.fix_frr(y, s, 1, '-15') == metrics_ref.fix_frr(list(y), list(s), 1, '-15')


@pytest.mark.parametrize('n,kind', [(2000, 'vad'), (50000, 'uniform'), (50000, 'vad')])
def test_roc_curve_and_auc_match_sklearn(vad, n, kind):
    y, s = _scores(n, n + 1, kind)
    s = np.round(s, 3) if kind == 'uniform' else s         # force ties
    fpr, tpr, thr = vad.roc_curve(y, s)
    rf, rt, rth, rauc = metrics_ref.roc_auc(y, s)
    np.testing.assert_array_equal(fpr, rf)
    np.testing.assert_array_equal(tpr, rt)
    np.testing.assert_array_equal(thr[1:], rth[1:].astype(np.float64))
    assert abs(vad.metrics.auc(fpr, tpr) - rauc) <= 1e-12
    assert abs(vad.metrics.auc(fpr, tpr) - skm.roc_auc_score(y, s)) <= 1e-9


def test_sweep_at_scale_properties(vad):
    """100 M scores (8.3 h of stride-1 windows x 100): counts stay exact where torch can check them."""
    n = 100_000_000
    g = torch.Generator(device='cuda').manual_seed(5)
    z = torch.randn(n, device='cuda', generator=g) * 3
    lab = (torch.rand(n, device='cuda', generator=g) < 0.4)
    s = torch.sigmoid(z + torch.where(lab, 4.0, -6.0))
    thr = vad.metrics.frr_thresholds('-3')
    tp, fp, pos, neg = vad.threshold_sweep(lab, s, thr)
    assert pos == int(lab.sum()) and neg == n - pos
    assert np.all(np.diff(tp) <= 0) and np.all(np.diff(fp) <= 0)
    s64 = s.double()
    for k in (0, 57, 150, len(thr) - 1):
        ge = s64 >= thr[k]
        assert tp[k] == int((ge & lab).sum()) and fp[k] == int((ge & ~lab).sum())
