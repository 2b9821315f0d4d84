"""The evaluation restatement (oracle/metrics_ref.py) against hand-checked cases; sklearn is the pinned reference."""
import numpy as np

from oracle import metrics_ref
from voice_activity_detection_amd import metrics as vm


def test_fix_frr_hand_case():
    # 4 speech scores, 4 non-speech scores; noise level 'None' starts at t = 1e-4
    y_true = [1, 1, 1, 1, 0, 0, 0, 0]
    y_score = np.array([0.9, 0.8, 0.7, 2e-4, 0.5, 1e-5, 3e-4, 1e-6], np.float32)
    far, frr = metrics_ref.fix_frr(y_true, y_score, 1, 'None')
    # first t with a missed speech frame: t just above 2e-4 -> fn = 1 (25 %), fp = {0.5, 3e-4} = 2 of 4
    assert frr == 25.0 and far == 50.0


def test_threshold_table_matches_reference_loop():
    for lvl, t0 in (('None', 1e-4), ('-15', 1e-5), ('-3', 1e-9)):
        thr = vm.frr_thresholds(lvl)
        assert thr[0] == t0 and thr[-1] < 1.0 and thr[-1] * 1.1 >= 1.0
        t = t0
        for v in thr:                                   # bit-identical repeated multiplication
            assert v == t
            t *= 1.1
        assert len(thr) <= vm.MAX_THRESHOLDS
    assert len(vm.frr_thresholds('-3')) == 218


def test_fix_frr_never_reaching_target_returns_last():
    y_true = [1, 1, 0, 0]
    y_score = np.array([1.0, 1.0, 0.0, 0.0], np.float32)      # speech always >= any t < 1
    far, frr = metrics_ref.fix_frr(y_true, y_score, 1, '-15')
    assert (far, frr) == (0.0, 0.0)
