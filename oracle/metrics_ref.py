"""CPU restatement of the evaluation helpers that consume the posteriors (core/train.py:448-536).

TEST INFRASTRUCTURE ONLY.  Pinned: scikit-learn, the library the reference calls, is importable
here, so ``fix_frr`` below is the reference's loop over ``sklearn.metrics.confusion_matrix`` and
ROC/AUC are ``sklearn.metrics.roc_curve`` / ``auc`` themselves.
"""
import numpy as np
from sklearn import metrics
from sklearn.metrics import confusion_matrix


def apply_threshold(y_score, t=0.5):
    """core/train.py:486-487; the scores are numpy.float32 scalars compared with a Python float.  The result
    depends on the numpy the reference runs on (requirements.txt does not pin it): numpy >= 2 converts the
    Python float to the scalar's type and compares in float32, numpy 1.x compared in float64.  The rule of
    numpy >= 2 is the one restated here, explicitly, whatever numpy is installed."""
    t32 = np.float32(t)
    return [1 if np.float32(y) >= t32 else 0 for y in y_score]


def fix_frr(y_true, y_score, frr_target, noise_level):
    """core/train.py:489-528."""
    if noise_level == 'None':
        t = 1e-4
    elif noise_level == '-15':
        t = 1e-5
    else:
        t = 1e-9
    far = frr = 0
    while t < 1.0:
        tn, fp, fn, tp = confusion_matrix(y_true, apply_threshold(y_score, t), labels=[0, 1]).ravel()
        far = (fp * 100) / (fp + tn) if (fp + tn) > 0 else 0
        frr = (fn * 100) / (fn + tp) if (fn + tp) > 0 else 0
        if frr >= frr_target:
            return far, frr
        t *= 1.1
    return far, frr


def roc_auc(y_true, y_score):
    """core/train.py:464-465."""
    fpr, tpr, thresholds = metrics.roc_curve(y_true, y_score)
    return fpr, tpr, thresholds, metrics.auc(fpr, tpr)
