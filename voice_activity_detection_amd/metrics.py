"""Consumers of the posteriors, on the device: threshold sweeps, FAR at fixed FRR, ROC / AUC.

Mirrors the evaluation helpers of core/train.py:

* ``fix_frr``        the search inside ``far`` (core/train.py:489-528)
* ``roc_curve/auc``  ``sklearn.metrics.roc_curve`` / ``auc`` as called at core/train.py:464-465

``threshold_sweep`` runs in ``vad_threshold_sweep_f32`` of libvad_amd.so (one pass over the scores
for all thresholds).  ``roc_curve`` needs the scores sorted; the sort itself is delegated to
``torch.sort`` (rocPRIM radix sort) and only bookkeeping is added around it.
"""
import ctypes

import numpy as np
import torch

from . import _lib

MAX_THRESHOLDS = 2048


def _scores_labels(y_true, y_score):
    """Device views: (score tensor, element stride, uint8 labels, n)."""
    if torch.is_tensor(y_score) and y_score.is_cuda:
        s = y_score
    else:
        if not torch.cuda.is_available():
            raise _lib.VadError('no HIP device visible: the metrics have no CPU fallback')
        s = torch.as_tensor(np.asarray(y_score, dtype=np.float32)).cuda()
    if s.dtype != torch.float32:
        s = s.float()
    stride = 1
    if s.dim() == 2 and s.size(1) == 2:                 # an (n,2) posterior array: column 1 in place
        if not s.is_contiguous():
            s = s.contiguous()
        base, stride = s, 2
        n = s.size(0)
        ptr = s.data_ptr() + 4
    else:
        s = s.reshape(-1).contiguous()
        base, n, ptr = s, s.numel(), s.data_ptr()
    if torch.is_tensor(y_true) and y_true.is_cuda:
        lab = (y_true.reshape(-1) != 0).to(torch.uint8)
    else:
        lab = torch.as_tensor((np.asarray(y_true).reshape(-1) != 0).astype(np.uint8)).to(base.device)
    if lab.numel() != n:
        raise ValueError('y_true has %d entries, y_score %d' % (lab.numel(), n))
    return base, ptr, stride, lab.contiguous(), n


def threshold_sweep(y_true, y_score, thresholds):
    """Per threshold ``t`` of the ascending table: ``tp = #(label 1, score >= t)`` and
    ``fp = #(label 0, score >= t)``; also the class totals.  Returns int64 numpy arrays
    ``(tp (T,), fp (T,), positives, negatives)``."""
    L = _lib.lib()
    thr = np.ascontiguousarray(np.asarray(thresholds, dtype=np.float64).reshape(-1))
    if thr.size == 0:
        raise ValueError('thresholds is empty')
    if np.any(np.isnan(thr)) or np.any(np.diff(thr) < 0):
        raise ValueError('thresholds must be ascending numbers')
    base, ptr, stride, lab, n = _scores_labels(y_true, y_score)
    dev = base.device
    tp = np.empty(thr.size, np.int64)
    fp = np.empty(thr.size, np.int64)
    pos = neg = 0
    with torch.cuda.device(dev):
        for lo in range(0, thr.size, MAX_THRESHOLDS):            # tables longer than the kernel's LDS table
            part = thr[lo:lo + MAX_THRESHOLDS]
            T = part.size
            thr_dev = torch.from_numpy(part).to(dev)
            hist = torch.empty(2 * (T + 1), dtype=torch.int64, device=dev)
            ge = torch.empty((T, 2), dtype=torch.int64, device=dev)
            total = torch.empty(2, dtype=torch.int64, device=dev)
            _lib.check(L.vad_threshold_sweep_f32(ctypes.c_void_p(ptr), stride, ctypes.c_void_p(lab.data_ptr()), n,
                                                 ctypes.c_void_p(thr_dev.data_ptr()), T,
                                                 ctypes.c_void_p(hist.data_ptr()), ctypes.c_void_p(ge.data_ptr()),
                                                 ctypes.c_void_p(total.data_ptr()), _lib.current_stream()))
            g = ge.cpu().numpy()
            fp[lo:lo + T], tp[lo:lo + T] = g[:, 0], g[:, 1]
            neg, pos = (int(v) for v in total.cpu().numpy())
    return tp, fp, pos, neg


def frr_thresholds(noise_level):
    """The candidate thresholds of fix_frr: t0 * 1.1**k while < 1.0, built by the same repeated
    multiplication (core/train.py:491-500, 525)."""
    if noise_level == 'None':
        t = 1e-4
    elif noise_level == '-15':
        t = 1e-5
    else:
        t = 1e-9
    out = []
    while t < 1.0:
        out.append(t)
        t *= 1.1
    return out


def fix_frr(y_true, y_score, frr_target, noise_level):
    """FAR (%) at the first threshold whose FRR (%) reaches ``frr_target``; the last candidate's
    pair if none does.  Same result as fix_frr in core/train.py:489-528, from one pass over the scores.

    The reference compares numpy.float32 scores with a Python float threshold (``y >= t``).  Under
    numpy >= 2 that comparison is made in float32, i.e. against the threshold rounded to float32 --
    the same rule as the model kernels' ``p1 >= threshold`` decision -- so the candidates are rounded
    to float32 before the sweep (whose own comparison is exact, in double).  numpy 1.x compared in
    float64; the two rules differ only for a score equal to a rounded threshold."""
    thr = [float(np.float32(t)) for t in frr_thresholds(noise_level)]
    tp, fp, pos, neg = threshold_sweep(y_true, y_score, thr)
    far = frr = 0
    for k in range(len(thr)):
        tn, fn = neg - int(fp[k]), pos - int(tp[k])
        far = (int(fp[k]) * 100) / (int(fp[k]) + tn) if (int(fp[k]) + tn) > 0 else 0
        frr = (fn * 100) / (fn + int(tp[k])) if (fn + int(tp[k])) > 0 else 0
        if frr >= frr_target:
            return far, frr
    return far, frr


def roc_curve(y_true, y_score, drop_intermediate=True):
    """``sklearn.metrics.roc_curve`` (core/train.py:464) on the device: ``(fpr, tpr, thresholds)`` numpy
    arrays.  Counts are integers, so fpr/tpr are the exactly rounded quotients sklearn returns."""
    base, ptr, stride, lab, n = _scores_labels(y_true, y_score)
    s = base[:, 1] if stride == 2 else base
    order = torch.sort(s, descending=True, stable=True)
    ss, yy = order.values, lab[order.indices].to(torch.int64)
    last = torch.ones(n, dtype=torch.bool, device=ss.device)
    if n > 1:
        last[:-1] = ss[1:] != ss[:-1]                           # last element of every run of equal scores
    idx = torch.nonzero(last).reshape(-1)
    tps = torch.cumsum(yy, 0)[idx]
    fps = idx + 1 - tps
    thr = ss[idx]
    tps, fps, thr = tps.cpu().numpy(), fps.cpu().numpy(), thr.cpu().numpy().astype(np.float64)
    if drop_intermediate and len(fps) > 2:                       # collinear points, as sklearn drops them
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    thr = np.r_[np.inf, thr]
    fpr = fps / fps[-1] if fps[-1] > 0 else np.full(fps.shape, np.nan)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.full(tps.shape, np.nan)
    return fpr, tpr, thr


def auc(fpr, tpr):
    """Trapezoidal area, ``sklearn.metrics.auc`` (core/train.py:465)."""
    fpr, tpr = np.asarray(fpr, np.float64), np.asarray(tpr, np.float64)
    return float(np.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) * 0.5))
