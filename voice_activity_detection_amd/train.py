"""The inference-side callers of the reference's ``core/train.py`` with their names and argument meaning:
``test_predict`` (:393-445), ``far`` (:480-536), ``roc_auc`` (:448-477) and ``netvad`` (:539-630).

They are the loops that sit directly on either side of the model forward -- the generator's batches going in,
the scores going into the FAR / ROC evaluation, the frame decisions coming out -- rebuilt around the device
kernels: features are uploaded once per noise level and the models read their windows from that matrix
(``score_rows`` for the generator's sampled windows, a stride-1 ``score`` for ``netvad``), so no (B,30,24)
batch is ever assembled; FAR@FRR comes from one threshold sweep instead of one pass over the scores per
candidate threshold.  Training (``train_net``), checkpoints and plotting are not part of this path: the two
functions that end in a plot in the reference return what the reference hands to its plotting call.

``data`` is what the reference's functions take: a mapping with ``labels`` and, per noise level of
core/common.py:25, ``frames-<lvl>``, ``mfcc-<lvl>`` and ``delta-<lvl>`` (an open HDF5 file, or a dict of
arrays such as ``process_test_data`` returns).
"""
import time

import numpy as np
import torch

from . import metrics
from .common import BATCH_SIZE, FEATURES, FRAMES, NOISE_LEVELS_DB, STEP_SIZE
from .generator import DataGenerator
from .models import _cuda_device
from .stream import DECISION_OFFSET

NOISE_LEVELS = list(NOISE_LEVELS_DB.keys())            # core/train.py:53


def _feature_matrix(data, noise_level, first, last, device):
    """Rows [first, last) of the stored features as one packed (n,24) float32 device matrix: columns 0-11 the
    MFCCs, 12-23 their deltas (the np.hstack of core/generator.py:145-148, core/train.py:569)."""
    mfcc = np.asarray(data['mfcc-' + noise_level][first:last], dtype=np.float32)
    delta = np.asarray(data['delta-' + noise_level][first:last], dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(np.hstack((mfcc, delta)))).to(device)


def test_predict(net, data, size_limit, noise_level):
    """core/train.py:393-445: ``(y_true, y_score)`` over the test partition, the generator drawing the same
    windows from the same seed.  Returns None after the reference's message when the noise level is unknown or
    there is no test data."""
    generator = DataGenerator(data, size_limit)
    generator.setup_generation(FRAMES, STEP_SIZE, BATCH_SIZE)
    if noise_level not in NOISE_LEVELS:
        print('Error: invalid noise level!')
        return
    if generator.test_size == 0:
        print('Error: no test data was found!')
        return
    net.eval()
    generator.use_test_data()
    generator.set_noise_level_db(noise_level)

    # batch i draws its windows from rows initial_pos + i * batch * step .. + frames + step * batch: one upload of
    # the rows all batches draw from
    first = generator.initial_pos
    last = first + generator.batch_count * BATCH_SIZE * STEP_SIZE + FRAMES
    feat = None
    y_true, y_score = [], []
    for i in range(generator.batch_count):
        rows, y = generator.get_batch_rows(i)
        if len(rows) == 0 or len(y) == 0:
            continue
        if feat is None:
            feat = _feature_matrix(data, noise_level, first, last, _cuda_device())
        out = net.score_rows(feat, rows - first)[0]
        y_true.extend(np.asarray(y))
        y_score.extend(out[:, 1].cpu().numpy())
    print("y_true distribution:", np.bincount(np.asarray(y_true, dtype=np.int64)) if len(y_true) else [])
    print("y_score distribution:", np.unique(y_score, return_counts=True))
    return y_true, y_score


def far(net, data, size_limit=0, frr=1, model_name=''):
    """core/train.py:480-536: FAR at the first threshold whose FRR reaches ``frr`` percent, per noise level;
    prints the reference's lines and also returns ``{level: (far, frr)}``."""
    print('Network metrics: ' + model_name)
    results = {}
    for lvl in NOISE_LEVELS:
        y_true, y_score = test_predict(net, data, size_limit, lvl)
        results[lvl] = metrics.fix_frr(np.asarray(y_true), np.asarray(y_score, dtype=np.float32), frr, lvl)
        print('FAR: %0.2f%% for fixed FRR at %0.2f%% and noise level' % results[lvl], lvl)
    return results


def roc_auc(nets, data, noise_lvl, size_limit=0):
    """core/train.py:448-477 without the figure: ``{name: (fpr, tpr, thresholds, auc)}``, the arrays
    ``sklearn.metrics.roc_curve`` / ``auc`` give the reference's plot."""
    curves = {}
    for key in nets:
        y_true, y_score = test_predict(nets[key], data, size_limit, noise_lvl)
        fpr, tpr, thresholds = metrics.roc_curve(np.asarray(y_true), np.asarray(y_score, dtype=np.float32))
        curves[key] = (fpr, tpr, thresholds, metrics.auc(fpr, tpr))
    return curves


def netvad(net, data, noise_level='-3', init_pos=50, length=700, title=None, timeit=True):
    """core/train.py:539-630 up to its plotting call: returns ``(raw_frames, labels, accum_out)`` as the
    reference passes them to ``Vis.plot_evaluation`` -- the frames and labels with 15 entries cut from either end
    and the network's argmax decision per remaining frame (window i decides frame i + 15, :600-626).  Like the
    reference's ``while i < length - FRAMES`` it scores length - 30 windows, one fewer than the sample holds."""
    generator = DataGenerator(data)
    generator.setup_generation(FRAMES, STEP_SIZE, BATCH_SIZE)
    if noise_level not in NOISE_LEVELS:
        print('Error: invalid noise level!')
        return
    if generator.test_size == 0:
        print('Error: no test data was found!')
        return
    net.eval()
    generator.use_test_data()
    generator.set_noise_level_db(noise_level)
    raw_frames, mfcc, delta, labels = generator.get_data(init_pos, init_pos + length)

    n_windows = max(length - FRAMES, 0)
    dev = _cuda_device()
    feat = torch.from_numpy(np.ascontiguousarray(np.hstack((np.asarray(mfcc, dtype=np.float32),
                                                            np.asarray(delta, dtype=np.float32))))).to(dev)
    if timeit:
        torch.cuda.synchronize(dev)
        start_net = time.time()
    if n_windows:
        _, decision = net.score(feat, window_stride=FEATURES, n_windows=n_windows)
        accum_out = decision.cpu().numpy().astype(np.float32)
    else:
        accum_out = np.zeros(0, np.float32)
    if timeit:
        dur_net = str((time.time() - start_net) * 1000).split('.')[0]
        batches = -(-n_windows // BATCH_SIZE)
        seq_dur = int((length / 100) * 3)
        print(f'Network processed {batches * BATCH_SIZE} frames ({seq_dur}s) in {dur_net}ms on GPU.')
    offset = DECISION_OFFSET
    return raw_frames[offset:-offset], labels[offset:-offset], accum_out
