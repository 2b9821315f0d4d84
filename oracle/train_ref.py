"""CPU restatement of the loops around the model forward in the reference's evaluation code: the generator's
test batches (core/generator.py:13-165), ``test_predict`` (core/train.py:393-445), ``far`` (:480-536) and
``netvad`` up to its plotting call (:539-626).

TEST INFRASTRUCTURE ONLY.  ``core.generator`` and ``core.train`` cannot be imported here (pydub, h5py_cache,
webrtcvad, IPython are absent), so these are restatements written from the source lines cited at each step,
run with the CPU model oracle (oracle/models_ref.py, pinned to outputs of the reference's own classes).  The
random draws go through numpy's global generator in the reference's order, so a seed gives the reference's
windows.  The loops are written as plain functions over an explicit state dict, not as the reference's class.
"""
import numpy as np
import torch

from . import metrics_ref

BATCH_SIZE, FRAMES, FEATURES = 2048, 30, 24          # core/common.py:21-23
STEP_SIZE, SEED = 6, 1337                            # core/train.py:54, core/generator.py:10
NOISE_LEVELS = ['None', '-15', '-3']                 # core/common.py:25, core/train.py:53


def test_partition(data, size_limit=0, frame_count=FRAMES, step_size=STEP_SIZE, batch_size=BATCH_SIZE):
    """What DataGenerator(data, size_limit) + setup_generation + use_test_data leave behind: the partition's first
    row, its size and the number of batches (core/generator.py:15-24, 43-58, 82-89).  Seeds and draws from numpy's
    global generator exactly as the constructor does (:22-24)."""
    size = size_limit if size_limit > 0 else len(data['labels'])
    indices = np.arange(size)
    np.random.seed(SEED)
    np.random.shuffle(indices)
    test_index = int(size * (1 - 0.1))
    test_size = size - test_index
    n = int((test_size - frame_count) / step_size) + 1
    return {'size': size, 'initial_pos': test_index, 'test_size': test_size, 'batch_count': int(n / batch_size),
            'frame_count': frame_count, 'step_size': step_size, 'batch_size': batch_size}


def draw_batch(data, part, noise_level, index):
    """get_batch (core/generator.py:111-165): the balanced, shuffled sample of window offsets inside the batch's
    span and the materialised (x, y)."""
    fc, st, bs = part['frame_count'], part['step_size'], part['batch_size']
    pos = part['initial_pos'] + (bs * index) * st                                  # :115
    span = fc + st * bs                                                            # :118
    assert pos >= 0 and pos + span <= part['size']                                 # get_data's assert, :92
    mfcc = np.asarray(data['mfcc-' + noise_level][pos: pos + span])
    delta = np.asarray(data['delta-' + noise_level][pos: pos + span])
    labels = np.asarray(data['labels'][pos: pos + span])
    zeros, ones = np.where(labels == 0)[0], np.where(labels == 1)[0]               # :122-123
    half = bs // 2                                                                 # :131
    picked = np.hstack((np.random.choice(zeros, half, replace=True),               # :134-138
                        np.random.choice(ones, half, replace=True)))
    np.random.shuffle(picked)                                                      # :139
    x, y = [], []
    for i in picked:                                                               # :143-153
        a, b = mfcc[i: i + fc], delta[i: i + fc]
        if a.shape[0] == fc and b.shape[0] == fc:
            x.append(np.hstack((a, b)))
            y.append(labels[i])
    return np.array(x), np.array(y)


def test_predict(net, data, size_limit, noise_level):
    """core/train.py:393-445 with a CPU torch module."""
    part = test_partition(data, size_limit)
    if noise_level not in NOISE_LEVELS or part['test_size'] == 0:
        return None
    net.eval()
    y_true, y_score = [], []
    for i in range(part['batch_count']):
        x, y = draw_batch(data, part, noise_level, i)
        if len(x) == 0 or len(y) == 0:
            continue
        with torch.no_grad():
            out = net(torch.from_numpy(np.array(x)).float())
        y_true.extend(y)
        y_score.extend(out.numpy()[:, 1])                                          # :440
    return y_true, y_score


def far(net, data, size_limit=0, frr=1):
    """core/train.py:532-536: (far, frr) per noise level."""
    out = {}
    for lvl in NOISE_LEVELS:
        y_true, y_score = test_predict(net, data, size_limit, lvl)
        out[lvl] = metrics_ref.fix_frr(y_true, y_score, frr, lvl)
    return out


def netvad(net, data, noise_level='-3', init_pos=50, length=700):
    """core/train.py:539-626: (raw_frames, labels, accum_out, posteriors) -- the three arrays handed to
    Vis.plot_evaluation, plus the posteriors of the scored windows for tolerance checks."""
    part = test_partition(data)                                                    # :547-548, 559
    assert init_pos >= 0 and init_pos + length <= part['size']
    raw = data['frames-' + noise_level][init_pos: init_pos + length]               # :562
    mfcc = np.asarray(data['mfcc-' + noise_level][init_pos: init_pos + length])
    delta = np.asarray(data['delta-' + noise_level][init_pos: init_pos + length])
    labels = data['labels'][init_pos: init_pos + length]
    frames, i = [], 0
    while i < length - FRAMES:                                                     # :566-570
        frames.append(np.hstack((mfcc[i: i + FRAMES], delta[i: i + FRAMES])))
        i += 1
    batches, i, num_frames = [], 0, -1                                             # :583-593
    while i < len(frames):
        full = i + BATCH_SIZE >= len(frames)
        window = frames[i: (len(frames) if full else i + BATCH_SIZE)]
        if full:
            num_frames = len(window)
            while len(window) < BATCH_SIZE:
                window.append(np.zeros((FRAMES, FEATURES)))
        batches.append(np.stack(window))
        i += BATCH_SIZE
    offset = 15                                                                    # :600
    accum_out, post = [0] * offset, []
    net.eval()
    for batch in batches:                                                          # :602-609
        with torch.no_grad():
            p = net(torch.from_numpy(batch).float())
        accum_out.extend(torch.max(p, 1)[1].float().numpy())
        post.append(p.numpy())
    if num_frames > 0:                                                             # :619-620
        accum_out = accum_out[:len(accum_out) - (BATCH_SIZE - num_frames)]
    accum_out = np.array(accum_out)
    post = np.concatenate(post)[:len(frames)] if post else np.zeros((0, 2), np.float32)
    return raw[offset:-offset], labels[offset:-offset], accum_out[offset:], post   # :624-626
