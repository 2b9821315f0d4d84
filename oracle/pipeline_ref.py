"""CPU restatement of the glue around the hot path (TEST INFRASTRUCTURE).

* ``features_sliced``  the slice loop of core/process_data.py:171-255
  (``process_test_data``): fixed ``SLICE_MAX`` = 166-frame slices, 3 alignment
  frames taken from the end of the previous slice (zeros for the first),
  results stored as float32.
* ``windows``          stride-1 window assembly of core/train.py:566-570 /
  core/generator.py:145-148: ``hstack(mfcc[i:i+30], delta[i:i+30])``.
* ``score_windows``    batched forward in chunks of BATCH_SIZE = 2048 with the
  last batch zero-padded and trimmed (core/train.py:583-609, 618-620);
  posterior column 1 is the speech score (core/train.py:440).
* ``decide``           argmax (core/train.py:605-607) or ``y >= t``
  (core/train.py:486-487).
"""
import numpy as np
import torch

from . import mfcc_ref

BATCH_SIZE = 2048     # core/common.py:21
FRAMES = 30           # core/common.py:22
SLICE_MAX = 166       # core/process_data.py:53  int(5000 / 30)


def features_sliced(frames, slice_len=SLICE_MAX, first_align=None):
    """frames (F,480) int16 -> feat (F,24) float32, per-slice edge effects kept.

    ``slice_len`` <= 0 treats the whole stream as one slice.  A ragged last
    slice is processed with its own length (the reference's test path drops
    it, core/process_data.py:215-217; its training path keeps it, :93).
    """
    frames = np.asarray(frames, dtype=np.int16).reshape(-1, mfcc_ref.FRAME_SIZE)
    F = len(frames)
    if slice_len <= 0:
        slice_len = F
    if first_align is None:
        first_align = np.zeros((mfcc_ref.ALIGN_FRAMES, mfcc_ref.FRAME_SIZE), np.int16)   # :226-227
    feat = np.zeros((F, 24), np.float32)
    pos = 0
    while pos < F:
        end = min(pos + slice_len, F)
        align = first_align if pos == 0 else frames[pos - mfcc_ref.ALIGN_FRAMES: pos]     # :228-229
        _, m, d = mfcc_ref.process_test_frames(frames[pos:end], align)
        feat[pos:end, :12] = m          # float32 store, :204,236
        feat[pos:end, 12:] = d          # float32 store, :205,237
        pos = end
    return feat


def windows(feat, frames=FRAMES, stride=1):
    """(F,24) -> (F-frames+1 [strided], frames, 24) float32 view-copy of sliding windows.

    The reference's ``netvad`` iterates ``i < length - FRAMES`` and so leaves
    out the last possible window; callers that mirror it slice ``[:-1]``.
    """
    feat = np.asarray(feat, dtype=np.float32)
    n = (len(feat) - frames) // stride + 1
    if n <= 0:
        return np.zeros((0, frames, feat.shape[1]), np.float32)
    idx = stride * np.arange(n)[:, None] + np.arange(frames)[None, :]
    return feat[idx]


def score_windows(net, x, batch_size=BATCH_SIZE):
    """Posterior (B,2) float32 via batches of ``batch_size``, zero-padded like netvad."""
    x = np.asarray(x, dtype=np.float32)
    out = np.zeros((len(x), 2), np.float32)
    with torch.no_grad():
        for i in range(0, len(x), batch_size):
            chunk = x[i:i + batch_size]
            n = len(chunk)
            if n < batch_size:
                chunk = np.concatenate((chunk, np.zeros((batch_size - n,) + chunk.shape[1:], np.float32)))
            y = net(torch.from_numpy(chunk).float()).numpy()
            out[i:i + n] = y[:n]
    return out


def decide(p, threshold=None):
    """uint8 decisions: argmax over the 2 classes, or ``p[:,1] >= threshold``."""
    p = np.asarray(p)
    if threshold is None:
        return (p[:, 1] > p[:, 0]).astype(np.uint8)      # torch.max returns the first max on ties
    return (p[:, 1] >= np.float32(threshold)).astype(np.uint8)


def vad_stream(net, frames, slice_len=SLICE_MAX, threshold=None, batch_size=BATCH_SIZE):
    """End-to-end oracle: int16 frames -> (feat, posteriors per stride-1 window, decisions).

    Window ``i`` covers frames ``i .. i+29``; the reference aligns its decision
    with frame ``i + 15`` (core/train.py:600-601, 624-626).
    """
    feat = features_sliced(frames, slice_len)
    x = windows(feat)
    p = score_windows(net, x, batch_size)
    return feat, p, decide(p, threshold)


def process_training_data(speech_frames, speech_labels, noise_frames, seed=1337):
    """The loop of process_training_data (core/process_data.py:76-158) on in-memory arrays, one add_noise call
    (overlay_ref.overlay + mfcc_ref.process_test_frames) per slice and noise level, with numpy's legacy global
    generator seeded as at :76 and drawn from in the reference's order.  As in the reference (:139-153) the
    variable holding the slice's frames is overwritten by add_noise's result, so the levels after the first mix
    the noise into the previous level's output.  Returns the dict of arrays the HDF5 datasets would hold."""
    from . import overlay_ref
    SLICE_MIN, WINDOW = 33, 4                              # core/process_data.py:52, 49
    levels = {'None': None, '-15': -15, '-3': -3}          # core/common.py:25
    speech = np.asarray(speech_frames, np.int16).reshape(-1, 480)
    noise_set = np.asarray(noise_frames, np.int16).reshape(-1, 480)
    rng = np.random.RandomState(seed)
    pos, l, slices = 0, len(speech), []
    while pos + SLICE_MIN < l:                             # :87-90
        s = (pos, pos + rng.randint(SLICE_MIN, SLICE_MAX + 1))
        slices.append(s)
        pos = s[1]
    slices[-1] = (slices[-1][0], l)                        # :93
    pos = 0
    while pos + SLICE_MIN < l:                             # :98-102
        length = rng.randint(SLICE_MIN, SLICE_MAX + 1)
        slices.append((length, length))
        pos += length
    total = l + pos + WINDOW                               # :105
    rng.shuffle(slices)                                    # :108
    data = {'labels': np.zeros((total,), np.int8)}
    for key in levels:
        data['frames-' + key] = np.zeros((total, 480), np.int16)
        data['mfcc-' + key] = np.zeros((total, 12), np.float32)
        data['delta-' + key] = np.zeros((total, 12), np.float32)
    pos = 0
    for s in slices:                                       # :123-158
        if s[0] == s[1]:
            frames = np.zeros((s[0], 480))
            labels = np.zeros(s[0])
        else:
            frames = speech[s[0]: s[1]]
            labels = np.asarray(speech_labels)[s[0]: s[1]]
        i = rng.randint(0, len(noise_set) - len(labels))
        noise = noise_set[i: i + len(labels)]
        for key in levels:
            align = np.zeros((WINDOW - 1, 480)) if pos == 0 else data['frames-' + key][pos - WINDOW + 1: pos]
            mixed = overlay_ref.overlay(np.int16(noise), np.int16(frames), levels[key]).reshape(-1, 480)   # add_noise :283-297
            frames, mfcc, delta = mfcc_ref.process_test_frames(mixed, np.int16(align))                      # add_noise :299-309
            data['frames-' + key][pos: pos + len(labels)] = frames
            data['mfcc-' + key][pos: pos + len(labels)] = mfcc
            data['delta-' + key][pos: pos + len(labels)] = delta
        data['labels'][pos: pos + len(labels)] = labels
        pos += len(labels)
    data['count'] = pos
    data['slices'] = slices
    return data
