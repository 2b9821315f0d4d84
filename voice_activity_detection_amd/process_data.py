"""Feature-extraction entry points of the reference, computed by the HIP front end.

Mirrors the two hot functions of core/process_data.py:

* ``process_test_frames(data_frames, align_frames)``          (core/process_data.py:257-279)
* ``add_noise(speech, noise, align_frames, noise_level_db)``  (core/process_data.py:282-309)

with the same argument meaning and return triple ``(frames, mfcc, delta)``,
plus device-resident batch forms that the per-slice Python loop of
``process_test_data`` (core/process_data.py:214-237) collapses into:

* ``mfcc_delta_slices``  a batch of independently stored slices
* ``mfcc_delta_stream``  one contiguous stream cut into fixed-length slices
* ``mfcc_delta_ragged``  one contiguous stream cut into slices of different lengths
* ``overlay_device``     the int16 noise/speech mix of add_noise on device tensors
* ``process_test_data``  the whole per-slice loop of a recording as one call
* ``process_training_data``  the slice x noise-level loop of the training set: one mix and one
  front-end launch per noise level

All arithmetic runs in ``vad_mfcc_delta_f32`` / ``vad_mfcc_delta_slices_f32`` / ``vad_overlay_i16`` of
libvad_amd.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .common import ALIGN_FRAMES, FRAME_SIZE, MFCC_WINDOW_FRAME_SIZE, NOISE_LEVELS_DB, SEED, SLICE_MAX, SLICE_MIN


def _device():
    if not torch.cuda.is_available():
        raise _lib.VadError('no HIP device visible: the front end has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _as_frames(a, name):
    a = np.asarray(a)
    if a.dtype != np.int16:
        a = a.astype(np.int16)                 # callers pass np.int16(...) (core/process_data.py:232)
    if a.size % FRAME_SIZE:
        raise ValueError('%s: sample count %d is not a multiple of %d' % (name, a.size, FRAME_SIZE))
    return np.ascontiguousarray(a.reshape(-1, FRAME_SIZE))


def mfcc_delta_stream(stream_dev, n_frames, slice_len=SLICE_MAX, out=None):
    """Packed (n_frames, 24) float32 features of a contiguous device-resident stream.

    ``stream_dev``  int16 CUDA tensor with (n_frames + 3) * 480 samples: the 3
                    alignment rows of the first slice followed by the data rows.
    ``slice_len``   frames per slice (the reference's test path uses 166); <= 0
                    makes the whole stream one slice.  A ragged last slice is kept.
    Columns 0-11 are c1..c12, columns 12-23 their deltas (core/generator.py:145-148).
    """
    L = _lib.lib()
    if stream_dev.dtype != torch.int16 or not stream_dev.is_cuda or not stream_dev.is_contiguous():
        raise ValueError('stream_dev must be a contiguous int16 CUDA tensor')
    if stream_dev.numel() < (n_frames + ALIGN_FRAMES) * FRAME_SIZE:
        raise ValueError('stream_dev holds %d samples, need %d' %
                         (stream_dev.numel(), (n_frames + ALIGN_FRAMES) * FRAME_SIZE))
    if out is None:
        out = torch.empty((n_frames, 24), dtype=torch.float32, device=stream_dev.device)
    if n_frames == 0:
        return out
    if slice_len <= 0 or slice_len > n_frames:
        slice_len = n_frames
    n_slices = (n_frames + slice_len - 1) // slice_len
    last = n_frames - (n_slices - 1) * slice_len
    with torch.cuda.device(stream_dev.device):
        _lib.check(L.vad_mfcc_delta_f32(ctypes.c_void_p(stream_dev.data_ptr()), n_slices, slice_len, last,
                                        slice_len * FRAME_SIZE,
                                        ctypes.c_void_p(out.data_ptr()), 24,
                                        ctypes.c_void_p(out.data_ptr() + 12 * 4), 24, _lib.current_stream()))
    return out


def mfcc_delta_ragged(stream_dev, slice_starts, out=None):
    """Packed (n_frames, 24) float32 features of a contiguous device-resident stream cut into slices of
    different lengths, one launch (``vad_mfcc_delta_slices_f32``).

    ``stream_dev``    int16 CUDA tensor with (n_frames + 3) * 480 samples: 3 alignment rows, then the data rows.
    ``slice_starts``  ascending first-frame indices of the slices, ``[0, ..., n_frames]`` (n_slices + 1 entries);
                      an int64 CUDA tensor or any integer sequence.
    Every slice is treated as its own call of the reference's feature function: its first alignment row is
    not pre-emphasised and its deltas replicate its own edge rows.
    """
    L = _lib.lib()
    if stream_dev.dtype != torch.int16 or not stream_dev.is_cuda or not stream_dev.is_contiguous():
        raise ValueError('stream_dev must be a contiguous int16 CUDA tensor')
    host = slice_starts.cpu().numpy() if torch.is_tensor(slice_starts) else np.asarray(slice_starts)
    host = host.astype(np.int64).reshape(-1)
    if host.size < 2 or host[0] != 0 or np.any(np.diff(host) < 0):
        raise ValueError('slice_starts must be ascending, start at 0 and end at the frame count')
    n_frames = int(host[-1])
    if stream_dev.numel() < (n_frames + ALIGN_FRAMES) * FRAME_SIZE:
        raise ValueError('stream_dev holds %d samples, need %d' %
                         (stream_dev.numel(), (n_frames + ALIGN_FRAMES) * FRAME_SIZE))
    if out is None:
        out = torch.empty((n_frames, 24), dtype=torch.float32, device=stream_dev.device)
    if n_frames == 0:
        return out
    starts = torch.from_numpy(host).to(stream_dev.device)
    with torch.cuda.device(stream_dev.device):
        _lib.check(L.vad_mfcc_delta_slices_f32(ctypes.c_void_p(stream_dev.data_ptr()),
                                               ctypes.c_void_p(starts.data_ptr()), host.size - 1, n_frames,
                                               int(np.diff(host).max()),
                                               ctypes.c_void_p(out.data_ptr()), 24,
                                               ctypes.c_void_p(out.data_ptr() + 12 * 4), 24, _lib.current_stream()))
    return out


def mfcc_delta_slices(slices_dev):
    """(S, N+3, 480) int16 CUDA tensor of aligned slices -> (mfcc (S,N,12), delta (S,N,12)) float32."""
    L = _lib.lib()
    if slices_dev.dtype != torch.int16 or not slices_dev.is_cuda or slices_dev.dim() != 3 \
            or slices_dev.size(2) != FRAME_SIZE or slices_dev.size(1) <= ALIGN_FRAMES:
        raise ValueError('slices_dev must be an int16 CUDA tensor of shape (S, N+3, 480)')
    slices_dev = slices_dev.contiguous()
    S, rows = slices_dev.size(0), slices_dev.size(1)
    N = rows - ALIGN_FRAMES
    mfcc = torch.empty((S, N, 12), dtype=torch.float32, device=slices_dev.device)
    delta = torch.empty((S, N, 12), dtype=torch.float32, device=slices_dev.device)
    if S == 0:
        return mfcc, delta
    with torch.cuda.device(slices_dev.device):
        _lib.check(L.vad_mfcc_delta_f32(ctypes.c_void_p(slices_dev.data_ptr()), S, N, N, rows * FRAME_SIZE,
                                        ctypes.c_void_p(mfcc.data_ptr()), 12,
                                        ctypes.c_void_p(delta.data_ptr()), 12, _lib.current_stream()))
    return mfcc, delta


def process_test_frames(data_frames, align_frames):
    """Drop-in for core/process_data.py:257-279.

    ``data_frames`` (N,480) int16 and ``align_frames`` (3,480) int16 numpy
    arrays -> ``(frames (N,480) int16, mfcc (N,12), delta (N,12))``.  The
    reference returns float64 arrays that its callers store as float32
    (core/process_data.py:204-205, 236-237); here the values are computed in
    the kernel's precision and widened to float64 for the same dtype.
    """
    frames = _as_frames(data_frames, 'data_frames')
    align = _as_frames(align_frames, 'align_frames')
    if len(align) != ALIGN_FRAMES:
        raise ValueError('align_frames must hold %d frames' % ALIGN_FRAMES)
    if len(frames) == 0:
        raise ValueError('data_frames is empty')
    aligned = np.concatenate((align, frames))                          # core/process_data.py:270
    dev = torch.from_numpy(aligned).to(_device()).unsqueeze(0)
    mfcc, delta = mfcc_delta_slices(dev)
    return frames, mfcc[0].cpu().numpy().astype(np.float64), delta[0].cpu().numpy().astype(np.float64)


def recording_slices(n_frames):
    """The slices process_test_data cuts a recording of ``n_frames`` labelled frames into
    (core/process_data.py:190-195, 214-217): consecutive ``(start, start + SLICE_MAX)`` pairs while
    ``start + SLICE_MIN < n_frames``, of which those whose end is not below ``n_frames`` are skipped."""
    pos, slices = 0, []
    while pos + SLICE_MIN < n_frames:
        s = (pos, pos + SLICE_MAX)
        pos = s[1]
        if s[0] < n_frames and s[1] < n_frames:
            slices.append(s)
    return slices


def process_test_data(frames, labels):
    """In-memory form of the slice loop of process_test_data (core/process_data.py:186-246), which calls
    process_test_frames once per 166-frame slice with the previous slice's last 3 frames as alignment rows
    and stores the rows in HDF5 datasets.  Here the whole recording takes ONE device call; the result is a
    dict of arrays laid out like the datasets the loop fills: ``frames`` (int16), ``mfcc`` / ``delta``
    (float32, as stored at :204-205), ``labels`` (int8), each ``len(frames)`` rows of which the first
    ``count`` are written (the reference drops the ragged end of the recording)."""
    fr = _as_frames(frames, 'frames')
    lab = np.asarray(labels)
    total = len(fr)
    if len(lab) != total:
        raise ValueError('labels has %d entries, frames %d' % (len(lab), total))
    slices = recording_slices(total)
    count = slices[-1][1] if slices else 0
    out = {'frames': np.zeros((total, FRAME_SIZE), np.int16), 'mfcc': np.zeros((total, 12), np.float32),
           'delta': np.zeros((total, 12), np.float32), 'labels': np.zeros((total,), np.int8), 'count': count}
    if count:
        dev = _device()
        stream = torch.zeros((count + ALIGN_FRAMES) * FRAME_SIZE, dtype=torch.int16, device=dev)   # zero lead-in, :227
        stream[ALIGN_FRAMES * FRAME_SIZE:] = torch.from_numpy(fr[:count].reshape(-1)).to(dev)
        feat = mfcc_delta_stream(stream, count, SLICE_MAX).cpu().numpy()
        out['frames'][:count] = fr[:count]
        out['mfcc'][:count] = feat[:, :12]
        out['delta'][:count] = feat[:, 12:]
        out['labels'][:count] = lab[:count]
    return out


def training_slices(n_speech_frames, rng):
    """The slice plan of process_training_data (core/process_data.py:82-108) for a speech set of
    ``n_speech_frames`` labelled frames: random 33..166-frame cuts of the speech, as many frames again of
    silence in random 33..166-frame pieces (marked ``(length, length)``), shuffled.  ``rng`` is the
    ``numpy.random`` state the reference draws from (seeded with SEED at :76); the draw order is kept.
    Returns (slices, total rows of the output datasets)."""
    pos, slices = 0, []
    while pos + SLICE_MIN < n_speech_frames:
        cut = (pos, pos + rng.randint(SLICE_MIN, SLICE_MAX + 1))
        slices.append(cut)
        pos = cut[1]
    slices[-1] = (slices[-1][0], n_speech_frames)                        # remainder joins the last slice (:93)
    pos = 0
    while pos + SLICE_MIN < n_speech_frames:
        length = rng.randint(SLICE_MIN, SLICE_MAX + 1)
        slices.append((length, length))
        pos += length
    total = n_speech_frames + pos + MFCC_WINDOW_FRAME_SIZE
    rng.shuffle(slices)
    return slices, total


def process_training_data(speech_frames, speech_labels, noise_frames, seed=SEED):
    """In-memory form of process_training_data (core/process_data.py:64-168): the speech set is cut into random
    slices, silence slices are mixed in, every slice gets a random stretch of the noise set, and for each noise
    level in NOISE_LEVELS_DB order add_noise is called once per slice with the previous 3 output frames of that
    level as alignment rows.  Here the slices of a level are laid out as one stream on the device and the level
    takes ONE overlay launch and ONE front-end launch.

    Kept as the reference computes it (:139-153): inside the slice loop ``frames`` is overwritten by add_noise's
    result, so the second and third level mix the noise into the PREVIOUS level's output, not into the clean
    speech.  Returns a dict laid out like the HDF5 datasets the loop fills: ``frames-<key>`` (int16), ``mfcc-<key>``,
    ``delta-<key>`` (float32) for key in NOISE_LEVELS_DB, ``labels`` (int8), each ``total`` rows of which the
    first ``count`` are written, plus ``slices``."""
    speech = _as_frames(speech_frames, 'speech_frames')
    noise = _as_frames(noise_frames, 'noise_frames')
    labels = np.asarray(speech_labels)
    if len(labels) != len(speech):
        raise ValueError('speech_labels has %d entries, speech_frames %d' % (len(labels), len(speech)))
    if len(speech) <= SLICE_MIN:
        raise ValueError('the speech set needs more than %d frames' % SLICE_MIN)
    rng = np.random.RandomState(seed)                                     # np.random.seed(SEED), :76
    slices, total = training_slices(len(speech), rng)
    lengths = [s[0] if s[0] == s[1] else s[1] - s[0] for s in slices]
    starts = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    count = int(starts[-1])
    clean = np.zeros((count, FRAME_SIZE), np.int16)
    mixin = np.zeros((count, FRAME_SIZE), np.int16)
    lab = np.zeros((count,), np.int8)
    for s, lo, n in zip(slices, starts[:-1], lengths):
        if s[0] != s[1]:                                                  # speech; silence stays zero (:126-132)
            clean[lo: lo + n] = speech[s[0]: s[1]]
            lab[lo: lo + n] = labels[s[0]: s[1]]
        i = rng.randint(0, len(noise) - n)                                # :135
        mixin[lo: lo + n] = noise[i: i + n]
    dev = _device()
    noise_dev = torch.from_numpy(mixin.reshape(-1)).to(dev)
    prev = torch.from_numpy(clean.reshape(-1)).to(dev)
    starts_dev = torch.from_numpy(starts).to(dev)
    out = {'labels': np.zeros((total,), np.int8), 'count': count, 'slices': slices}
    out['labels'][:count] = lab
    for key, gain in NOISE_LEVELS_DB.items():
        stream = torch.zeros((count + ALIGN_FRAMES) * FRAME_SIZE, dtype=torch.int16, device=dev)   # zero lead-in, :143
        mixed = stream[ALIGN_FRAMES * FRAME_SIZE:]
        overlay_device(noise_dev, prev, gain, out=mixed)
        feat = mfcc_delta_ragged(stream, starts_dev).cpu().numpy()
        for name, dtype, width, val in (('frames-', np.int16, FRAME_SIZE, mixed.cpu().numpy().reshape(-1, FRAME_SIZE)),
                                        ('mfcc-', np.float32, 12, feat[:, :12]), ('delta-', np.float32, 12, feat[:, 12:])):
            out[name + key] = np.zeros((total, width), dtype)
            out[name + key][:count] = val
        prev = mixed
    return out


def overlay_device(noise_dev, speech_dev, gain_during_overlay=None, out=None):
    """Device form of ``noise_track.overlay(speech_track, gain_during_overlay=dB)``
    (core/process_data.py:293) on int16 CUDA tensors: the noise is scaled by 10^(dB/20)
    (floor, saturate) over the overlaid part and the speech is added with saturation; noise
    beyond the end of the speech is passed through.  A falsy gain (None or 0) leaves the
    noise unscaled, as in pydub.  ``out`` may be the noise tensor itself.
    """
    L = _lib.lib()
    for name, t in (('noise_dev', noise_dev), ('speech_dev', speech_dev)):
        if t.dtype != torch.int16 or not t.is_cuda or not t.is_contiguous():
            raise ValueError('%s must be a contiguous int16 CUDA tensor' % name)
    if out is None:
        out = torch.empty_like(noise_dev)
    elif out.dtype != torch.int16 or not out.is_cuda or not out.is_contiguous() or out.numel() != noise_dev.numel():
        raise ValueError('out must be a contiguous int16 CUDA tensor of the size of noise_dev')
    factor = 10.0 ** (float(gain_during_overlay) / 20.0) if gain_during_overlay else 1.0
    with torch.cuda.device(noise_dev.device):
        _lib.check(L.vad_overlay_i16(ctypes.c_void_p(noise_dev.data_ptr()), noise_dev.numel(),
                                     ctypes.c_void_p(speech_dev.data_ptr()), speech_dev.numel(),
                                     1 if gain_during_overlay else 0, factor,
                                     ctypes.c_void_p(out.data_ptr()), _lib.current_stream()))
    return out


def overlay(noise_frames, speech_frames, gain_during_overlay=None):
    """numpy in, numpy out wrapper of ``overlay_device``: (M,480) int16 mixed frames."""
    noise = _as_frames(noise_frames, 'noise_frames')
    speech = _as_frames(speech_frames, 'speech_frames')
    dev = _device()
    mixed = overlay_device(torch.from_numpy(noise).to(dev), torch.from_numpy(speech).to(dev), gain_during_overlay)
    return mixed.cpu().numpy()


def add_noise(speech_frames, noise_frames, align_frames, noise_level_db):
    """Drop-in for core/process_data.py:282-309: mix on the device, then the same MFCC/delta
    tail, without the mixed audio leaving the device in between."""
    noise = _as_frames(noise_frames, 'noise_frames')
    speech = _as_frames(speech_frames, 'speech_frames')
    align = _as_frames(align_frames, 'align_frames')
    if len(align) != ALIGN_FRAMES:
        raise ValueError('align_frames must hold %d frames' % ALIGN_FRAMES)
    if len(noise) == 0:
        raise ValueError('noise_frames is empty')
    dev = _device()
    aligned = torch.empty((1, len(noise) + ALIGN_FRAMES, FRAME_SIZE), dtype=torch.int16, device=dev)
    aligned[0, :ALIGN_FRAMES] = torch.from_numpy(align).to(dev)
    mixed = aligned[0, ALIGN_FRAMES:]                                    # contiguous view: rows 3.. of the slice
    overlay_device(torch.from_numpy(noise).to(dev), torch.from_numpy(speech).to(dev), noise_level_db, out=mixed)
    mfcc, delta = mfcc_delta_slices(aligned)
    return (mixed.cpu().numpy(), mfcc[0].cpu().numpy().astype(np.float64),
            delta[0].cpu().numpy().astype(np.float64))
