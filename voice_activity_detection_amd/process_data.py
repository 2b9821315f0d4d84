"""Feature-extraction entry points of the reference, computed by the HIP front end.

Mirrors the two hot functions of core/process_data.py:

* ``process_test_frames(data_frames, align_frames)``          (core/process_data.py:257-279)
* ``add_noise(speech, noise, align_frames, noise_level_db)``  (core/process_data.py:282-309)

with the same argument meaning and return triple ``(frames, mfcc, delta)``,
plus device-resident batch forms that the per-slice Python loop of
``process_test_data`` (core/process_data.py:214-237) collapses into:

* ``mfcc_delta_slices``  a batch of independently stored slices
* ``mfcc_delta_stream``  one contiguous stream cut into fixed-length slices
* ``overlay_device``     the int16 noise/speech mix of add_noise on device tensors
* ``process_test_data``  the whole per-slice loop of a recording as one call

All arithmetic runs in ``vad_mfcc_delta_f32`` / ``vad_overlay_i16`` of libvad_amd.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .common import ALIGN_FRAMES, FRAME_SIZE, SLICE_MAX, SLICE_MIN


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
