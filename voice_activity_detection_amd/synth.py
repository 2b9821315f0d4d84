"""Deterministic synthetic 16 kHz int16 audio, generated on the device.

Stands in for the datasets the reference reads through core/prepare_files.py
(LibriSpeech / QUT-NOISE / private recordings, none of which ship).  Sample n
of stream ``seed`` is a pure integer function of (seed, n).
"""
import ctypes

import torch

from . import _lib
from .common import ALIGN_FRAMES, FRAME_SIZE


def synth_audio(first_sample, n_samples, seed=1337, device=None, out=None):
    """int16 CUDA tensor with samples [first_sample, first_sample + n_samples)."""
    L = _lib.lib()
    if out is None:
        device = device or torch.device('cuda', torch.cuda.current_device())
        out = torch.empty((n_samples,), dtype=torch.int16, device=device)
    with torch.cuda.device(out.device):
        _lib.check(L.vad_synth_audio_i16(ctypes.c_void_p(out.data_ptr()), first_sample, n_samples, seed,
                                         _lib.current_stream()))
    return out


def synth_stream(first_frame, n_frames, seed=1337, device=None):
    """Aligned stream for ``mfcc_delta_stream``: 3 alignment rows + n_frames rows.

    For ``first_frame == 0`` the alignment rows are zeros (core/process_data.py:226-227);
    otherwise they are the 3 frames that precede ``first_frame`` in the same stream,
    which is what a shard that starts mid-stream needs.
    """
    device = device or torch.device('cuda', torch.cuda.current_device())
    buf = torch.zeros(((n_frames + ALIGN_FRAMES) * FRAME_SIZE,), dtype=torch.int16, device=device)
    lead = min(first_frame, ALIGN_FRAMES)
    start = first_frame - lead
    synth_audio(start * FRAME_SIZE, (n_frames + lead) * FRAME_SIZE, seed,
                out=buf[(ALIGN_FRAMES - lead) * FRAME_SIZE:])
    return buf
