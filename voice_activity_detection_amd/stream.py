"""End-to-end scoring of a device-resident audio stream.

``vad_stream`` is the device form of the reference's ``netvad``
(core/train.py:539-626): features for every frame, a posterior for every
stride-1 window of 30 frames, and a decision per window that the reference
aligns with frame ``i + 15`` (core/train.py:600-601, 624-626).  Windows are
never materialised: the model kernels read the packed (F,24) feature matrix
with a sliding index.
"""
import torch

from .common import FEATURES, FRAMES, SLICE_MAX
from .process_data import mfcc_delta_stream

DECISION_OFFSET = 15      # core/train.py:600


class StreamWorkspace:
    """Caller-owned output buffers, reusable across calls of the same size."""

    def __init__(self, n_frames, device):
        n_win = max(n_frames - FRAMES + 1, 0)
        self.n_frames = n_frames
        self.feat = torch.empty((n_frames, FEATURES), dtype=torch.float32, device=device)
        self.p = torch.empty((n_win, 2), dtype=torch.float32, device=device)
        self.decision = torch.empty((n_win,), dtype=torch.uint8, device=device)


def vad_stream(net, stream_dev, n_frames, slice_len=SLICE_MAX, threshold=None, workspace=None):
    """int16 stream (3 alignment rows + n_frames rows) -> (feat, p, decision), all on the device.

    ``p[i]`` is the softmax posterior of window i = frames i..i+29; ``p[i, 1]``
    is the speech score (core/train.py:440).  ``decision[i]`` is the argmax, or
    ``p[i,1] >= threshold`` when a threshold is given (core/train.py:486-487).
    """
    ws = workspace or StreamWorkspace(n_frames, stream_dev.device)
    if ws.n_frames != n_frames:
        raise ValueError('workspace was sized for %d frames' % ws.n_frames)
    mfcc_delta_stream(stream_dev, n_frames, slice_len, out=ws.feat)
    if n_frames >= FRAMES:
        net.score(ws.feat, window_stride=FEATURES, n_windows=n_frames - FRAMES + 1, threshold=threshold,
                  out=ws.p, decisions=ws.decision)
    return ws.feat, ws.p, ws.decision
