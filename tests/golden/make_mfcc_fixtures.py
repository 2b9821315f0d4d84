"""Golden vectors of the MFCC/delta front end: (169, 480) int16 aligned rows -> (166, 12) MFCC + (166, 12) delta.

PARITY UNPINNED: python_speech_features is not available offline and the reference holds no vectors
for this stage, so these come from this repository's own float64 restatement (oracle/mfcc_ref.py)
and pin it against regressions only.  Slice 0: synthetic stream (silence / noise / tone segments);
slice 1: the same kind of stream with frames zeroed the way the reference embeds silent frames
(core/process_data.py:127).

    python tests/golden/make_mfcc_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mfcc_ref, synth_ref  # noqa: E402

if __name__ == '__main__':
    a = synth_ref.synth_frames(0, 169, seed=1337)
    b = synth_ref.synth_frames(40000, 169, seed=7).copy()
    b[20:31] = 0                                   # an embedded run of silent frames
    b[100] = 0
    rows = np.stack([a, b])
    mf, de = [], []
    for s in range(2):
        _, m, d = mfcc_ref.process_test_frames(rows[s, 3:], rows[s, :3])
        mf.append(m)
        de.append(d)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mfcc_slices.npz')
    np.savez_compressed(out, rows=rows, mfcc=np.stack(mf), delta=np.stack(de))
    print(out, os.path.getsize(out), 'bytes')
