"""The training-data slice loop (core/process_data.py:76-158): the oracle's restatement and the product's slice
plan draw the same random numbers in the same order; the loop's per-slice semantics."""
import numpy as np

from oracle import mfcc_ref, overlay_ref, pipeline_ref, synth_ref
from voice_activity_detection_amd.process_data import training_slices


def _sets():
    speech = synth_ref.synth_frames(0, 400, seed=41)
    labels = (np.arange(400) // 25 % 2).astype(np.int8)
    noise = (synth_ref.synth_frames(5000, 600, seed=43) // 4).astype(np.int16)
    return speech, labels, noise


def test_slice_plan_matches_the_oracle_loop():
    speech, labels, noise = _sets()
    ref = pipeline_ref.process_training_data(speech, labels, noise)
    slices, total = training_slices(len(speech), np.random.RandomState(1337))
    assert slices == ref['slices']
    assert total == len(ref['labels']) == len(ref['mfcc-None'])
    lengths = [s[0] if s[0] == s[1] else s[1] - s[0] for s in slices]
    assert sum(lengths) == ref['count'] and min(lengths) >= 33
    # speech slices tile the speech set; silence adds about as many frames again
    cuts = sorted(s for s in slices if s[0] != s[1])
    assert cuts[0][0] == 0 and cuts[-1][1] == len(speech)
    assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))


def test_first_slice_is_one_add_noise_call_and_levels_accumulate():
    speech, labels, noise = _sets()
    ref = pipeline_ref.process_training_data(speech, labels, noise)
    n0 = ref['slices'][0]
    n0 = n0[0] if n0[0] == n0[1] else n0[1] - n0[0]
    f_none, f_15 = ref['frames-None'][:n0], ref['frames--15'][:n0]
    # level 'None' = noise + clean (no gain); recover the noise stretch from it for a silence or speech slice
    s0 = ref['slices'][0]
    clean = np.zeros((n0, 480), np.int16) if s0[0] == s0[1] else speech[s0[0]: s0[1]]
    noise_used = (f_none.astype(np.int64) - clean).astype(np.int16)      # nothing saturates at these levels
    # level '-15' mixes the scaled noise into level 'None''s frames, not into the clean speech (:148 overwrites `frames`)
    np.testing.assert_array_equal(f_15, overlay_ref.overlay(noise_used, f_none, -15).reshape(-1, 480))
    _, m, d = mfcc_ref.process_test_frames(f_15, np.zeros((3, 480), np.int16))
    np.testing.assert_array_equal(ref['mfcc--15'][:n0], m.astype(np.float32))
    np.testing.assert_array_equal(ref['delta--15'][:n0], d.astype(np.float32))
    assert not ref['frames-None'][ref['count']:].any()
