"""DataGenerator mirror (core/generator.py:13-165): split arithmetic, the order of the RNG draws, batch layout."""
import numpy as np
import pytest

from voice_activity_detection_amd.generator import DataGenerator


def _data(n=3000, seed=0):
    rng = np.random.default_rng(seed)
    labels = (np.sin(np.arange(n) / 40.0) > 0).astype(np.int64)          # alternating speech / non-speech runs
    d = {'labels': labels}
    for lvl in ('None', '-15', '-3'):
        d['frames-' + lvl] = rng.integers(-100, 100, size=(n, 480)).astype(np.int16)
        d['mfcc-' + lvl] = rng.normal(size=(n, 12)).astype(np.float32)
        d['delta-' + lvl] = rng.normal(size=(n, 12)).astype(np.float32)
    return d


def test_splits_and_batch_counts():
    g = DataGenerator(_data())
    g.setup_generation(30, 6, 64)
    assert (g.train_size, g.val_size, g.test_size) == (2400, 300, 300)
    g.set_noise_level_db('-3')
    assert g.data_mode == 0 and g.initial_pos == 0 and g.batch_count == int((int((2400 - 30) / 6) + 1) / 64)
    g.use_test_data()
    assert g.initial_pos == 2700 and g.batch_count == int((int((300 - 30) / 6) + 1) / 64)
    with pytest.raises(Exception):
        g.set_noise_level_db('-7')


def test_get_batch_follows_the_reference_draw_order():
    d = _data()
    g = DataGenerator(d)
    g.setup_generation(30, 6, 64)
    g.set_noise_level_db('-15')
    x, y = g.get_batch(2)
    # the same sequence of global-RNG calls, written out
    np.random.seed(1337)
    np.random.shuffle(np.arange(3000))
    pos, span = (64 * 2) * 6, 30 + 6 * 64
    lab = d['labels'][pos: pos + span]
    c0, c1 = np.where(lab == 0)[0], np.where(lab == 1)[0]
    s0 = np.random.choice(c0, 32, replace=True)
    s1 = np.random.choice(c1, 32, replace=True)
    idx = np.hstack((s0, s1))
    np.random.shuffle(idx)
    idx = [i for i in idx if i + 30 <= span]
    assert x.shape == (len(idx), 30, 24) and y.shape == (len(idx),)
    for b, i in enumerate(idx):
        np.testing.assert_array_equal(x[b, :, :12], d['mfcc--15'][pos + i: pos + i + 30])
        np.testing.assert_array_equal(x[b, :, 12:], d['delta--15'][pos + i: pos + i + 30])
        assert y[b] == lab[i]
    assert set(np.unique(y)) == {0, 1}


def test_rows_form_draws_the_same_windows():
    d = _data(10000)
    a, b = DataGenerator(d), None
    a.setup_generation(30, 6, 64)
    a.set_noise_level_db('None')
    a.use_test_data()
    x, y = a.get_batch(0)
    b = DataGenerator(d)
    b.setup_generation(30, 6, 64)
    b.set_noise_level_db('None')
    b.use_test_data()
    rows, y2 = b.get_batch_rows(0)
    np.testing.assert_array_equal(y, y2)
    feat = np.hstack((d['mfcc-None'], d['delta-None']))
    np.testing.assert_array_equal(x, np.stack([feat[r: r + 30] for r in rows]))


def test_skip_single_class():
    d = _data()
    d['labels'] = np.zeros_like(d['labels'])
    g = DataGenerator(d)
    g.setup_generation(30, 6, 8)
    g.set_noise_level_db('-3')
    assert g.get_batch(0, skip_single_class=True) == ([], [])
    rows, y = g.get_batch_rows(0, skip_single_class=True)
    assert rows.size == 0 and y.size == 0


def test_slice_list_follows_the_reference_loop():
    """core/process_data.py:190-195 and the skip rule at :214-217, written out."""
    from voice_activity_detection_amd.process_data import recording_slices
    for l in (0, 33, 34, 166, 167, 199, 200, 332, 333, 1000, 1029, 1030):
        pos, ref = 0, []
        while pos + 33 < l:
            s = (pos, pos + 166)
            ref.append(s)
            pos = s[1]
        ref = [s for s in ref if (np.array(s) < l).all()]
        assert recording_slices(l) == ref
    assert recording_slices(167) == [(0, 166)] and recording_slices(166) == []
