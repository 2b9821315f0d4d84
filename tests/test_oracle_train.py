"""The restated evaluation loops (oracle/train_ref.py) on the CPU: known answers for the alignment of netvad's
decisions and for the generator's partition arithmetic, and agreement of the two independent restatements of the
generator's draw (the oracle's flat functions and the product's DataGenerator) on the same seed."""
import numpy as np
import torch

from oracle import train_ref
from voice_activity_detection_amd.generator import DataGenerator


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    data = {'labels': (np.sin(np.arange(n) / 37.0) > 0).astype(np.int64)}
    for lvl in ('None', '-15', '-3'):
        data['frames-' + lvl] = np.arange(n, dtype=np.int64).reshape(n, 1)
        data['mfcc-' + lvl] = rng.normal(0, 8, size=(n, 12)).astype(np.float32)
        data['delta-' + lvl] = rng.normal(0, 2, size=(n, 12)).astype(np.float32)
    return data


class CentreFrameNet(torch.nn.Module):
    """Posterior from the window's frame 15 alone: speech iff its first coefficient is positive."""

    def forward(self, x):
        s = (x[:, 15, 0] > 0).float()
        return torch.stack((1 - s, s), 1)


def test_netvad_decision_i_belongs_to_frame_i_plus_15():
    n = 3000
    data = _data(n)
    for lvl in ('None', '-15', '-3'):
        data['mfcc-' + lvl][:, 0] = data['labels'] * 2.0 - 1.0
    raw, labels, out, post = train_ref.netvad(CentreFrameNet(), data, noise_level='-3', init_pos=50, length=700)
    assert len(raw) == len(labels) == len(out) == 700 - 30 and post.shape == (670, 2)
    np.testing.assert_array_equal(raw[:, 0], np.arange(50 + 15, 50 + 700 - 15))
    np.testing.assert_array_equal(out, labels)          # window i decides frame i + 15 (core/train.py:600, 624-626)


def test_test_partition_is_the_last_tenth():
    part = train_ref.test_partition({'labels': np.zeros(260000)})
    assert part['initial_pos'] == 234000 and part['test_size'] == 26000
    assert part['batch_count'] == int((int((26000 - 30) / 6) + 1) / 2048) == 2
    assert train_ref.test_partition({'labels': np.zeros(260000)}, size_limit=1000)['batch_count'] == 0


def test_generator_draws_agree_between_the_two_restatements():
    data = _data(30000, seed=4)
    g = DataGenerator(data)
    g.setup_generation(30, 6, 64)
    g.use_test_data()
    g.set_noise_level_db('-15')
    got = [g.get_batch(i) for i in range(g.batch_count)]
    part = train_ref.test_partition(data, batch_size=64)
    assert part['batch_count'] == g.batch_count > 2 and part['initial_pos'] == g.initial_pos
    for i, (x, y) in enumerate(got):
        xr, yr = train_ref.draw_batch(data, part, '-15', i)
        np.testing.assert_array_equal(x, xr)
        np.testing.assert_array_equal(y, yr)
