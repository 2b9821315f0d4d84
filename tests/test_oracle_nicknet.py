"""oracle/nicknet_ref.py against vectors captured from the reference's own NickNet sub-modules
(tests/golden/make_nicknet_fixtures.py): the conv trunk, the GRU and the embed MLP each reproduce the
reference bit for bit on the CPU.  The whole-model output has no reference counterpart (the reference's
forward raises); `y_glue` is the reference's sub-modules composed with the documented head input."""
import os

import numpy as np
import pytest
import torch

from oracle import nicknet_ref
from oracle.models_ref import load_state_dict_numpy


def _load(golden_dir, size):
    fx = np.load(os.path.join(golden_dir, 'nicknet_%s.npz' % size))
    x = torch.from_numpy(np.load(os.path.join(golden_dir, 'model_inputs.npz'))['x'])
    net = load_state_dict_numpy(nicknet_ref.NickNetRef(large=(size == 'large')), fx)
    return fx, x, net


@pytest.mark.parametrize('size', ['large', 'small'])
def test_state_dict_keys_and_parameter_count(golden_dir, size):
    fx, _, net = _load(golden_dir, size)
    extra = {'trunk', 'seq', 'hn', 'v', 'ev', 'y_glue', 'n_params'}
    assert set(net.state_dict().keys()) == set(fx.files) - extra
    assert sum(p.numel() for p in net.parameters()) == int(fx['n_params']) == (29605 if size == 'large' else 10019)


@pytest.mark.parametrize('size', ['large', 'small'])
def test_submodules_reproduce_the_reference(golden_dir, size):
    fx, x, net = _load(golden_dir, size)
    with torch.no_grad():
        trunk = net.trunk(x)
        seq, hn = net.gru(torch.from_numpy(fx['trunk']))
        ev = net.embed(torch.from_numpy(fx['v']))
        y = net(x)
    np.testing.assert_array_equal(trunk.numpy(), fx['trunk'])
    np.testing.assert_array_equal(seq.numpy(), fx['seq'])
    np.testing.assert_array_equal(hn.numpy(), fx['hn'])
    np.testing.assert_array_equal(ev.numpy(), fx['ev'])
    np.testing.assert_array_equal(y.numpy(), fx['y_glue'])
    assert np.allclose(y.sum(1).numpy(), 1.0, atol=1e-6)


def test_seeded_constructor_matches_the_fixture_weights(golden_dir):
    """make_nicknet() draws the same weights as the reference under the reference's seed (same module
    construction order), so GPU tests can build the oracle without the fixture file."""
    for size in ('large', 'small'):
        fx, _, _ = _load(golden_dir, size)
        net = nicknet_ref.make_nicknet(large=(size == 'large'))
        for k, v in net.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), fx[k], err_msg=k)
