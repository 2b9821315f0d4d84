"""Pin oracle/models_ref.py to vectors captured from the reference classes.

tests/golden/model_*.npz were written by tests/golden/make_model_fixtures.py,
which imports core/models.py of the reference; here the oracle's own torch.nn
restatement loads the same state dict and must reproduce the reference output.
"""
import os

import numpy as np
import pytest
import torch

from oracle import models_ref

KINDS = ['gru_large', 'gru_small', 'lstm_large', 'lstm_small', 'densenet', 'densenet_large']
N_PARAMS = {'gru_large': 30322, 'gru_small': 30322, 'lstm_large': 32002, 'lstm_small': 32002,
            'densenet': 9966, 'densenet_large': 29696}


def _load(golden_dir, kind):
    z = np.load(os.path.join(golden_dir, 'model_%s.npz' % kind))
    x = np.load(os.path.join(golden_dir, 'model_inputs.npz'))['x']
    return z, x


@pytest.mark.parametrize('kind', KINDS)
def test_oracle_reproduces_reference_forward(golden_dir, kind):
    z, x = _load(golden_dir, kind)
    net = models_ref.make_model(kind)
    models_ref.load_state_dict_numpy(net, z)
    with torch.no_grad():
        y = net(torch.from_numpy(x)).numpy()
    assert y.shape == (64, 2)
    # same torch build, same layers, same weights: bit-for-bit on this machine,
    # 1e-6 leaves room for a different CPU's GEMM blocking on the GPU box.
    np.testing.assert_allclose(y, z['y'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(y.sum(1), 1.0, atol=1e-6)


@pytest.mark.parametrize('kind', KINDS)
def test_param_counts_match_reference(golden_dir, kind):
    z, _ = _load(golden_dir, kind)
    net = models_ref.make_model(kind)
    assert sum(p.numel() for p in net.parameters()) == N_PARAMS[kind] == int(z['n_params'])
    assert set(net.state_dict().keys()) == set(k for k in z.files if k not in ('y', 'n_params'))


def test_same_seed_gives_reference_weights(golden_dir):
    """Construction order matches the reference, so SEED 1337 yields identical weights."""
    z, _ = _load(golden_dir, 'gru_large')
    net = models_ref.make_model('gru_large')
    for k, v in net.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), z[k])


def test_model_stack_names(golden_dir):
    with open(os.path.join(golden_dir, 'model_stack_names.txt')) as f:
        names = [line.split()[0] for line in f]
    assert names == ['net', 'net_large', 'densenet', 'densenet_large']


SEQ_INPUTS = ['x2d', 'x15', 'x45', 'x1']


@pytest.mark.parametrize('kind', KINDS[:4])
def test_oracle_reproduces_reference_on_other_input_shapes(golden_dir, kind):
    """Inputs that are not (B,30,24) windows (tests/golden/make_seq_fixtures.py): 2-D rows become one-step sequences
    (core/models.py:44-47) and all hidden states are re-cut into rows of 900 (core/models.py:66); a 1-D input and a
    sequence count the cut does not divide raise, as in the reference."""
    z, _ = _load(golden_dir, kind)
    fx = np.load(os.path.join(golden_dir, 'model_seq_shapes.npz'))
    net = models_ref.make_model(kind)
    models_ref.load_state_dict_numpy(net, z)
    with torch.no_grad():
        for name in SEQ_INPUTS:
            y = net(torch.from_numpy(fx[name])).numpy()
            np.testing.assert_allclose(y, fx['%s.%s' % (kind, name)], rtol=0, atol=1e-6, err_msg=name)
        for bad in (torch.zeros(24), torch.zeros(1, 7, 24)):
            with pytest.raises(RuntimeError, match='invalid for input of size'):
                net(bad)
    assert 'invalid for input of size 30' in str(fx['%s.raises_1d' % kind])
    assert 'invalid for input of size 210' in str(fx['%s.raises_7' % kind])
