"""The packed NickNet weight blob and the kernel's lane algebra, checked on the CPU: a numpy restatement of the
data flow of csrc/nicknet.hip (tools/nicknet_lane_model.py) evaluates the blob that vad_pack_nicknet_weights
writes and must reproduce the vectors captured from the reference's sub-modules."""
import os

import numpy as np
import pytest
import torch

from tools import nicknet_lane_model as lane_model


@pytest.mark.parametrize('size', ['large', 'small'])
def test_packed_blob_reproduces_the_reference_submodules(golden_dir, size):
    import voice_activity_detection_amd as pkg
    from oracle.models_ref import load_state_dict_numpy
    large = size == 'large'
    fx = np.load(os.path.join(golden_dir, 'nicknet_%s.npz' % size))
    x = np.load(os.path.join(golden_dir, 'model_inputs.npz'))['x'][:24]
    net = load_state_dict_numpy(pkg.NickNet(large=large), fx)
    blob = net._pack()
    assert blob.size == pkg.lib().vad_nicknet_blob_floats(int(large))
    got = lane_model.forward(blob, x, large)
    assert np.abs(got['trunk'] - fx['trunk'][:24]).max() <= 1e-5
    assert np.abs(got['seq'] - fx['seq'][:24]).max() <= 1e-5
    assert np.abs(got['hn'] - fx['hn'][:, :24]).max() <= 1e-5
    assert np.abs(got['p'] - fx['y_glue'][:24]).max() <= 1e-5
    ev = lane_model.embed(blob, fx['v'].astype(np.float64), large)
    assert np.abs(ev - fx['ev']).max() <= 2e-5 * max(1.0, np.abs(fx['ev']).max())


def test_other_configurations_are_refused():
    import voice_activity_detection_amd as pkg
    for kw in ({'gated': False}, {'residual_connections': True}, {'lstm': True}, {'bidir': True}, {'fcl': False}):
        with pytest.raises(NotImplementedError):
            pkg.NickNet(**kw)
