import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def vad():
    """The product package with its HIP library loaded; GPU tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    import voice_activity_detection_amd as pkg
    pkg.lib()          # raises if the in-tree libvad_amd.so is missing
    return pkg


@pytest.fixture
def f32_matrix_mode(vad):
    """The float32-MFMA form of the stride-1 GRU / LSTM kernels for the duration of a test: the mode in which
    stride-1 windows, dense batches and sampled windows give the same bits."""
    prev = vad.set_matrix_mode('f32')
    yield
    vad.set_matrix_mode(prev)
