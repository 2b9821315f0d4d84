"""Generate the model golden vectors by IMPORTING THE REFERENCE (build container only).

Run from the repository root:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        python3 tests/golden/make_model_fixtures.py

For each reference model variant it instantiates the reference class
(core/models.py), seeds it with the reference's SEED (core/train.py:56), gives
the BatchNorm layers non-trivial running statistics, switches to eval mode (as
core/train.py:410,558 do before inference) and stores

    tests/golden/model_<kind>.npz   state_dict arrays (float32) + 'y' = forward(x)
    tests/golden/model_inputs.npz   'x' (64, 30, 24) float32 shared input windows

Only data is written; no reference source travels with the fixtures.
"""
import os
import sys

import numpy as np
import torch

from core.models import Net, DenseNet, MODEL_STACK   # the reference

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.models_ref import randomize_batchnorm     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 1337

VARIANTS = {
    'gru_large': lambda: Net(large=True, lstm=False),
    'gru_small': lambda: Net(large=False, lstm=False),
    'lstm_large': lambda: Net(large=True, lstm=True),
    'lstm_small': lambda: Net(large=False, lstm=True),
    'densenet': lambda: DenseNet(large=False),
    'densenet_large': lambda: DenseNet(large=True),
}


def main():
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(64, 30, 24, generator=g)
    x[:, :, :12] *= 6.0          # lifted cepstra span roughly +-20
    x[:, :, 12:] *= 1.5          # deltas are an order of magnitude smaller
    x[0] = 0.0                   # the zero-padded windows netvad appends (core/train.py:590-591)
    np.savez_compressed(os.path.join(HERE, 'model_inputs.npz'), x=x.numpy())

    for kind, ctor in VARIANTS.items():
        torch.manual_seed(SEED)
        net = ctor()
        with torch.no_grad():
            randomize_batchnorm(net, torch.Generator().manual_seed(SEED + 1))
        net.eval()
        with torch.no_grad():
            y = net(x)
        arrays = {k: v.detach().numpy() for k, v in net.state_dict().items()}
        arrays['y'] = y.numpy()
        arrays['n_params'] = np.int64(sum(p.numel() for p in net.parameters()))
        np.savez_compressed(os.path.join(HERE, 'model_%s.npz' % kind), **arrays)
        print(kind, 'params', int(arrays['n_params']), 'y[1]', y[1].tolist())

    # registry names the boundary must keep (core/models.py:465-514)
    with open(os.path.join(HERE, 'model_stack_names.txt'), 'w') as f:
        for name, entry in MODEL_STACK.items():
            m = entry['model']
            f.write('%s %s large=%s lstm=%s\n' % (name, type(m).__name__, getattr(m, 'large', None),
                                                   getattr(m, 'lstm', None)))


if __name__ == '__main__':
    main()
