"""Golden vectors for Net.forward on inputs that are NOT (B,30,24) windows, made by IMPORTING THE REFERENCE (build
container only).  Run from the repository root, after make_model_fixtures.py:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        python3 tests/golden/make_seq_fixtures.py

The reference widens 2-D inputs to one-step sequences (core/models.py:44-47) and re-cuts all hidden states into rows
of 900 (core/models.py:66).  For each of its four Net variants, with the weights of tests/golden/model_<kind>.npz,
this stores the reference's outputs for

    x2d   (60, 24)      60 one-step sequences        -> (2, 2)
    x15   (4, 15, 24)   4 sequences of 15 steps      -> (2, 2)
    x45   (2, 45, 24)   2 sequences of 45 steps      -> (3, 2)
    x1    (30, 1, 24)   the 3-D form of a 2-D input  -> (1, 2)

in tests/golden/model_seq_shapes.npz, and the RuntimeError the reference raises for a 1-D input and for (1, 7, 24).
Only data is written.
"""
import os

import numpy as np
import torch

from core.models import Net   # the reference

HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = {'gru_large': (True, False), 'gru_small': (False, False), 'lstm_large': (True, True), 'lstm_small': (False, True)}


def main():
    g = torch.Generator().manual_seed(4242)
    inputs = {'x2d': torch.randn(60, 24, generator=g) * 4.0, 'x15': torch.randn(4, 15, 24, generator=g) * 4.0,
              'x45': torch.randn(2, 45, 24, generator=g) * 4.0, 'x1': torch.randn(30, 1, 24, generator=g) * 4.0}
    out = {k: v.numpy() for k, v in inputs.items()}
    for kind, (large, lstm) in KINDS.items():
        fx = np.load(os.path.join(HERE, 'model_%s.npz' % kind))
        net = Net(large=large, lstm=lstm)
        net.load_state_dict({k: torch.from_numpy(fx[k]) for k in net.state_dict().keys()})
        net.eval()
        with torch.no_grad():
            for name, x in inputs.items():
                out['%s.%s' % (kind, name)] = net(x).numpy()
            for name, bad in (('1d', torch.zeros(24)), ('7', torch.zeros(1, 7, 24))):
                try:
                    net(bad)
                    raised = ''
                except RuntimeError as e:
                    raised = str(e).splitlines()[0]
                out['%s.raises_%s' % (kind, name)] = np.array(raised)
        print(kind, {k: out['%s.%s' % (kind, k)].shape for k in inputs}, str(out['%s.raises_1d' % kind]))
    np.savez_compressed(os.path.join(HERE, 'model_seq_shapes.npz'), **out)


if __name__ == '__main__':
    main()
