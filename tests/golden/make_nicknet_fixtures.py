"""Generate the NickNet golden vectors by IMPORTING THE REFERENCE (build container only).

Run from the repository root:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        python3 tests/golden/make_nicknet_fixtures.py

``NickNet.forward`` of the reference raises (core/models.py:166-174: ``BiRNN.forward`` flattens all 30
GRU outputs into 870 columns where its ``embed`` takes 58), so there is no whole-model output to capture.
Everything below that line runs, and is captured sub-module by sub-module for ``NickNet(large=True)`` and
``NickNet(large=False)`` (seeded with the reference's SEED, BatchNorm statistics randomised, eval mode):

    trunk   conv1 .. conv4 on x.permute(0, 2, 1)          core/models.py:308-323   (B, C, 30)
    seq, hn net.rnn.rnn(trunk.permute(0, 2, 1))           core/models.py:151,162   (B, 30, H), (L, B, H)
    v, ev   net.rnn.embed(v) for a seeded v               core/models.py:109-128   (B, L*H) -> (B, 2)
    y_glue  softmax(embed(hn.permute(1, 0, 2).reshape(B, -1)))  -- the reference's own sub-modules composed
            with the one reading of the head input that has embed's width (L*H columns: the final hidden
            state of every GRU layer); this composition is NOT something the reference executes.

Stored as tests/golden/nicknet_<large|small>.npz next to the state_dict arrays; inputs are the shared
tests/golden/model_inputs.npz windows.  Only data is written.
"""
import os
import sys

import numpy as np
import torch

from core.models import NickNet   # the reference

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.models_ref import randomize_batchnorm     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 1337


def main():
    x = torch.from_numpy(np.load(os.path.join(HERE, 'model_inputs.npz'))['x'])
    for large in (True, False):
        torch.manual_seed(SEED)
        net = NickNet(large=large)
        with torch.no_grad():
            randomize_batchnorm(net, torch.Generator().manual_seed(SEED + 1))
        net.eval()
        with torch.no_grad():
            t = net.conv1(x.permute(0, 2, 1))
            t = net.conv3(net.conv2(t))
            if large:
                t = net.conv4(t)
            seq, hn = net.rnn.rnn(t.permute(0, 2, 1))
            width = net.rnn.embed[0].in_features
            v = torch.randn(64, width, generator=torch.Generator().manual_seed(SEED + 2))
            ev = net.rnn.embed(v)
            head_in = hn.permute(1, 0, 2).reshape(x.size(0), -1)
            assert head_in.shape[1] == width
            y_glue = torch.softmax(net.rnn.embed(head_in), dim=1)
        arrays = {k: v_.detach().numpy() for k, v_ in net.state_dict().items()}
        arrays.update(trunk=t.numpy(), seq=seq.numpy(), hn=hn.numpy(), v=v.numpy(), ev=ev.numpy(),
                      y_glue=y_glue.numpy(), n_params=np.int64(sum(p.numel() for p in net.parameters())))
        name = 'nicknet_%s.npz' % ('large' if large else 'small')
        np.savez_compressed(os.path.join(HERE, name), **arrays)
        print(name, 'params', int(arrays['n_params']), 'trunk', tuple(t.shape), 'seq', tuple(seq.shape), 'hn',
              tuple(hn.shape), 'embed in', width, 'y_glue[1]', y_glue[1].tolist())


if __name__ == '__main__':
    main()
