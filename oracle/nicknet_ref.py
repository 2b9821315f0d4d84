"""torch.nn CPU restatement of the reference's conv + GRU hybrid, ``NickNet`` (TEST INFRASTRUCTURE).

Restates ``GatedConv`` (core/models.py:180-202), the default (gated, non-residual, GRU, uni-directional,
fully connected head) configuration of ``NickNet`` (core/models.py:245-330) and of its recurrent part
``BiRNN`` (core/models.py:84-177) on stock ``torch.nn`` layers with the reference's ``state_dict`` key
names.

What is pinned and what is not.  ``NickNet.forward`` of the reference raises: ``BiRNN.forward`` flattens
all 30 GRU outputs (core/models.py:166 -> 870 columns) and then refuses them because ``embed`` takes
L*H = 58 (large) or 21 (small) columns (core/models.py:172-174).  The sub-modules below that line do run,
and ``trunk``, ``gru`` and ``embed`` here are pinned to vectors captured from them
(tests/golden/make_nicknet_fixtures.py, tests/test_oracle_nicknet.py).  The glue between GRU and
``embed`` is a choice: the head input is the final hidden state of every GRU layer,
``h_n.permute(1, 0, 2).reshape(B, L*H)`` -- the only tensor the GRU returns that has ``embed``'s width
(and what the class's own comment, "only use last output from forward sequence", describes for one layer).
PARITY OF THAT GLUE IS UNPINNED: nothing in the reference executes it.
"""
import torch
from torch import nn

FRAMES = 30      # core/common.py:22
FEATURES = 24    # core/common.py:23


def nicknet_dims(large):
    """(conv channels per layer, GRU hidden, GRU layers, embed widths) of the default gated configuration:
    core/models.py:264-275 (channels), :303 (hidden = last channels + 11 / + 5), :94 (layers),
    :99-128 (sz = 18; large: L*H -> 32 -> 18 -> 2, small: H -> 18 -> 2)."""
    if large:
        return (32, 28, 25, 18), 29, 2, (32, 18)
    return (20, 18, 16), 21, 1, (18,)


class _GatedConv(nn.Module):                   # core/models.py:180-202
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv1d(cin, cout, kernel_size=3, dilation=1, padding=1),
                                  nn.BatchNorm1d(cout), nn.Tanh())
        self.conv_gate = nn.Sequential(nn.Conv1d(cin, cout, kernel_size=3, dilation=1, padding=1),
                                       nn.BatchNorm1d(cout), nn.Sigmoid())

    def forward(self, x):
        return self.conv(x) * self.conv_gate(x)


class _Recurrent(nn.Module):                   # BiRNN, core/models.py:84-128 (lstm=False, bidir=False, fcl=True)
    def __init__(self, num_in, hidden, layers, widths):
        super().__init__()
        self.rnn = nn.GRU(num_in, hidden, num_layers=layers, bidirectional=False, batch_first=True)
        mods, cin = [], hidden * layers
        for w in widths:
            mods += [nn.Linear(cin, w), nn.BatchNorm1d(w), nn.Dropout(p=0.2), nn.ReLU()]
            cin = w
        mods.append(nn.Linear(cin, 2))
        self.embed = nn.Sequential(*mods)


class NickNetRef(nn.Module):
    def __init__(self, large=True):
        super().__init__()
        chans, hidden, layers, widths = nicknet_dims(large)
        self.large = large
        cin = FEATURES
        for i, c in enumerate(chans):
            setattr(self, 'conv%d' % (i + 1), _GatedConv(cin, c))
            cin = c
        self.n_conv = len(chans)
        self.rnn = _Recurrent(cin, hidden, layers, widths)

    def trunk(self, x):
        """(B, 30, 24) -> (B, C, 30): core/models.py:308-323."""
        x = x.permute(0, 2, 1)
        for i in range(self.n_conv):
            x = getattr(self, 'conv%d' % (i + 1))(x)
        return x

    def gru(self, trunk):
        """(B, C, 30) -> ((B, 30, H) outputs of the last layer, (L, B, H) final states): core/models.py:151,162."""
        return self.rnn.rnn(trunk.permute(0, 2, 1))

    def embed(self, v):
        """(B, L*H) -> (B, 2) logits: core/models.py:109-128 (eval mode: Dropout off, BatchNorm running statistics)."""
        return self.rnn.embed(v)

    def forward(self, x):
        _, hn = self.gru(self.trunk(x))
        head_in = hn.permute(1, 0, 2).reshape(x.size(0), -1)        # the unpinned glue (module docstring)
        return torch.softmax(self.embed(head_in), dim=1)            # core/models.py:330


def make_nicknet(large=True, seed=1337):
    """Seeded, eval-mode oracle model with non-trivial BatchNorm statistics."""
    from .models_ref import randomize_batchnorm
    torch.manual_seed(seed)
    m = NickNetRef(large=large)
    with torch.no_grad():
        randomize_batchnorm(m, torch.Generator().manual_seed(seed + 1))
    return m.eval()
