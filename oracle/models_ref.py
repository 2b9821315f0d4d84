"""torch.nn CPU restatement of the reference models (TEST INFRASTRUCTURE).

Restates ``Net`` (core/models.py:9-81) and ``DenseNet`` with its building
blocks (core/models.py:333-461) on top of stock ``torch.nn`` layers, with the
same ``state_dict`` key names, so that a state dict dumped from the reference
loads unchanged.  Pinned by tests/test_oracle_models.py against input/output
vectors captured by importing the reference (tests/golden/make_model_fixtures.py).

``NetRef`` additionally takes ``hidden`` so that the enlarged-GRU benchmark
configuration (hidden 90; no class in the reference) has a CPU oracle too;
``hidden=30`` is the reference architecture.
"""
import numpy as np
import torch
from torch import nn

FRAMES = 30      # core/common.py:22
FEATURES = 24    # core/common.py:23


class NetRef(nn.Module):
    """core/models.py:9-81.  All three Linear layers exist whatever ``large`` is."""

    def __init__(self, large=True, lstm=True, hidden=FRAMES):
        super().__init__()
        self.large, self.lstm, self.hidden = large, lstm, hidden
        cls = nn.LSTM if lstm else nn.GRU
        self.rnn = cls(input_size=FEATURES, hidden_size=hidden, num_layers=1, batch_first=True)
        self.lin1 = nn.Linear(FRAMES * hidden, 26)
        self.lin2 = nn.Linear(26, 2)
        self.lin = nn.Linear(FRAMES * hidden, 2)

    def forward(self, x):
        if x.dim() == 2:                       # core/models.py:44-47
            x = x.unsqueeze(1)
        elif x.dim() == 1:
            x = x.unsqueeze(0).unsqueeze(0)
        if self.lstm:
            z = torch.zeros(1, x.size(0), self.hidden, dtype=x.dtype)
            x, _ = self.rnn(x, (z, z.clone()))
        else:
            x, _ = self.rnn(x)
        x = x.contiguous().view(-1, FRAMES * self.hidden)
        if self.large:
            x = self.lin2(torch.relu(self.lin1(x)))
        else:
            x = self.lin(x)
        return torch.softmax(x, dim=1)


class _DenseSingle(nn.Module):                 # core/models.py:333-348
    def __init__(self, cin, growth):
        super().__init__()
        self.layer = nn.Sequential(
            nn.Conv1d(cin, growth, kernel_size=3, padding=1, stride=1, dilation=1, bias=False),
            nn.BatchNorm1d(growth), nn.LeakyReLU(), nn.Dropout(p=0.4))

    def forward(self, x):
        return torch.cat([x, self.layer(x)], 1)


class _DenseBlock(nn.Module):                  # core/models.py:351-364
    def __init__(self, cin, n_layers, growth):
        super().__init__()
        self.block = nn.Sequential(*[_DenseSingle(cin + i * growth, growth) for i in range(n_layers)])

    def forward(self, x):
        return self.block(x)


class _Transition(nn.Module):                  # core/models.py:367-380
    def __init__(self, cin, cout):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Conv1d(cin, cout, kernel_size=1, stride=1, bias=False), nn.BatchNorm1d(cout),
            nn.MaxPool1d(kernel_size=2, stride=2, padding=0), nn.LeakyReLU(), nn.Dropout(p=0.4))

    def forward(self, x):
        return self.layers(x)


def densenet_dims(large):
    """(stem channels, dense layers per block, growth rate, channels after a block)."""
    return (48, 8, 4, 80) if large else (24, 6, 3, 42)     # core/models.py:387-447


class DenseNetRef(nn.Module):
    """core/models.py:383-461 (conv -> BN -> MaxPool -> LeakyReLU ordering kept)."""

    def __init__(self, large=False):
        super().__init__()
        c0, nl, g, c1 = densenet_dims(large)
        self.large = large
        self.cnn_in = nn.Sequential(
            nn.Conv1d(24, c0, kernel_size=6, stride=1, padding=0, dilation=4, bias=False),
            nn.BatchNorm1d(c0), nn.MaxPool1d(kernel_size=2, stride=2, padding=0),
            nn.LeakyReLU(), nn.Dropout(p=0.4))
        self.dense1 = _DenseBlock(c0, nl, g)
        self.trans1 = _Transition(c1, c0)
        self.dense2 = _DenseBlock(c0, nl, g)
        self.cnn_out = nn.Sequential(
            nn.Conv1d(c1, c1, kernel_size=1, stride=1, bias=False), nn.BatchNorm1d(c1),
            nn.MaxPool1d(kernel_size=2, stride=2, padding=0), nn.LeakyReLU(), nn.Dropout(p=0.4))
        self.out = nn.Linear(c1, 2, bias=False)

    def forward(self, x):
        x = x.permute(0, 2, 1)
        x = self.cnn_out(self.dense2(self.trans1(self.dense1(self.cnn_in(x)))))
        return torch.softmax(self.out(x.view(x.size(0), -1)), dim=1)


def randomize_batchnorm(module, generator):
    """Give every BatchNorm1d non-trivial eval-mode statistics (fresh BN is ~identity)."""
    for m in module.modules():
        if isinstance(m, nn.BatchNorm1d):
            n = m.num_features
            m.running_mean.copy_(torch.randn(n, generator=generator) * 0.5)
            m.running_var.copy_(torch.rand(n, generator=generator) * 1.5 + 0.5)
            m.weight.data.copy_(torch.rand(n, generator=generator) + 0.5)
            m.bias.data.copy_(torch.randn(n, generator=generator) * 0.3)


def make_model(kind, seed=1337, hidden=FRAMES):
    """Seeded, eval-mode oracle model.  kind: gru_large, gru_small, lstm_large,
    lstm_small, densenet, densenet_large."""
    torch.manual_seed(seed)                    # the reference's SEED, core/train.py:56
    if kind.startswith('densenet'):
        m = DenseNetRef(large=kind.endswith('large'))
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            randomize_batchnorm(m, g)
    else:
        rnn, size = kind.split('_')
        m = NetRef(large=(size == 'large'), lstm=(rnn == 'lstm'), hidden=hidden)
    return m.eval()


def state_dict_numpy(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def load_state_dict_numpy(module, arrays):
    sd = {k: torch.from_numpy(np.asarray(arrays[k])) for k in module.state_dict().keys()}
    module.load_state_dict(sd)
    return module.eval()
