"""``Net`` / ``DenseNet`` / ``MODEL_STACK`` with the reference's constructor and
``forward`` signatures (core/models.py:9-81, 383-461, 465-514), evaluated by
the HIP kernels of libvad_amd.so.

The modules own ordinary ``torch.nn`` parameters under the reference's
``state_dict`` names, so a reference checkpoint's state dict loads unchanged;
the torch layers themselves are never called.  ``forward`` is inference only
(eval-mode BatchNorm, Dropout off, no autograd), which is how the reference
uses the models on this path (``net.eval()`` at core/train.py:410,558).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _lib
from .common import BATCH_SIZE, FEATURES, FRAMES

DECIDE_ARGMAX = 0
DECIDE_THRESHOLD = 1
MATRIX_MODES = {'f32': 0, 'bf16x6': 1, 'bf16x9': 2, 'bf16x8': 3, 'f16x6': 4, 'f16x4': 5}


def set_matrix_mode(mode):
    """Process-wide default (a model's ``matrix_mode`` attribute or ``score(matrix_mode=...)`` override it per
    model / per call): how the stride-1 GRU / LSTM kernels (and the wide GRU's stride-1 workspace path) form their
    float32 matrix products.
    'bf16x9' splits every float32 operand exactly into three bfloat16 and accumulates all nine exact partial
    products in float32 on the bf16 matrix pipe; 'bf16x8' leaves out the lo * lo product, under 2^-30
    of the product it belongs to; 'f32' is the float32 MFMA (a k-ordered FMA chain, bit-identical to the
    dense-batch and sampled-window paths); 'bf16x6' drops the three smallest partial products (up to 2^-23).
    'f16x6' (default) does the same with three float16 terms per operand (11 significant bits each): six partial products
    leave out under 2^-35 of a product; the state and the weights ride on exact power-of-two scales that keep them
    inside float16's exponent range.  'f16x4' keeps four of the products (to 2^-23).
    Returns the previous mode's name."""
    L = _lib.lib()
    if mode not in MATRIX_MODES:
        raise ValueError('matrix mode must be one of %s' % sorted(MATRIX_MODES))
    prev = L.vad_get_matrix_mode()
    _lib.check(L.vad_set_matrix_mode(MATRIX_MODES[mode]))
    return {v: k for k, v in MATRIX_MODES.items()}[prev]


def _cuda_device():
    if not torch.cuda.is_available():
        raise _lib.VadError('no HIP device visible: the models have no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


class _HipModule(nn.Module):
    """Shared plumbing: packed-weight cache and the windows -> posterior call."""

    def __init__(self):
        super().__init__()
        self._blob = None
        self._blob_key = None

    def _pack(self):                                   # -> numpy float32 blob
        raise NotImplementedError

    def _launch(self, L, x_ptr, n, stride, blob_ptr, p_ptr, d_ptr, mode, thr, stream, matrix_mode=None,
                workspace=None):
        raise NotImplementedError

    def _weights_on(self, device):
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.state_dict().values())
        if self._blob is None or self._blob_key != key:
            self._blob = torch.from_numpy(self._pack()).to(device)
            self._blob_key = key
        return self._blob

    def score(self, x, window_stride=None, n_windows=None, threshold=None, out=None, decisions=None,
              matrix_mode=None, workspace=None):
        """Posteriors for many windows without materialising them.

        ``x``  float32 CUDA tensor.  A (B,30,24) batch, or -- with
               ``window_stride`` given -- a packed (F,24) feature matrix from
               which window w reads rows starting at float offset
               ``w * window_stride`` (24 = the stride-1 sliding window of
               core/train.py:566-570).
        Returns ``(p (n,2) float32, decision (n,) uint8)``; the decision is the
        argmax (core/train.py:605-607) or ``p[:,1] >= threshold``
        (core/train.py:486-487).

        ``matrix_mode``  'f32', 'f16x6', 'f16x4', 'bf16x9', 'bf16x8' or 'bf16x6' for this call (GRU / LSTM over stride-1
               windows; see ``set_matrix_mode``).  None: the module's ``matrix_mode`` attribute, and if
               that is None too, the process-wide mode.
        ``workspace``    uint8 CUDA tensor for the stride-1 scratch of GRUs with more than 32 units
               (``Net.workspace_bytes``); None: a buffer the module keeps per (device, stream).
        """
        L = _lib.lib()
        matrix_mode = matrix_mode if matrix_mode is not None else getattr(self, 'matrix_mode', None)
        if matrix_mode is not None and matrix_mode not in MATRIX_MODES:
            raise ValueError('matrix mode must be one of %s' % sorted(MATRIX_MODES))
        if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError('x must be a contiguous float32 CUDA tensor')
        if window_stride is None:
            if x.dim() != 3 or x.size(1) != FRAMES or x.size(2) != FEATURES:
                raise ValueError('expected a (B,%d,%d) batch, got %s' % (FRAMES, FEATURES, tuple(x.shape)))
            window_stride, n = FRAMES * FEATURES, x.size(0)
        else:
            total = x.numel()
            n = 0 if total < FRAMES * FEATURES else (total - FRAMES * FEATURES) // window_stride + 1
            if n_windows is not None:
                if n_windows > n:
                    raise ValueError('n_windows %d exceeds the %d windows x can hold' % (n_windows, n))
                n = n_windows
        p = out if out is not None else torch.empty((n, 2), dtype=torch.float32, device=x.device)
        d = decisions if decisions is not None else torch.empty((n,), dtype=torch.uint8, device=x.device)
        if n == 0:
            return p, d
        blob = self._weights_on(x.device)
        mode = DECIDE_ARGMAX if threshold is None else DECIDE_THRESHOLD
        with torch.cuda.device(x.device):
            _lib.check(self._launch(L, ctypes.c_void_p(x.data_ptr()), n, window_stride,
                                    ctypes.c_void_p(blob.data_ptr()), ctypes.c_void_p(p.data_ptr()),
                                    ctypes.c_void_p(d.data_ptr()), mode,
                                    ctypes.c_float(0.0 if threshold is None else threshold),
                                    _lib.current_stream(), matrix_mode, workspace))
        return p, d

    def score_rows(self, feat, window_rows, threshold=None, out=None, decisions=None):
        """Posteriors of sampled windows: window w covers rows ``window_rows[w] .. +29`` of the packed
        (F,24) float32 CUDA feature matrix ``feat`` -- the windows DataGenerator.get_batch stacks into
        a batch (core/generator.py:143-150) -- without building the (B,30,24) tensor.
        ``window_rows`` is an int64 CUDA tensor or any integer sequence."""
        L = _lib.lib()
        if not feat.is_cuda or feat.dtype != torch.float32 or not feat.is_contiguous() or feat.dim() != 2 \
                or feat.size(1) != FEATURES:
            raise ValueError('feat must be a contiguous float32 CUDA tensor of shape (F,%d)' % FEATURES)
        if feat.size(0) < FRAMES:
            raise ValueError('feat holds %d rows, a window needs %d' % (feat.size(0), FRAMES))
        rows = torch.as_tensor(window_rows, dtype=torch.int64).to(feat.device).reshape(-1).contiguous()
        n = rows.numel()
        if n and (int(rows.min()) < 0 or int(rows.max()) > feat.size(0) - FRAMES):
            raise IndexError('window start out of range 0..%d' % (feat.size(0) - FRAMES))
        p = out if out is not None else torch.empty((n, 2), dtype=torch.float32, device=feat.device)
        d = decisions if decisions is not None else torch.empty((n,), dtype=torch.uint8, device=feat.device)
        if n == 0:
            return p, d
        blob = self._weights_on(feat.device)
        mode = DECIDE_ARGMAX if threshold is None else DECIDE_THRESHOLD
        with torch.cuda.device(feat.device):
            _lib.check(self._launch_rows(L, ctypes.c_void_p(feat.data_ptr()), feat.size(0),
                                         ctypes.c_void_p(rows.data_ptr()), n, ctypes.c_void_p(blob.data_ptr()),
                                         ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(d.data_ptr()), mode,
                                         ctypes.c_float(0.0 if threshold is None else threshold),
                                         _lib.current_stream()))
        return p, d

    def forward(self, x):
        # (Net overrides this for the other shapes core/models.py:44-66 accepts; the convolutional models only
        # work on 30-step windows in the reference too.)
        if x.dim() != 3 or x.size(1) != FRAMES or x.size(2) != FEATURES:
            raise RuntimeError('expected input of shape (batch, %d, %d), got %s'
                               % (FRAMES, FEATURES, tuple(x.shape)))
        x = x.detach().to(device=_cuda_device(), dtype=torch.float32).contiguous()   # core/models.py:50-51
        return self.score(x)[0]


class Net(_HipModule):
    """core/models.py:9-81: 1-layer LSTM or GRU (24 -> 30), flatten, lin1+ReLU+lin2 or lin, softmax."""

    def __init__(self, large=True, lstm=True, hidden=FRAMES, matrix_mode=None):
        # ``hidden`` is an extension: the reference hard-wires FRAMES = 30 (core/models.py:19-25); the
        # enlarged-GRU benchmark configuration uses 90.  ``matrix_mode`` (also an extension) fixes how this
        # model's stride-1 kernels form their float32 products; None follows the process-wide mode.
        super().__init__()
        if matrix_mode is not None and matrix_mode not in MATRIX_MODES:
            raise ValueError('matrix mode must be one of %s' % sorted(MATRIX_MODES))
        self.large = large
        self.lstm = lstm
        self.matrix_mode = matrix_mode
        self.relu = nn.ReLU()
        cls = nn.LSTM if lstm else nn.GRU
        self.rnn = cls(input_size=FEATURES, hidden_size=hidden, num_layers=1, batch_first=True)
        self.lin1 = nn.Linear(FRAMES * hidden, 26)
        self.lin2 = nn.Linear(26, 2)
        self.lin = nn.Linear(FRAMES * hidden, 2)
        self.softmax = nn.Softmax(dim=1)
        self._ws = {}

    def _pack(self):
        L = _lib.lib()
        H = self.rnn.hidden_size
        n = L.vad_net_blob_floats(int(self.lstm), int(self.large), H)
        if n == 0:
            raise _lib.VadError('hidden size %d is not supported (LSTM 1..32, GRU 1..96)' % H)
        blob = np.zeros(n, np.float32)
        a = {k: _np32(v) for k, v in self.state_dict().items()}
        hp = _lib.host_ptr
        _lib.check(L.vad_pack_net_weights(
            int(self.lstm), int(self.large), H,
            hp(a['rnn.weight_ih_l0']), hp(a['rnn.weight_hh_l0']), hp(a['rnn.bias_ih_l0']), hp(a['rnn.bias_hh_l0']),
            hp(a['lin1.weight']), hp(a['lin1.bias']), hp(a['lin2.weight']), hp(a['lin2.bias']),
            hp(a['lin.weight']), hp(a['lin.bias']), hp(blob), n))
        return blob

    # largest scratch the wide GRU's stride-1 path is given (bigger launches go in chunks)
    WORKSPACE_CAP = 16 << 30

    def workspace_bytes(self, n_windows):
        """Scratch the stride-1 path of a GRU with more than 32 units uses for ``n_windows`` windows (0 for the
        other models): the size of the ``workspace`` tensor ``score`` accepts."""
        if self.lstm or self.rnn.hidden_size <= 32:
            return 0
        return min(_lib.lib().vad_gru_workspace_bytes(self.rnn.hidden_size, n_windows), self.WORKSPACE_CAP)

    def reserve_workspace(self, n_windows, device=None):
        """Allocate the module-kept scratch for the CURRENT stream of ``device`` ahead of time, e.g. before
        a HIP graph capture (the first ``score`` on a stream allocates it otherwise)."""
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(dev):
            return self._workspace(self.workspace_bytes(n_windows), dev)

    def _workspace(self, need, dev):
        # One buffer per (device, stream): launches of this module on different streams must not share
        # scratch, and a buffer is only ever replaced by the stream that uses it (the allocator reuses the
        # memory in that stream's order, so work still in flight on it finishes first).
        if need == 0:
            return None
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws[key] = ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return ws

    def _launch(self, L, x_ptr, n, stride, blob_ptr, p_ptr, d_ptr, mode, thr, stream, matrix_mode=None,
                workspace=None):
        H = self.rnn.hidden_size
        mm = -1 if matrix_mode is None else MATRIX_MODES[matrix_mode]          # -1 = VAD_MATRIX_DEFAULT
        if self.lstm:
            return L.vad_lstm_forward_mode_f32(x_ptr, n, stride, blob_ptr, H, int(self.large), p_ptr, d_ptr, mode, thr,
                                               mm, stream)
        ws_ptr, ws_bytes = None, 0
        if H > 32 and stride == FEATURES:
            # projected feature rows of the wide GRU: the caller's workspace, or one this module keeps
            dev = torch.device('cuda', torch.cuda.current_device())
            if workspace is not None:
                if not workspace.is_cuda or workspace.dtype != torch.uint8 or not workspace.is_contiguous() \
                        or workspace.device != dev:
                    raise ValueError('workspace must be a contiguous uint8 CUDA tensor on the device of x')
                ws = workspace
            else:
                ws = self._workspace(self.workspace_bytes(n), dev)
            ws_ptr, ws_bytes = ctypes.c_void_p(ws.data_ptr()), ws.numel()
        return L.vad_gru_forward_mode_f32(x_ptr, n, stride, blob_ptr, H, int(self.large), p_ptr, d_ptr, mode, thr,
                                          mm, ws_ptr, ws_bytes, stream)

    def _launch_rows(self, L, f_ptr, n_rows, r_ptr, n, blob_ptr, p_ptr, d_ptr, mode, thr, stream):
        fn = L.vad_lstm_forward_rows_f32 if self.lstm else L.vad_gru_forward_rows_f32
        return fn(f_ptr, n_rows, r_ptr, n, blob_ptr, self.rnn.hidden_size, int(self.large), p_ptr, d_ptr, mode, thr,
                  stream)

    def forward(self, x):
        """core/models.py:39-81.  (B,30,24) windows take the window kernels; any other shape the reference accepts
        -- 2-D (A,24) = A one-step sequences, 1-D, or (B,T,24) with B*T a multiple of 30 -- is run as sequences
        whose hidden states are re-cut into rows of 900 exactly as the reference's view(-1, FRAMES ** 2) does, and
        raises RuntimeError where that view (or torch's input-size check) would."""
        if x.dim() == 2:                                           # core/models.py:44-47
            x = x.unsqueeze(1)
        elif x.dim() == 1:
            x = x.unsqueeze(0).unsqueeze(0)
        if x.dim() == 3 and x.size(1) == FRAMES and x.size(2) == FEATURES:
            return super().forward(x)
        if x.dim() != 3 or x.size(2) != FEATURES:
            raise RuntimeError('input.size(-1) must be equal to input_size. Expected %d, got %s'
                               % (FEATURES, tuple(x.shape)))
        H = self.rnn.hidden_size
        if H != FRAMES:
            raise RuntimeError('inputs other than (batch, %d, %d) windows need the reference\'s %d hidden units'
                               % (FRAMES, FEATURES, FRAMES))
        n_seq, seq_len = x.size(0), x.size(1)
        if (n_seq * seq_len * H) % (FRAMES * FRAMES):
            raise RuntimeError("shape '[-1, %d]' is invalid for input of size %d"
                               % (FRAMES * FRAMES, n_seq * seq_len * H))
        dev = _cuda_device()
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        sd = {k: v.detach().to(device=dev, dtype=torch.float32).contiguous() for k, v in self.state_dict().items()}
        head = ('lin1.weight', 'lin1.bias', 'lin2.weight', 'lin2.bias') if self.large else ('lin.weight', 'lin.bias')
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        head_ptrs = [ptr(sd[k]) for k in head] + [None] * (4 - len(head))
        rows = n_seq * seq_len // FRAMES
        hseq = torch.empty((max(n_seq * seq_len, 1), H), dtype=torch.float32, device=dev)
        p = torch.empty((rows, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().vad_net_forward_seq_f32(
                ptr(x), n_seq, seq_len, ptr(sd['rnn.weight_ih_l0']), ptr(sd['rnn.weight_hh_l0']),
                ptr(sd['rnn.bias_ih_l0']), ptr(sd['rnn.bias_hh_l0']), *head_ptrs, int(self.lstm), int(self.large),
                ptr(hseq), ptr(p), _lib.current_stream()))
        return p


class _DenseSingle(nn.Module):                 # parameter container, core/models.py:333-348
    def __init__(self, input_size, output_size, dropout, dilation, padding, kernel_size, stride):
        super().__init__()
        self.layer = nn.Sequential(
            nn.Conv1d(input_size, output_size, kernel_size=kernel_size, padding=padding, stride=stride,
                      dilation=dilation, bias=False),
            nn.BatchNorm1d(output_size), nn.LeakyReLU(), nn.Dropout(p=dropout))


class _DenseBlock(nn.Module):                  # core/models.py:351-364
    def __init__(self, input_size, n_layers, growth_rate, dropout, dilation, padding, kernel_size, stride):
        super().__init__()
        self.block = nn.Sequential(*[
            _DenseSingle(input_size + i * growth_rate, growth_rate, dropout, dilation, padding, kernel_size, stride)
            for i in range(n_layers)])


class _TransitionBlock(nn.Module):             # core/models.py:367-380
    def __init__(self, input_size, output_size, dropout):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Conv1d(input_size, output_size, kernel_size=1, stride=1, bias=False), nn.BatchNorm1d(output_size),
            nn.MaxPool1d(kernel_size=2, stride=2, padding=0), nn.LeakyReLU(), nn.Dropout(p=dropout))


class DenseNet(_HipModule):
    """core/models.py:383-461."""

    def __init__(self, large=False):
        super().__init__()
        self.large = large
        c0, nl, g, c1 = (48, 8, 4, 80) if large else (24, 6, 3, 42)
        dropout = 0.4
        self.cnn_in = nn.Sequential(
            nn.Conv1d(in_channels=24, out_channels=c0, kernel_size=6, stride=1, padding=0, dilation=4, bias=False),
            nn.BatchNorm1d(c0), nn.MaxPool1d(kernel_size=2, stride=2, padding=0), nn.LeakyReLU(),
            nn.Dropout(p=dropout))
        self.dense1 = _DenseBlock(input_size=c0, n_layers=nl, growth_rate=g, kernel_size=3, dilation=1, padding=1,
                                  stride=1, dropout=dropout)
        self.trans1 = _TransitionBlock(input_size=c1, output_size=c0, dropout=dropout)
        self.dense2 = _DenseBlock(input_size=c0, n_layers=nl, growth_rate=g, kernel_size=3, dilation=1, padding=1,
                                  stride=1, dropout=dropout)
        self.cnn_out = nn.Sequential(
            nn.Conv1d(in_channels=c1, out_channels=c1, kernel_size=1, stride=1, bias=False), nn.BatchNorm1d(c1),
            nn.MaxPool1d(kernel_size=2, stride=2, padding=0), nn.LeakyReLU(), nn.Dropout(p=dropout))
        self.out = nn.Linear(c1, 2, bias=False)

    def _pack(self):
        L = _lib.lib()
        arrays = [_np32(v) for k, v in self.state_dict().items() if not k.endswith('num_batches_tracked')]
        if len(arrays) != L.vad_densenet_tensor_count(int(self.large)):
            raise _lib.VadError('unexpected DenseNet tensor count %d' % len(arrays))
        ptrs = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        n = L.vad_densenet_blob_floats(int(self.large))
        blob = np.zeros(n, np.float32)
        _lib.check(L.vad_pack_densenet_weights(int(self.large), ptrs, len(arrays), _lib.host_ptr(blob), n))
        return blob

    def _launch(self, L, x_ptr, n, stride, blob_ptr, p_ptr, d_ptr, mode, thr, stream, matrix_mode=None,
                workspace=None):
        # DenseNet always computes on the float32 MFMA: the matrix mode does not apply
        return L.vad_densenet_forward_f32(x_ptr, n, stride, blob_ptr, int(self.large), p_ptr, d_ptr, mode, thr, stream)

    def _launch_rows(self, L, f_ptr, n_rows, r_ptr, n, blob_ptr, p_ptr, d_ptr, mode, thr, stream):
        return L.vad_densenet_forward_rows_f32(f_ptr, n_rows, r_ptr, n, blob_ptr, int(self.large), p_ptr, d_ptr, mode,
                                               thr, stream)


class _GatedConv(nn.Module):                   # parameter container, core/models.py:180-202
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size=3, dilation=1, padding=1),
                                  nn.BatchNorm1d(out_channels), nn.Tanh())
        self.conv_gate = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size=3, dilation=1, padding=1),
                                       nn.BatchNorm1d(out_channels), nn.Sigmoid())


class _BiRNN(nn.Module):                       # parameter container, core/models.py:84-128 (GRU, one direction, fcl)
    def __init__(self, num_in, num_hidden, large):
        super().__init__()
        layers = 2 if large else 1
        self.rnn = nn.GRU(num_in, num_hidden, num_layers=layers, bidirectional=False, batch_first=True)
        sz, embed_sz = 18, num_hidden * layers
        if large:
            self.embed = nn.Sequential(nn.Linear(embed_sz, sz + 14), nn.BatchNorm1d(sz + 14), nn.Dropout(p=0.2), nn.ReLU(),
                                       nn.Linear(sz + 14, sz), nn.BatchNorm1d(sz), nn.Dropout(p=0.2), nn.ReLU(),
                                       nn.Linear(sz, 2))
        else:
            self.embed = nn.Sequential(nn.Linear(embed_sz, sz), nn.BatchNorm1d(sz), nn.Dropout(p=0.2), nn.ReLU(),
                                       nn.Linear(sz, 2))


class NickNet(_HipModule):
    """core/models.py:245-330 in its default configuration (gated convolutions, no residual connections, GRU,
    one direction, fully connected head); the other constructor switches of the reference are refused.

    The reference's own ``forward`` raises between the GRU and ``embed`` (core/models.py:166-174: 870 columns where
    ``embed`` takes 58 / 21).  The head input here is the final hidden state of every GRU layer,
    ``h_n.permute(1, 0, 2).reshape(B, L*H)`` -- the one tensor the GRU returns with ``embed``'s width.  ``stages``
    returns the outputs of the sub-modules that do run in the reference, for checking them one by one."""

    def __init__(self, large=True, residual_connections=False, gated=True, lstm=False, fcl=True, bidir=False,
                 frames=FRAMES, features=FEATURES):
        super().__init__()
        if residual_connections or not gated or lstm or not fcl or bidir or frames != FRAMES or features != FEATURES:
            raise NotImplementedError('NickNet is built in its default configuration only: gated=True, '
                                      'residual_connections=False, lstm=False, fcl=True, bidir=False, 30 x 24 windows')
        self.large = large
        self.residual_connections = residual_connections
        chans = (32, 28, 25, 18) if large else (20, 18, 16)           # core/models.py:264-275
        cin = features
        for i, c in enumerate(chans):
            setattr(self, 'conv%d' % (i + 1), _GatedConv(cin, c))
            cin = c
        self.rnn = _BiRNN(cin, cin + 11 if large else cin + 5, large)  # core/models.py:303-304

    def _tensors(self):
        return [_np32(v) for k, v in self.state_dict().items() if not k.endswith('num_batches_tracked')]

    def _pack(self):
        L = _lib.lib()
        arrays = self._tensors()
        if len(arrays) != L.vad_nicknet_tensor_count(int(self.large)):
            raise _lib.VadError('unexpected NickNet tensor count %d' % len(arrays))
        ptrs = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        n = L.vad_nicknet_blob_floats(int(self.large))
        blob = np.zeros(n, np.float32)
        _lib.check(L.vad_pack_nicknet_weights(int(self.large), ptrs, len(arrays), _lib.host_ptr(blob), n))
        return blob

    def _launch(self, L, x_ptr, n, stride, blob_ptr, p_ptr, d_ptr, mode, thr, stream, matrix_mode=None,
                workspace=None):
        return L.vad_nicknet_forward_f32(x_ptr, n, stride, blob_ptr, int(self.large), p_ptr, d_ptr, mode, thr, stream)

    def _launch_rows(self, L, f_ptr, n_rows, r_ptr, n, blob_ptr, p_ptr, d_ptr, mode, thr, stream):
        return L.vad_nicknet_forward_rows_f32(f_ptr, n_rows, r_ptr, n, blob_ptr, int(self.large), p_ptr, d_ptr, mode,
                                              thr, stream)

    @property
    def _dims(self):
        gru = self.rnn.rnn
        return gru.input_size, gru.hidden_size, gru.num_layers

    def stages(self, x=None, trunk=None):
        """Outputs of the sub-modules for a (B,30,24) batch ``x`` -- or, with ``trunk`` (B,C,30) given instead,
        of the GRU and the head started from that trunk output: dict with ``trunk`` (B,C,30) (core/models.py:308-323),
        ``seq`` (B,30,H) and ``hn`` (L,B,H) (``net.rnn.rnn(...)``, core/models.py:162), ``logits`` and ``p`` (B,2)."""
        L = _lib.lib()
        if (x is None) == (trunk is None):
            raise ValueError('give exactly one of x and trunk')
        c, hid, layers = self._dims
        dev = _cuda_device()
        src = (x if x is not None else trunk).detach().to(device=dev, dtype=torch.float32).contiguous()
        if x is not None and (src.dim() != 3 or tuple(src.shape[1:]) != (FRAMES, FEATURES)):
            raise ValueError('x must be (B,%d,%d)' % (FRAMES, FEATURES))
        if trunk is not None and (src.dim() != 3 or tuple(src.shape[1:]) != (c, FRAMES)):
            raise ValueError('trunk must be (B,%d,%d)' % (c, FRAMES))
        b = src.size(0)
        out = {'seq': torch.empty((b, FRAMES, hid), device=dev), 'hn': torch.empty((layers, b, hid), device=dev),
               'logits': torch.empty((b, 2), device=dev), 'p': torch.empty((b, 2), device=dev)}
        if x is not None:
            out['trunk'] = torch.empty((b, c, FRAMES), device=dev)
        if b == 0:
            return out
        blob = self._weights_on(dev)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(dev):
            _lib.check(L.vad_nicknet_stages_f32(ptr(src if x is not None else None), b, FRAMES * FEATURES,
                                                ptr(src if trunk is not None else None), ptr(blob), int(self.large),
                                                ptr(out.get('trunk')), ptr(out['seq']), ptr(out['hn']),
                                                ptr(out['logits']), ptr(out['p']), _lib.current_stream()))
        return out

    def embed(self, v):
        """``net.rnn.embed(v)`` (core/models.py:109-128, eval mode) for rows ``v`` (B, L*H): the logits (B,2)."""
        L = _lib.lib()
        _, hid, layers = self._dims
        dev = _cuda_device()
        v = v.detach().to(device=dev, dtype=torch.float32).contiguous()
        if v.dim() != 2 or v.size(1) != hid * layers:
            raise ValueError('v must be (B,%d)' % (hid * layers))
        logits = torch.empty((v.size(0), 2), device=dev)
        if v.size(0):
            blob = self._weights_on(dev)
            with torch.cuda.device(dev):
                _lib.check(L.vad_nicknet_embed_f32(ctypes.c_void_p(v.data_ptr()), v.size(0), ctypes.c_void_p(blob.data_ptr()),
                                                   int(self.large), ctypes.c_void_p(logits.data_ptr()), None,
                                                   _lib.current_stream()))
        return logits


def _model_stack():
    # names, descriptions and training kwargs as in core/models.py:465-514
    return {
        'net': {'desc': "LSTM, small, γ = 0", 'model': Net(large=False), 'kwargs': {'gamma': 0}},
        'net_large': {'desc': "LSTM, large, γ = 2", 'model': Net(), 'kwargs': {'gamma': 2}},
        'densenet': {'desc': "DenseNet, small, γ = 2", 'model': DenseNet(large=False),
                     'kwargs': {'use_adam': False, 'lr': 1, 'momentum': 0.7, 'gamma': 2}},
        'densenet_large': {'desc': "DenseNet, large, γ = 2", 'model': DenseNet(large=True),
                           'kwargs': {'use_adam': False, 'lr': 1, 'momentum': 0.7, 'gamma': 2}},
    }


MODEL_STACK = _model_stack()
