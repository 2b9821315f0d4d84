"""Accuracy margins of the HIP path against the CPU oracle on a longer synthetic sample (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voice_activity_detection_amd as pkg
from oracle import models_ref, pipeline_ref, synth_ref

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
host = synth_ref.synth_frames(0, n, seed=1337)
stream = pkg.synth_stream(0, n, seed=1337)
feat = pkg.mfcc_delta_stream(stream, n, 166).cpu().numpy().astype(np.float64)
ref = pipeline_ref.features_sliced(host, 166).astype(np.float64)       # float32-stored oracle features
err = np.abs(feat - ref) / np.maximum(np.abs(ref), 1.0)
for name, sl in (('mfcc', slice(0, 12)), ('delta', slice(12, 24))):
    e = err[:, sl]
    print('%-5s scaled err: max %.3e  p99.9 %.3e  median %.3e   (|ref| max %.1f)'
          % (name, e.max(), np.quantile(e, 0.999), np.median(e), np.abs(ref[:, sl]).max()))
silent = (np.abs(host).max(1) == 0).astype(int)
run5 = np.zeros(n, bool)                       # frame t and its 4 predecessors are all silent
run5[4:] = np.convolve(silent, np.ones(5, int), 'valid') == 5
print('frames whose whole analysis window is silent: %d of %d; max |mfcc| there: %.3e'
      % (run5.sum(), n, np.abs(feat[run5][:, :12]).max() if run5.any() else 0.0))
xw = pipeline_ref.windows(ref.astype(np.float32))
for kind in ('gru_large', 'gru_small', 'lstm_large', 'lstm_small', 'densenet', 'densenet_large'):
    r = models_ref.make_model(kind)
    net = pkg.DenseNet(large=kind.endswith('large')) if kind.startswith('densenet') else \
        pkg.Net(large=kind.endswith('large'), lstm=kind.startswith('lstm'))
    net.load_state_dict(r.state_dict())
    y = pipeline_ref.score_windows(r, xw[:20000])
    p_same, d_same = net.score(torch.from_numpy(ref[:20029].astype(np.float32)).cuda(), window_stride=24)
    p_e2e, d_e2e = net.score(torch.from_numpy(feat[:20029].astype(np.float32)).cuda(), window_stride=24)
    ref_dec = pipeline_ref.decide(y)
    print('%-15s posterior err: same features max %.2e | own features max %.2e ; decisions differing %d / %d (same), %d (own)'
          % (kind, np.abs(p_same.cpu().numpy() - y).max(), np.abs(p_e2e.cpu().numpy() - y).max(),
             int((d_same.cpu().numpy() != ref_dec).sum()), len(y), int((d_e2e.cpu().numpy() != ref_dec).sum())))

# float32 products on the bf16 / f16 matrix pipe (vad_set_matrix_mode) against a float64 evaluation of the same models:
# how far each kernel -- and float32 PyTorch on the CPU -- is from the exact answer
print('\nerror against float64 evaluation of the same weights (20000 windows):')
x64 = torch.from_numpy(xw[:20000].astype(np.float64))
fdev = torch.from_numpy(ref[:20029].astype(np.float32)).cuda()
for kind in ('gru_large', 'gru_small', 'lstm_large', 'lstm_small'):
    r = models_ref.make_model(kind)
    net = pkg.Net(large=kind.endswith('large'), lstm=kind.startswith('lstm'))
    net.load_state_dict(r.state_dict())
    y32 = pipeline_ref.score_windows(r, xw[:20000])
    with torch.no_grad():
        y64 = r.double()(x64).numpy()
    row = ['torch-cpu-f32 %.2e' % np.abs(y32 - y64).max()]
    for mode in ('f32', 'bf16x9', 'bf16x8', 'bf16x6', 'f16x6', 'f16x4'):
        pkg.set_matrix_mode(mode)
        p, _ = net.score(fdev, window_stride=24)
        row.append('%s max %.2e mean %.2e' % (mode, np.abs(p.cpu().numpy() - y64).max(), np.abs(p.cpu().numpy() - y64).mean()))
    pkg.set_matrix_mode('f32')
    print('%-12s ' % kind + ' | '.join(row))

# the wide GRU (hidden 90) and a stress case for the float16 split: weights and features scaled down so that most
# operands lie below 0.5, where float16's subnormal grid (2^-24) rounds them
print('\nhidden 90 and small-operand stress (weights x 0.05), against float64:')
for name, scale, hidden in (('gru_h90', 1.0, 90), ('gru_large x0.05', 0.05, 30), ('gru_h90 x0.05', 0.05, 90)):
    r = models_ref.make_model('gru_large', hidden=hidden)
    with torch.no_grad():
        r.rnn.weight_hh_l0.mul_(scale)
        r.lin1.weight.mul_(scale * 4)
    net = pkg.Net(large=True, lstm=False, hidden=hidden)
    net.load_state_dict(r.state_dict())
    with torch.no_grad():
        y64 = r.double()(x64[:5000]).numpy()
    r.float()
    row = []
    for mode in ('f32', 'bf16x9', 'bf16x8', 'bf16x6', 'f16x6', 'f16x4'):
        p, _ = net.score(fdev[:5029], window_stride=24, matrix_mode=mode)
        e = np.abs(p.cpu().numpy() - y64)
        row.append('%s max %.2e mean %.2e' % (mode, e.max(), e.mean()))
    print('%-16s ' % name + ' | '.join(row))

# NickNet: float32 MFMA for the convolutions and the first GRU layer's input, six float16 partial products for the
# products of a hidden state; the dense-batch kernel and the stride-1 kernel (trunk once per row) against float64
print('\nNickNet against a float64 evaluation of the same weights (20000 windows):')
from oracle import nicknet_ref
for large in (True, False):
    r = nicknet_ref.make_nicknet(large)
    net = pkg.NickNet(large=large)
    net.load_state_dict(r.state_dict())
    with torch.no_grad():
        y32 = r(torch.from_numpy(xw[:20000])).numpy()
        y64 = r.double()(x64).numpy()
    r.float()
    p_stream, _ = net.score(fdev, window_stride=24)
    p_batch, _ = net.score(torch.from_numpy(xw[:20000]).cuda())
    e = np.abs(p_stream.cpu().numpy() - y64)
    print('nicknet_%-6s torch-cpu-f32 %.2e | stride-1 kernel max %.2e mean %.2e | dense-batch kernel equal bits: %s'
          % ('large' if large else 'small', np.abs(y32 - y64).max(), e.max(), e.mean(), bool(torch.equal(p_stream, p_batch))))
