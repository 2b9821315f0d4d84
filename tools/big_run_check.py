"""One-off: the whole path on a shard of config-5 size (1250 h = 150 M frames, 144 GB of int16 on one GPU):
64-bit indexing check -- the tail of the big run must equal a small run over the same audio bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voice_activity_detection_amd as pkg
from bench import make_net

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1250.0
kind = sys.argv[2] if len(sys.argv) > 2 else 'gru_large'
n = int(hours * 120000) // 166 * 166 + 77                       # ragged last slice
net = make_net(kind)
print('frames', n, 'audio GB', (n + 3) * 960 / 1e9, flush=True)
stream = pkg.synth_stream(0, n, seed=1337)
ws = pkg.StreamWorkspace(n, stream.device)
torch.cuda.synchronize()
t0 = time.perf_counter()
feat, p, dec = pkg.vad_stream(net, stream, n, workspace=ws)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('whole path: %.2f s = %.0f audio-hours/s' % (dt, hours / dt), flush=True)
assert bool(torch.isfinite(p[-100000:]).all()) and bool(torch.isfinite(p[:100000]).all())
# recompute the last 40 slices alone: frames [f0, n) with their 3 alignment rows regenerated
f0 = (n // 166 - 40) * 166
small = pkg.synth_stream(f0, n - f0, seed=1337)
feat_s, p_s, dec_s = pkg.vad_stream(net, small, n - f0)
assert torch.equal(feat_s, feat[f0:]), 'features differ at the far end'
assert torch.equal(p_s, p[f0:]) and torch.equal(dec_s, dec[f0:]), 'posteriors differ at the far end'
# and a middle stretch past the 2^31-float / 2^32-byte marks of every buffer
for f1 in (4_700_000 // 166 * 166, 95_000_000 // 166 * 166):
    if f1 + 166 * 20 < n:
        mid = pkg.synth_stream(f1, 166 * 20, seed=1337)
        feat_m, p_m, _ = pkg.vad_stream(net, mid, 166 * 20)
        assert torch.equal(feat_m, feat[f1:f1 + 166 * 20])
        assert torch.equal(p_m, p[f1:f1 + 166 * 20 - 29])
print('OK: far-end and mid-stream recomputations are bit-identical; speech fraction %.4f' % float(dec.float().mean()))
