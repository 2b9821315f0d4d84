"""Does running the front end of chunk k+1 beside the GRU of chunk k (two streams) beat running them back to back?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voice_activity_detection_amd as pkg
from bench import make_net

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else 'f32'
pkg.set_matrix_mode(mode)
net = make_net("gru_large")
n = int(hours * 120000) // (166 * chunks) * (166 * chunks)
stream = pkg.synth_stream(0, n, seed=1337)
feat = torch.empty(n, 24, device='cuda')
p = torch.empty(n - 29, 2, device='cuda')
d = torch.empty(n - 29, dtype=torch.uint8, device='cuda')
per = n // chunks


def sequential():
    pkg.mfcc_delta_stream(stream, n, 166, out=feat)
    net.score(feat, window_stride=24, n_windows=n - 29, out=p, decisions=d)


sa, sb = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)       # the model kernel's blocks go first


def pipelined():
    evs = []
    for k in range(chunks):
        with torch.cuda.stream(sa):
            lo = k * per
            # chunk k of the stream with its 3 alignment rows (the stream tensor starts with 3 rows of lead-in)
            pkg.mfcc_delta_stream(stream[lo * 480:], per, 166, out=feat[lo:lo + per])
            e = torch.cuda.Event()
            e.record(sa)
            evs.append(e)
        with torch.cuda.stream(sb):
            # windows starting in chunk k-1 need rows up to the first 29 of chunk k
            if k >= 1:
                sb.wait_event(evs[k])
                w0 = (k - 1) * per
                net.score(feat[w0:], window_stride=24, n_windows=per, out=p[w0:w0 + per], decisions=d[w0:w0 + per])
    with torch.cuda.stream(sb):
        w0 = (chunks - 1) * per
        net.score(feat[w0:], window_stride=24, n_windows=per - 29, out=p[w0:], decisions=d[w0:])
    sa.synchronize()
    sb.synchronize()


for fn in (sequential, pipelined, sequential, pipelined):
    fn()
    torch.cuda.synchronize()
    if fn is sequential:
        ref = p.clone()
    else:
        assert torch.equal(p, ref), 'pipelined result differs'
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print('%-11s %s %6.2f ms per step' % (fn.__name__, mode, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
