"""GPU parity of the MFCC/delta front end (vad_mfcc_delta_f32) against the CPU oracle.

Tolerance (north star: "MFCC features within 1e-4 relative tolerance"): the
reference computes in float64, the kernel's FFT in float32, and lifted cepstra
and especially deltas pass through zero, so the check is
    |gpu - ref| <= 1e-4 * max(|ref|, FLOOR)      with FLOOR = 1.0
(lifted cepstra of audible frames span roughly +-60).
"""
import numpy as np
import pytest
import torch

from oracle import mfcc_ref, pipeline_ref, synth_ref

pytestmark = pytest.mark.gpu

RTOL = 1e-4
FLOOR = 1.0


def assert_feat_close(got, ref, what=''):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), FLOOR)
    worst = np.unravel_index(np.argmax(err), err.shape)
    assert err.max() <= RTOL, '%s: max scaled err %.3e at %s (got %r ref %r)' % (
        what, err.max(), worst, got[worst], ref[worst])
    return err.max()


def test_synth_audio_bit_exact(vad):
    for first, n, seed in [(0, 48000, 1337), (123457, 9999, 7), (2 ** 32 - 1000, 4000, 1337)]:
        dev = vad.synth_audio(first, n, seed=seed)
        np.testing.assert_array_equal(dev.cpu().numpy(), synth_ref.synth_audio(first, n, seed))


def test_process_test_frames_matches_oracle(vad):
    fr = synth_ref.synth_frames(50, 169, seed=1337)
    frames, m, d = vad.process_test_frames(fr[3:], fr[:3])
    rf, rm, rd = mfcc_ref.process_test_frames(fr[3:], fr[:3])
    np.testing.assert_array_equal(frames, rf)
    assert m.shape == (166, 12) and d.shape == (166, 12)
    assert_feat_close(m, rm, 'mfcc')
    assert_feat_close(d, rd, 'delta')


def test_golden_slices(vad):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mfcc_slices.npz'))
    mf, de = vad.mfcc_delta_slices(torch.from_numpy(g['rows']).cuda())
    for s in range(g['rows'].shape[0]):
        assert_feat_close(mf[s].cpu().numpy(), g['mfcc'][s], 'golden mfcc %d' % s)
        assert_feat_close(de[s].cpu().numpy(), g['delta'][s], 'golden delta %d' % s)


def test_zero_alignment_and_silence(vad):
    z = np.zeros((3, 480), np.int16)
    fr = synth_ref.synth_frames(0, 60, seed=5)
    _, m, d = vad.process_test_frames(fr, z)
    _, rm, rd = mfcc_ref.process_test_frames(fr, z)
    assert_feat_close(m, rm, 'mfcc')
    assert_feat_close(d, rd, 'delta')
    # an all-zero slice drives every filter into the zero-energy clamp: cepstra and deltas vanish
    zz = np.zeros((20, 480), np.int16)
    _, m0, d0 = vad.process_test_frames(zz, z)
    assert np.abs(m0).max() < 1e-6 and np.abs(d0).max() < 1e-6


def test_full_scale_and_dc(vad):
    rng = np.random.default_rng(3)
    fr = rng.integers(-32768, 32768, size=(40, 480)).astype(np.int16)
    fr[10:14] = 32767                      # saturated DC rows
    fr[20:22] = -32768
    al = rng.integers(-32768, 32768, size=(3, 480)).astype(np.int16)
    _, m, d = vad.process_test_frames(fr, al)
    _, rm, rd = mfcc_ref.process_test_frames(fr, al)
    assert_feat_close(m, rm, 'mfcc')
    assert_feat_close(d, rd, 'delta')


@pytest.mark.parametrize('n', [1, 2, 3, 5, 33])
def test_short_slices(vad, n):
    fr = synth_ref.synth_frames(200, n + 3, seed=11)
    _, m, d = vad.process_test_frames(fr[3:], fr[:3])
    _, rm, rd = mfcc_ref.process_test_frames(fr[3:], fr[:3])
    assert_feat_close(m, rm, 'mfcc')
    assert_feat_close(d, rd, 'delta')


def test_batched_slices(vad):
    S, N = 7, 50
    rng = np.random.default_rng(9)
    slices = np.stack([synth_ref.synth_frames(1000 * s + 17, N + 3, seed=s) for s in range(S)])
    slices[3] = rng.integers(-500, 500, size=(N + 3, 480))
    mf, de = vad.mfcc_delta_slices(torch.from_numpy(slices).cuda())
    for s in range(S):
        _, rm, rd = mfcc_ref.process_test_frames(slices[s, 3:], slices[s, :3])
        assert_feat_close(mf[s].cpu().numpy(), rm, 'mfcc slice %d' % s)
        assert_feat_close(de[s].cpu().numpy(), rd, 'delta slice %d' % s)


@pytest.mark.parametrize('n_frames,slice_len', [(400, 166), (332, 166), (167, 166), (100, 0), (500, 33)])
def test_stream_slicing(vad, n_frames, slice_len):
    stream = vad.synth_stream(0, n_frames, seed=1337)
    feat = vad.mfcc_delta_stream(stream, n_frames, slice_len).cpu().numpy()
    host = synth_ref.synth_frames(0, n_frames, seed=1337)
    ref = pipeline_ref.features_sliced(host, slice_len)
    assert_feat_close(feat[:, :12], ref[:, :12], 'mfcc')
    assert_feat_close(feat[:, 12:], ref[:, 12:], 'delta')


def test_stream_starting_mid_stream(vad):
    """A shard that starts at frame f0 > 0 carries the 3 preceding frames as alignment rows."""
    f0, n = 166 * 3, 166 * 2
    stream = vad.synth_stream(f0, n, seed=1337)
    feat = vad.mfcc_delta_stream(stream, n, 166).cpu().numpy()
    host = synth_ref.synth_frames(0, f0 + n, seed=1337)
    ref = pipeline_ref.features_sliced(host, 166)[f0:]
    assert_feat_close(feat, ref, 'shard')


def test_argument_errors(vad):
    with pytest.raises(ValueError):
        vad.process_test_frames(np.zeros((4, 481), np.int16), np.zeros((3, 480), np.int16))
    with pytest.raises(ValueError):
        vad.process_test_frames(np.zeros((4, 480), np.int16), np.zeros((2, 480), np.int16))
    with pytest.raises(ValueError):
        vad.mfcc_delta_stream(torch.zeros(480 * 5, dtype=torch.int16, device='cuda'), 10)


def test_large_stream_properties(vad):
    """Size-independent checks at a size the oracle does not sweep in full (10 min of audio):
    slices are independent, so any slice recomputed alone must reproduce the big run bit for bit,
    and spot slices must match the oracle."""
    n = 166 * 120 + 41
    stream = vad.synth_stream(0, n, seed=99)
    feat = vad.mfcc_delta_stream(stream, n, 166)
    assert torch.isfinite(feat).all()
    host = stream.cpu().numpy().reshape(-1, 480)
    for s in (0, 57, 119, 120):
        lo = 166 * s
        hi = min(lo + 166, n)
        aligned = host[lo: hi + 3]
        mf, de = vad.mfcc_delta_slices(torch.from_numpy(aligned[None]).cuda())
        assert torch.equal(mf[0], feat[lo:hi, :12]) and torch.equal(de[0], feat[lo:hi, 12:])
        _, rm, rd = mfcc_ref.process_test_frames(aligned[3:], aligned[:3])
        assert_feat_close(feat[lo:hi, :12].cpu().numpy(), rm, 'mfcc slice %d' % s)
        assert_feat_close(feat[lo:hi, 12:].cpu().numpy(), rd, 'delta slice %d' % s)


def test_full_size_stream_properties(vad):
    """Benchmark configuration size: 100 h = 12 M frames (11.5 GB of int16) through the whole path.
    The oracle cannot sweep this, so the checks are size-independent properties: determinism,
    slice independence (any slice recomputed alone reproduces the big run bit for bit), window
    independence (any window recomputed from a materialised batch reproduces its posterior bit for
    bit), oracle parity on spot slices, and well-formed outputs."""
    from oracle import models_ref
    n = 12_000_000
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * 2 ** 30:
        pytest.skip('the benchmark-size check needs 20 GB of free device memory (an idle MI355X has 288 GB)')
    ref = models_ref.make_model('gru_large')
    net = vad.Net(large=True, lstm=False)
    net.load_state_dict(ref.state_dict())
    stream = vad.synth_stream(0, n, seed=1337)
    ws = vad.StreamWorkspace(n, stream.device)
    feat, p, dec = vad.vad_stream(net, stream, n, workspace=ws)
    torch.cuda.synchronize()
    assert p.shape == (n - 29, 2) and dec.shape == (n - 29,)
    assert bool(torch.isfinite(feat).all()) and bool(torch.isfinite(p).all())
    assert float((p.sum(1) - 1).abs().max()) < 1e-6
    assert bool(((p[:, 1] > p[:, 0]).to(torch.uint8) == dec).all())
    frac_speech = float(dec.float().mean())
    assert 0.0 < frac_speech < 1.0
    # determinism
    p_first = p.clone()
    feat_first = feat[:100000].clone()
    vad.vad_stream(net, stream, n, workspace=ws)
    assert torch.equal(p, p_first) and torch.equal(feat[:100000], feat_first)
    # slice independence + oracle parity on spot slices (first, middle, last full, ragged tail)
    host_rows = lambda lo, hi: stream[lo * 480: hi * 480].cpu().numpy().reshape(-1, 480)
    n_slices = (n + 165) // 166
    for s in (0, 36144, n_slices - 2, n_slices - 1):
        lo, hi = 166 * s, min(166 * s + 166, n)
        aligned = host_rows(lo, hi + 3)                       # padded-stream rows lo .. hi+2
        mf, de = vad.mfcc_delta_slices(torch.from_numpy(aligned[None]).cuda())
        assert torch.equal(mf[0], feat[lo:hi, :12]) and torch.equal(de[0], feat[lo:hi, 12:])
        _, rm, rd = mfcc_ref.process_test_frames(aligned[3:], aligned[:3])
        assert_feat_close(feat[lo:hi, :12].cpu().numpy(), rm, 'mfcc slice %d' % s)
        assert_feat_close(feat[lo:hi, 12:].cpu().numpy(), rd, 'delta slice %d' % s)
    # window independence: posteriors of scattered windows from a materialised (B,30,24) batch, which the
    # float32-MFMA kernels compute; the stream's default matrix mode (f16x6) stays within the bar of them and
    # of the oracle, and in the float32 matrix mode the whole stream reproduces them bit for bit
    idx = torch.tensor([0, 1, 31, 32, 255, 256, 4_000_000, 7_777_777, n - 31, n - 30], device='cuda')
    batch = torch.stack([feat[i:i + 30] for i in idx.tolist()])
    pb, _ = net.score(batch.contiguous())
    with torch.no_grad():
        y = ref(batch.cpu()).numpy()
    assert np.abs(pb.cpu().numpy() - y).max() <= 1e-5
    assert np.abs(p[idx].cpu().numpy() - y).max() <= 1e-5
    prev = vad.set_matrix_mode('f32')
    try:
        p32, _ = net.score(feat, window_stride=24)
    finally:
        vad.set_matrix_mode(prev)
    assert torch.equal(pb, p32[idx])
    assert float((p32 - p).abs().max()) <= 1e-5                     # all 12 M windows, the two matrix modes


@pytest.mark.parametrize('kind', ['nicknet', 'nicknet_small', 'densenet', 'densenet_large', 'lstm_small', 'gru_h90'])
def test_full_size_window_independence_of_the_other_models(vad, kind):
    """The other stride-1 model kernels at the benchmark's size (100 h = 12 M feature rows): every window's posterior
    is a function of its own 30 rows, so windows sampled across the stream and recomputed from a materialised
    (B,30,24) batch -- which takes the any-placement kernel -- reproduce the stream's values: bit for bit where the two
    kernels run the same instruction chains (NickNet, DenseNet), within the 1e-5 bar where the stream kernel forms its
    products on the 16-bit pipe (LSTM, wide GRU).  Also determinism and well-formed outputs."""
    from oracle import models_ref, nicknet_ref
    n = 12_000_000
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2 ** 30:
        pytest.skip('needs 24 GB of free device memory (an idle MI355X has 288 GB)')
    if kind.startswith('nicknet'):
        ref = nicknet_ref.make_nicknet(kind == 'nicknet')
        net = vad.NickNet(large=kind == 'nicknet')
    elif kind.startswith('densenet'):
        ref = models_ref.make_model(kind)
        net = vad.DenseNet(large=kind.endswith('large'))
    elif kind == 'gru_h90':
        ref = models_ref.make_model('gru_large', hidden=90)
        net = vad.Net(large=True, lstm=False, hidden=90)
    else:
        ref = models_ref.make_model(kind)
        net = vad.Net(large=False, lstm=True)
    net.load_state_dict(ref.state_dict())
    g = torch.Generator(device='cuda').manual_seed(11)
    feat = torch.empty((n, 24), device='cuda')
    feat.normal_(0.0, 6.0, generator=g)
    p, dec = net.score(feat, window_stride=24)
    torch.cuda.synchronize()
    assert p.shape == (n - 29, 2) and bool(torch.isfinite(p).all())
    assert float((p.sum(1) - 1).abs().max()) < 1e-6
    assert bool(((p[:, 1] > p[:, 0]).to(torch.uint8) == dec).all())
    p2, _ = net.score(feat, window_stride=24)
    assert torch.equal(p, p2)
    idx = torch.cat((torch.tensor([0, 1, 15, 16, 17, 127, 128, n - 31, n - 30], device='cuda'),
                     torch.randint(0, n - 29, (2000,), generator=g, device='cuda')))
    rows = idx[:, None] + torch.arange(30, device='cuda')[None, :]
    pb, _ = net.score(feat[rows].contiguous())
    if kind.startswith(('nicknet', 'densenet')):
        assert torch.equal(pb, p[idx])
    else:
        assert float((pb - p[idx]).abs().max()) <= 1e-5
    with torch.no_grad():
        y = ref(feat[rows[:64]].cpu()).numpy()
    assert np.abs(p[idx[:64]].cpu().numpy() - y).max() <= 1e-5


def test_gain_and_shift_properties(vad):
    """Size-independent properties on a long stream: c1..c12 ignore a gain of 2 (the DCT rows annihilate the
    log 4 offset), and with the whole stream as one slice the features of a stream that starts one frame
    later are the same rows shifted by one (except the un-emphasised first row's reach and the delta edge)."""
    n = 50000
    stream = vad.synth_stream(0, n + 1, seed=77)
    quiet = torch.div(stream, 4, rounding_mode='trunc').to(torch.int16)          # headroom for the doubling
    f1 = vad.mfcc_delta_stream(quiet, n, 166)
    f2 = vad.mfcc_delta_stream((quiet * 2).to(torch.int16), n, 166)
    host = quiet.cpu().numpy().reshape(-1, 480)
    audible = torch.from_numpy(np.convolve((np.abs(host).max(1) > 0).astype(int), np.ones(4, int), 'full')[3:n + 3] > 0).cuda()
    err = ((f2 - f1).abs() / f1.abs().clamp(min=1.0))[audible]               # frames with any energy in their window
    assert float(err.max()) <= 2e-4                                          # two fp32 evaluations, each within 1e-4
    # time shift, single slice
    a = vad.mfcc_delta_stream(stream, n, 0)
    b = vad.mfcc_delta_stream(stream[480:], n, 0)
    assert torch.equal(a[6:n - 2, :12], b[5:n - 3, :12])                     # MFCC rows away from the slice's first frames
    assert torch.equal(a[8:n - 4, 12:], b[7:n - 5, 12:])                     # deltas two more rows in


def test_process_test_data_matches_the_slice_loop(vad):
    """One device call for a recording against the reference's loop of per-slice calls (oracle side)."""
    n = 166 * 3 + 50
    fr = synth_ref.synth_frames(0, n, seed=31)
    labels = (np.arange(n) // 40 % 2).astype(np.int8)
    out = vad.process_test_data(fr, labels)
    assert out['count'] == 166 * 3
    pos = 0
    for s in vad.recording_slices(n):
        align = np.zeros((3, 480), np.int16) if pos == 0 else fr[pos - 3: pos]
        _, rm, rd = mfcc_ref.process_test_frames(fr[s[0]: s[1]], align)
        assert_feat_close(out['mfcc'][pos: pos + 166], rm.astype(np.float32), 'mfcc slice at %d' % pos)
        assert_feat_close(out['delta'][pos: pos + 166], rd.astype(np.float32), 'delta slice at %d' % pos)
        pos += 166
    np.testing.assert_array_equal(out['frames'][:pos], fr[:pos])
    np.testing.assert_array_equal(out['labels'][:pos], labels[:pos])
    assert not out['mfcc'][pos:].any() and not out['frames'][pos:].any()
    assert out['mfcc'].dtype == np.float32 and out['labels'].dtype == np.int8
    assert vad.process_test_data(fr[:100], labels[:100])['count'] == 0


@pytest.mark.parametrize('lengths', [[33, 166, 1, 40, 166, 7], [166] * 3 + [5], [300, 2, 2, 700], [1] * 9, [0, 50, 0, 0, 34]])
def test_ragged_slices_match_per_slice_calls(vad, lengths):
    """vad_mfcc_delta_slices_f32: one launch over slices of different lengths (the training-data loop's
    33..166-frame cuts, also degenerate, empty and longer-than-a-block ones) against one oracle call per slice."""
    starts = np.concatenate(([0], np.cumsum(lengths)))
    n = int(starts[-1])
    stream = vad.synth_stream(0, n, seed=21)
    feat = vad.mfcc_delta_ragged(stream, starts).cpu().numpy()
    host = stream.cpu().numpy().reshape(-1, 480)                         # 3 zero rows, then the frames
    for lo, hi in zip(starts[:-1], starts[1:]):
        if hi == lo:
            continue
        _, rm, rd = mfcc_ref.process_test_frames(host[lo + 3: hi + 3], host[lo: lo + 3])
        assert_feat_close(feat[lo:hi, :12], rm, 'mfcc slice at %d' % lo)
        assert_feat_close(feat[lo:hi, 12:], rd, 'delta slice at %d' % lo)
    # the same slices through the uniform entry point, slice by slice: bit-identical
    for lo, hi in zip(starts[:-1], starts[1:]):
        if hi > lo:
            mf, de = vad.mfcc_delta_slices(torch.from_numpy(host[lo: hi + 3][None]).cuda())
            assert np.array_equal(mf[0].cpu().numpy(), feat[lo:hi, :12])
            assert np.array_equal(de[0].cpu().numpy(), feat[lo:hi, 12:])


def test_ragged_slices_understated_maximum_and_errors(vad):
    """A slice longer than the stated max_slice_frames is still computed in full (its blocks wrap around)."""
    import ctypes
    from voice_activity_detection_amd import _lib
    lengths = [40, 900, 35]
    starts = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    n = int(starts[-1])
    stream = vad.synth_stream(0, n, seed=23)
    want = vad.mfcc_delta_ragged(stream, starts)
    got = torch.full((n, 24), float('nan'), device='cuda')
    sd = torch.from_numpy(starts).cuda()
    L = vad.lib()
    for stated in (40, 300):
        got.fill_(float('nan'))
        _lib.check(L.vad_mfcc_delta_slices_f32(ctypes.c_void_p(stream.data_ptr()), ctypes.c_void_p(sd.data_ptr()), 3, n,
                                               stated, ctypes.c_void_p(got.data_ptr()), 24,
                                               ctypes.c_void_p(got.data_ptr() + 48), 24, _lib.current_stream()))
        assert torch.equal(got, want)
    with pytest.raises(ValueError):
        vad.mfcc_delta_ragged(stream, [0, 50, 40, n])
    with pytest.raises(ValueError):
        vad.mfcc_delta_ragged(stream, [0, n + 1])


def test_long_single_slice_is_split_without_seams(vad):
    """A one-slice stream longer than a block's run: the runs recompute two MFCC rows on either side of their
    seams, so the result equals the oracle's single call, and a differently placed seam changes no bit."""
    n = 1500
    stream = vad.synth_stream(0, n, seed=1337)
    feat = vad.mfcc_delta_stream(stream, n, 0)
    host = stream.cpu().numpy().reshape(-1, 480)
    _, rm, rd = mfcc_ref.process_test_frames(host[3:], host[:3])
    assert_feat_close(feat[:, :12].cpu().numpy(), rm, 'mfcc')
    assert_feat_close(feat[:, 12:].cpu().numpy(), rd, 'delta')
    # interior rows do not depend on where the slice ends, hence not on where the last seams fall
    shorter = vad.mfcc_delta_stream(stream, n - 100, 0)
    assert torch.equal(shorter[: n - 102], feat[: n - 102])


def test_process_training_data_matches_the_slice_loop(vad):
    """One mix and one front-end launch per noise level against the reference's loop of add_noise calls
    (restated on the oracle side, same random draws)."""
    rng = np.random.default_rng(5)
    speech = synth_ref.synth_frames(0, 700, seed=41)
    labels = (np.arange(700) // 25 % 2).astype(np.int8)
    noise = (synth_ref.synth_frames(5000, 900, seed=43) // 4).astype(np.int16)
    got = vad.process_training_data(speech, labels, noise)
    ref = pipeline_ref.process_training_data(speech, labels, noise)
    assert got['count'] == ref['count'] and got['slices'] == ref['slices']
    lens = sorted(s[0] if s[0] == s[1] else s[1] - s[0] for s in got['slices'])
    assert lens[0] >= 33 and len(lens) > 8
    np.testing.assert_array_equal(got['labels'], ref['labels'])
    for key in ('None', '-15', '-3'):
        np.testing.assert_array_equal(got['frames-' + key], ref['frames-' + key])
        assert got['mfcc-' + key].shape == ref['mfcc-' + key].shape and got['mfcc-' + key].dtype == np.float32
        assert_feat_close(got['mfcc-' + key], ref['mfcc-' + key], 'mfcc ' + key)
        assert_feat_close(got['delta-' + key], ref['delta-' + key], 'delta ' + key)
    assert not np.array_equal(got['frames-None'], got['frames--15'])
