"""GPU parity of the window models (vad_gru/lstm/densenet_forward_f32).

Bar (north star): speech posteriors within 1e-5 absolute of the reference in
fp32; decisions identical except for windows within tolerance of the threshold.
The golden outputs under tests/golden were produced by the reference's own
classes (tests/golden/make_model_fixtures.py).
"""
import os

import numpy as np
import pytest
import torch

from oracle import models_ref, pipeline_ref, synth_ref

pytestmark = pytest.mark.gpu

ATOL = 1e-5
NET_KINDS = {'gru_large': (True, False), 'gru_small': (False, False),
             'lstm_large': (True, True), 'lstm_small': (False, True)}
DENSE_KINDS = {'densenet': False, 'densenet_large': True}


def _fixture(golden_dir, kind):
    z = np.load(os.path.join(golden_dir, 'model_%s.npz' % kind))
    x = np.load(os.path.join(golden_dir, 'model_inputs.npz'))['x']
    sd = {k: torch.from_numpy(z[k]) for k in z.files if k not in ('y', 'n_params')}
    return sd, x, z['y']


def _make(vad, kind):
    if kind in NET_KINDS:
        large, lstm = NET_KINDS[kind]
        return vad.Net(large=large, lstm=lstm)
    return vad.DenseNet(large=DENSE_KINDS[kind])


def _gpu_model(vad, kind, ref):
    net = _make(vad, kind)
    net.load_state_dict(ref.state_dict())
    return net


def _decisions_agree(dec, p_ref, threshold=None, tol=ATOL):
    p_ref = np.asarray(p_ref)
    ref = pipeline_ref.decide(p_ref, threshold)
    margin = np.abs(p_ref[:, 1] - (p_ref[:, 0] if threshold is None else threshold))
    diff = dec != ref
    assert not np.any(diff & (margin > 2 * tol)), 'decisions differ away from the threshold'
    return int(diff.sum())


@pytest.mark.parametrize('kind', list(NET_KINDS) + list(DENSE_KINDS))
def test_forward_matches_reference_fixture(vad, golden_dir, kind):
    sd, x, y = _fixture(golden_dir, kind)
    net = _make(vad, kind)
    net.load_state_dict(sd)
    out = net(torch.from_numpy(x))                 # CPU tensor in, like the reference's callers
    assert out.is_cuda and out.shape == (64, 2) and out.dtype == torch.float32
    got = out.cpu().numpy()
    assert np.abs(got - y).max() <= ATOL, np.abs(got - y).max()
    np.testing.assert_allclose(got.sum(1), 1.0, atol=1e-6)


@pytest.mark.parametrize('kind', list(NET_KINDS) + list(DENSE_KINDS))
@pytest.mark.parametrize('batch', [1, 31, 33, 127, 129, 2048])
def test_ragged_batches_match_oracle(vad, kind, batch):
    ref = models_ref.make_model(kind)
    net = _make(vad, kind)
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(batch)
    x = torch.randn(batch, 30, 24, generator=g) * 4.0
    with torch.no_grad():
        y = ref(x).numpy()
    p, dec = net.score(x.cuda())
    assert np.abs(p.cpu().numpy() - y).max() <= ATOL
    _decisions_agree(dec.cpu().numpy(), y)


@pytest.mark.parametrize('kind', list(NET_KINDS))
def test_fixture_through_the_stride1_kernels(vad, golden_dir, kind):
    """The reference-generated fixture through the stride-1 GRU / LSTM kernels in every matrix mode (the default,
    f16x6, first): the 64 fixture windows are laid end to end as a (1920,24) feature matrix, whose stride-1
    windows 0, 30, 60, ... are exactly the fixture windows."""
    sd, x, y = _fixture(golden_dir, kind)
    net = _make(vad, kind)
    net.load_state_dict(sd)
    feat = torch.from_numpy(x.reshape(-1, 24)).cuda()
    assert vad.lib().vad_get_matrix_mode() == 4                       # VAD_MATRIX_F16X6 is the default
    for mode in ('f16x6', 'f16x4', 'bf16x8', 'bf16x9', 'bf16x6', 'f32'):
        prev = vad.set_matrix_mode(mode)
        try:
            p, _ = net.score(feat, window_stride=24)
        finally:
            vad.set_matrix_mode(prev)
        got = p[::30].cpu().numpy()
        assert got.shape == y.shape
        assert np.abs(got - y).max() <= ATOL, (mode, np.abs(got - y).max())


@pytest.mark.parametrize('kind', ['gru_large', 'lstm_small', 'densenet', 'densenet_large'])
def test_sliding_windows_equal_explicit_windows(vad, kind, f32_matrix_mode):
    """Stride-1 windows read in place from the (F,24) matrix give the same bits as a
    materialised (B,30,24) batch (GRU / LSTM: in the float32-MFMA matrix mode), and both match the
    oracle on oracle-made features."""
    ref = models_ref.make_model(kind)
    net = _make(vad, kind)
    net.load_state_dict(ref.state_dict())
    frames = synth_ref.synth_frames(0, 700, seed=1337)
    feat = pipeline_ref.features_sliced(frames, 166)
    xw = pipeline_ref.windows(feat)
    y = pipeline_ref.score_windows(ref, xw)
    fdev = torch.from_numpy(feat).cuda()
    p_slide, d_slide = net.score(fdev, window_stride=24)
    p_batch, d_batch = net.score(torch.from_numpy(xw).cuda())
    assert p_slide.shape == (671, 2)
    assert torch.equal(p_slide, p_batch) and torch.equal(d_slide, d_batch)
    assert np.abs(p_slide.cpu().numpy() - y).max() <= ATOL
    # strided windows (STEP_SIZE = 6, core/train.py:54)
    p6, _ = net.score(fdev, window_stride=24 * 6)
    assert torch.equal(p6, p_slide[::6])
    # the default matrix mode of the stride-1 kernels (f16x6) holds the same bar against the oracle
    vad.set_matrix_mode('f16x6')
    p_x8, _ = net.score(fdev, window_stride=24)
    assert np.abs(p_x8.cpu().numpy() - y).max() <= ATOL


@pytest.mark.parametrize('kind', ['gru_large', 'lstm_small'])
def test_threshold_decisions(vad, kind):
    ref = models_ref.make_model(kind)
    net = _make(vad, kind)
    net.load_state_dict(ref.state_dict())
    x = torch.randn(1000, 30, 24, generator=torch.Generator().manual_seed(5)) * 5.0
    with torch.no_grad():
        y = ref(x).numpy()
    for t in (0.3, 0.5, 0.52):
        p, dec = net.score(x.cuda(), threshold=t)
        got = dec.cpu().numpy()
        np.testing.assert_array_equal(got, (p.cpu().numpy()[:, 1] >= np.float32(t)).astype(np.uint8))
        _decisions_agree(got, y, threshold=t)


@pytest.mark.parametrize('hidden,large', [(90, True), (90, False), (64, True), (33, False), (17, True)])
def test_gru_other_hidden_sizes(vad, hidden, large, f32_matrix_mode):
    """Enlarged GRU (hidden 90, ~100k parameters): no class in the reference, oracle = torch.nn.GRU + Linear."""
    ref = models_ref.make_model('gru_large' if large else 'gru_small', hidden=hidden)
    net = vad.Net(large=large, lstm=False, hidden=hidden)
    net.load_state_dict(ref.state_dict())
    if hidden == 90 and large:
        assert sum(p.numel() for p in ref.rnn.parameters()) + sum(p.numel() for p in ref.lin1.parameters()) \
            + sum(p.numel() for p in ref.lin2.parameters()) == 101600
    x = torch.randn(300, 30, 24, generator=torch.Generator().manual_seed(hidden)) * 4.0
    with torch.no_grad():
        y = ref(x).numpy()
    p, dec = net.score(x.cuda())
    assert np.abs(p.cpu().numpy() - y).max() <= ATOL
    _decisions_agree(dec.cpu().numpy(), y)
    # stride-1 windows over a feature matrix take the same path for wide models
    feat = torch.randn(200, 24, generator=torch.Generator().manual_seed(1)) * 3.0
    xw = torch.from_numpy(pipeline_ref.windows(feat.numpy()))
    ps, _ = net.score(feat.cuda(), window_stride=24)
    pb, _ = net.score(xw.cuda())
    assert torch.equal(ps, pb)


@pytest.mark.parametrize('hidden,large', [(90, True), (64, False), (40, True)])
def test_wide_gru_workspace_chunks(vad, hidden, large, f32_matrix_mode):
    """vad_gru_forward_ws_f32: the stride-1 path of wide GRUs projects each feature row once into a caller-owned
    workspace.  Whatever the workspace size (one chunk, several ragged chunks, too small to be used at all) the
    posteriors equal those of the plain entry point bit for bit (float32 matrix mode; the bf16 modes of the
    workspace path: test_wide_gru_split_kernel)."""
    import ctypes
    from voice_activity_detection_amd import _lib
    L = vad.lib()
    ref = models_ref.make_model('gru_large' if large else 'gru_small', hidden=hidden)
    net = vad.Net(large=large, lstm=False, hidden=hidden)
    net.load_state_dict(ref.state_dict())
    n_rows = 1000
    feat = (torch.randn(n_rows, 24, generator=torch.Generator().manual_seed(3)) * 3.0).cuda()
    n = n_rows - 29
    blob = net._weights_on(feat.device)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    want_p = torch.empty(n, 2, device='cuda')
    want_d = torch.empty(n, dtype=torch.uint8, device='cuda')
    _lib.check(L.vad_gru_forward_f32(ptr(feat), n, 24, ptr(blob), hidden, int(large), ptr(want_p), ptr(want_d), 0,
                                     ctypes.c_float(0.0), _lib.current_stream()))
    with torch.no_grad():
        y = ref(torch.from_numpy(pipeline_ref.windows(feat.cpu().numpy()))).numpy()
    assert np.abs(want_p.cpu().numpy() - y).max() <= ATOL
    row_bytes = 3 * ((hidden + 31) // 32) * 32 * 4
    full = L.vad_gru_workspace_bytes(hidden, n)
    assert full == n_rows * row_bytes
    for ws_bytes in (full, full + 4096, 300 * row_bytes + 100, 31 * row_bytes, 29 * row_bytes, 0):
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device='cuda')
        got_p = torch.full((n, 2), float('nan'), device='cuda')
        got_d = torch.full((n,), 7, dtype=torch.uint8, device='cuda')
        _lib.check(L.vad_gru_forward_ws_f32(ptr(feat), n, 24, ptr(blob), hidden, int(large), ptr(got_p), ptr(got_d), 0,
                                            ctypes.c_float(0.0), ptr(ws) if ws_bytes else None, ws_bytes,
                                            _lib.current_stream()))
        assert torch.equal(got_p, want_p) and torch.equal(got_d, want_d), ws_bytes
    # other strides and the reference's 30-unit model ignore the workspace
    ws = torch.empty(full, dtype=torch.uint8, device='cuda')
    x = feat[: 30 * 33].reshape(33, 30, 24).contiguous()
    a, b = torch.empty(33, 2, device='cuda'), torch.empty(33, 2, device='cuda')
    _lib.check(L.vad_gru_forward_f32(ptr(x), 33, 720, ptr(blob), hidden, int(large), ptr(a), None, 0,
                                     ctypes.c_float(0.0), _lib.current_stream()))
    _lib.check(L.vad_gru_forward_ws_f32(ptr(x), 33, 720, ptr(blob), hidden, int(large), ptr(b), None, 0,
                                        ctypes.c_float(0.0), ptr(ws), full, _lib.current_stream()))
    assert torch.equal(a, b)
    assert L.vad_gru_workspace_bytes(30, 1000) == 0


def test_lstm_wide_hidden_is_rejected(vad):
    net = vad.Net(large=False, lstm=True, hidden=64)
    with pytest.raises(vad.VadError):
        net(torch.zeros(4, 30, 24))


def test_weight_updates_are_picked_up(vad):
    net = vad.Net(large=True, lstm=False)
    x = torch.randn(64, 30, 24, generator=torch.Generator().manual_seed(1)).cuda()
    p0 = net(x).clone()
    with torch.no_grad():
        net.lin2.bias.add_(torch.tensor([0.5, -0.5]))
    p1 = net(x)
    assert not torch.equal(p0, p1)


def test_bad_shapes_raise(vad):
    net = vad.Net(large=False, lstm=True)
    with pytest.raises(RuntimeError):
        net(torch.zeros(4, 29, 24))                     # 116 hidden states: the reference's view(-1, 900) fails
    with pytest.raises(RuntimeError):
        net(torch.zeros(31, 24))
    assert net(torch.zeros(30, 24)).shape == (1, 2)     # 30 one-step sequences: one row in the reference too
    with pytest.raises(RuntimeError):
        vad.DenseNet()(torch.zeros(30, 24))


def test_model_stack_names(vad):
    assert list(vad.MODEL_STACK) == ['net', 'net_large', 'densenet', 'densenet_large']
    assert vad.MODEL_STACK['net']['model'].lstm and not vad.MODEL_STACK['net']['model'].large
    assert sum(p.numel() for p in vad.MODEL_STACK['net_large']['model'].parameters()) == 32002
    assert sum(p.numel() for p in vad.MODEL_STACK['densenet']['model'].parameters()) == 9966


def test_end_to_end_stream(vad):
    """int16 audio -> features -> posteriors -> decisions against the float64/fp32 CPU path.
    Features carry the front end's own tolerance (1e-4 relative, floor 1.0); the posteriors still
    hold the 1e-5 absolute bar end to end (measured margin: 4e-7, tools/accuracy_report.py)."""
    n = 1200
    ref = models_ref.make_model('gru_large')
    net = vad.Net(large=True, lstm=False)
    net.load_state_dict(ref.state_dict())
    stream = vad.synth_stream(0, n, seed=1337)
    feat, p, dec = vad.vad_stream(net, stream, n)
    host = synth_ref.synth_frames(0, n, seed=1337)
    feat_ref, p_ref, _ = pipeline_ref.vad_stream(ref, host)
    assert p.shape == (n - 29, 2)
    err_f = np.abs(feat.cpu().numpy() - feat_ref) / np.maximum(np.abs(feat_ref), 1.0)
    assert err_f.max() <= 1e-4
    assert np.abs(p.cpu().numpy() - p_ref).max() <= ATOL
    _decisions_agree(dec.cpu().numpy(), p_ref)


@pytest.mark.parametrize('kind', ['gru_large', 'gru_small', 'lstm_large', 'lstm_small', 'densenet', 'densenet_large'])
def test_sampled_windows_equal_materialised_batch(vad, kind):
    """score_rows (windows addressed by start row, core/generator.py:143-150) is bit-identical to scoring
    the stacked (B,30,24) batch, and matches the oracle."""
    ref = models_ref.make_model(kind)
    net = _gpu_model(vad, kind, ref)
    rng = np.random.default_rng(12)
    feat = torch.tensor(rng.normal(0, 8, size=(700, 24)).astype(np.float32)).cuda()
    rows = np.r_[0, 670, rng.integers(0, 671, size=197), 670, 0]           # repeats, both ends, ragged count
    p, d = net.score_rows(feat, rows)
    batch = torch.stack([feat[r: r + 30] for r in rows]).contiguous()
    pb, db = net.score(batch)
    assert torch.equal(p, pb) and torch.equal(d, db)
    with torch.no_grad():
        y = ref(batch.cpu()).numpy()
    assert np.abs(p.cpu().numpy() - y).max() <= ATOL
    with pytest.raises(IndexError):
        net.score_rows(feat, [671])
    with pytest.raises(IndexError):
        net.score_rows(feat, [-1])
    p0, d0 = net.score_rows(feat, [])
    assert p0.shape == (0, 2) and d0.shape == (0,)


def test_generator_feeds_models_without_batches(vad):
    """test_predict's loop (core/train.py:416-440) with the batch tensor replaced by window rows."""
    from voice_activity_detection_amd.generator import DataGenerator
    rng = np.random.default_rng(3)
    n = 12000
    data = {'labels': (np.sin(np.arange(n) / 40.0) > 0).astype(np.int64)}
    for lvl in ('None', '-15', '-3'):
        data['frames-' + lvl] = np.zeros((n, 1), np.int16)
        data['mfcc-' + lvl] = rng.normal(0, 8, size=(n, 12)).astype(np.float32)
        data['delta-' + lvl] = rng.normal(0, 2, size=(n, 12)).astype(np.float32)
    ref = models_ref.make_model('gru_large')
    net = _gpu_model(vad, 'gru_large', ref)
    feat = torch.tensor(np.hstack((data['mfcc--3'], data['delta--3']))).cuda()
    scores = {}
    for form in ('batch', 'rows'):
        g = DataGenerator(data)
        g.setup_generation(30, 6, 64)
        g.use_test_data()
        g.set_noise_level_db('-3')
        y_true, y_score = [], []
        for i in range(g.batch_count):
            if form == 'batch':
                X, y = g.get_batch(i)
                out = net(torch.from_numpy(np.array(X)).float())
            else:
                rows, y = g.get_batch_rows(i)
                out = net.score_rows(feat, rows)[0]
            y_true.extend(y)
            y_score.extend(out.cpu().numpy()[:, 1])
        scores[form] = (np.array(y_true), np.array(y_score))
    np.testing.assert_array_equal(scores['batch'][0], scores['rows'][0])
    np.testing.assert_array_equal(scores['batch'][1], scores['rows'][1])
    assert len(scores['rows'][1]) > 0


@pytest.mark.parametrize('mode', ['bf16x8', 'bf16x9', 'bf16x6', 'f16x6', 'f16x4'])
@pytest.mark.parametrize('kind', ['gru_large', 'gru_small', 'lstm_large', 'lstm_small'])
def test_split_bf16_matrix_modes(vad, kind, mode):
    """The stride-1 kernels with the products on the bf16 matrix pipe (float32 operands split exactly into
    three bf16) hold the same 1e-5 bar against the float32 oracle, and stay close to the float32-MFMA kernel."""
    ref = models_ref.make_model(kind)
    net = _gpu_model(vad, kind, ref)
    rng = np.random.default_rng(41)
    feat_np = rng.normal(0, 10, size=(1500, 24)).astype(np.float32)
    feat_np[200:260] = 0.0
    feat_np[700:705] *= 30.0                                   # outliers: saturated gates
    feat = torch.tensor(feat_np).cuda()
    prev = vad.set_matrix_mode('f32')
    try:
        p32, d32 = net.score(feat, window_stride=24)
        assert vad.set_matrix_mode(mode) == 'f32'
        p, d = net.score(feat, window_stride=24)
        p_again, _ = net.score(feat, window_stride=24)
    finally:
        vad.set_matrix_mode(prev)
    assert torch.equal(p, p_again)                              # deterministic
    windows = torch.stack([feat[i:i + 30] for i in range(1471)])
    with torch.no_grad():
        y = ref(windows.cpu()).numpy()
    err = np.abs(p.cpu().numpy() - y).max()
    assert err <= ATOL, err
    assert np.abs(p.cpu().numpy() - p32.cpu().numpy()).max() <= ATOL
    _decisions_agree(d.cpu().numpy(), y)
    with pytest.raises(ValueError):
        vad.set_matrix_mode('fp8')


@pytest.mark.parametrize('kind', ['gru_large', 'gru_small', 'lstm_large', 'lstm_small'])
def test_split_kernels_at_ragged_window_counts(vad, kind):
    """Window counts around the kernels' 16-window halves, 32-window waves and 192 / 256-window blocks: every
    window scored once, within the bar of the oracle, and nothing written past the last window."""
    ref = models_ref.make_model(kind)
    net = _gpu_model(vad, kind, ref)
    rng = np.random.default_rng(7)
    feat_np = rng.normal(0, 6, size=(600, 24)).astype(np.float32)
    feat = torch.tensor(feat_np).cuda()
    windows = torch.stack([torch.from_numpy(feat_np[i:i + 30]) for i in range(571)])
    with torch.no_grad():
        y = ref(windows).numpy()
    for mode in ('bf16x8', 'f16x6'):
        for n in (1, 15, 16, 17, 31, 32, 33, 191, 192, 193, 255, 256, 257, 571):
            p = torch.full((n + 3, 2), -7.0, device='cuda')
            d = torch.full((n + 3,), 9, dtype=torch.uint8, device='cuda')
            net.score(feat[:n + 29], window_stride=24, n_windows=n, out=p[:n], decisions=d[:n], matrix_mode=mode)
            assert np.abs(p[:n].cpu().numpy() - y[:n]).max() <= ATOL, (mode, n)
            assert bool((p[n:] == -7.0).all()) and bool((d[n:] == 9).all()), (mode, n)
            assert bool((d[:n] <= 1).all()), (mode, n)


@pytest.mark.parametrize('mode', ['bf16x8', 'bf16x9', 'bf16x6', 'f16x6', 'f16x4'])
@pytest.mark.parametrize('hidden,large', [(90, True), (90, False), (64, True), (40, False), (33, True)])
def test_wide_gru_split_kernel(vad, hidden, large, mode):
    """Stride-1 windows of the wide GRU with the products on the bf16 matrix pipe: within the bar of the float32
    oracle and of the float32-MFMA kernel, at window counts around the kernel's 16 / 32 / 128-window boundaries,
    through a workspace that forces several chunks, and without writing past the last window."""
    ref = models_ref.make_model('gru_large' if large else 'gru_small', hidden=hidden)
    net = vad.Net(large=large, lstm=False, hidden=hidden)
    net.load_state_dict(ref.state_dict())
    rng = np.random.default_rng(hidden)
    feat_np = rng.normal(0, 6, size=(460, 24)).astype(np.float32)
    feat_np[100:130] = 0.0
    feat_np[300:303] *= 25.0
    feat = torch.tensor(feat_np).cuda()
    prev = vad.set_matrix_mode('f32')
    try:
        p32, _ = net.score(feat, window_stride=24)
        vad.set_matrix_mode(mode)
        for n in (1, 15, 16, 17, 31, 32, 33, 127, 128, 129, 431):
            p = torch.full((n + 3, 2), -7.0, device='cuda')
            d = torch.full((n + 3,), 9, dtype=torch.uint8, device='cuda')
            net.score(feat[:n + 29], window_stride=24, n_windows=n, out=p[:n], decisions=d[:n])
            assert float((p[:n] - p32[:n]).abs().max()) <= ATOL, n
            assert bool((p[n:] == -7.0).all()) and bool((d[n:] == 9).all()), n
            assert bool((d[:n] <= 1).all()), n
        cap = type(net).WORKSPACE_CAP
        try:
            type(net).WORKSPACE_CAP = 100 * 3 * ((hidden + 31) // 32) * 32 * 4       # 100 rows: 71 windows per chunk
            net._ws.clear()
            p_chunks, _ = net.score(feat, window_stride=24)
        finally:
            type(net).WORKSPACE_CAP = cap
            net._ws.clear()
        p_whole, _ = net.score(feat, window_stride=24)
        assert torch.equal(p_chunks, p_whole)
    finally:
        vad.set_matrix_mode(prev)
    windows = torch.stack([torch.from_numpy(feat_np[i:i + 30]) for i in range(431)])
    with torch.no_grad():
        y = ref(windows).numpy()
    assert np.abs(p_whole.cpu().numpy() - y).max() <= ATOL


def test_calls_follow_the_current_stream(vad):
    """Every entry point launches on the caller's stream: two side streams working on different inputs at
    the same time give the results of the default stream (the library keeps no mutable device state)."""
    ref = models_ref.make_model('gru_large')
    net = _gpu_model(vad, 'gru_large', ref)
    n = 20000
    a = vad.synth_stream(0, n, seed=1)
    b = vad.synth_stream(0, n, seed=2)
    fa, pa, da = vad.vad_stream(net, a, n)
    fb, pb, db = vad.vad_stream(net, b, n)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = {}
    for _ in range(3):
        with torch.cuda.stream(s1):
            outs['a'] = vad.vad_stream(net, a, n)
        with torch.cuda.stream(s2):
            outs['b'] = vad.vad_stream(net, b, n)
    s1.synchronize()
    s2.synchronize()
    assert torch.equal(outs['a'][0], fa) and torch.equal(outs['a'][1], pa) and torch.equal(outs['a'][2], da)
    assert torch.equal(outs['b'][0], fb) and torch.equal(outs['b'][1], pb) and torch.equal(outs['b'][2], db)


def test_init_is_idempotent(vad):
    vad.init()
    vad.init()


def test_path_is_capturable_in_a_hip_graph(vad):
    """After init() the two launches of a step allocate nothing and synchronise nothing, so the step can be
    captured once and replayed: the replay on new audio equals a direct run."""
    ref = models_ref.make_model('gru_large')
    net = _gpu_model(vad, 'gru_large', ref)
    n = 166 * 60
    vad.init()
    stream = vad.synth_stream(0, n, seed=3)
    ws = vad.StreamWorkspace(n, stream.device)
    vad.vad_stream(net, stream, n, workspace=ws)                 # packs the weights outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        vad.mfcc_delta_stream(stream, n, 166, out=ws.feat)
        net.score(ws.feat, window_stride=24, n_windows=n - 29, out=ws.p, decisions=ws.decision)
    stream.copy_(vad.synth_stream(0, n, seed=4))                 # new audio in the captured buffer
    g.replay()
    torch.cuda.synchronize()
    p_graph, d_graph, f_graph = ws.p.clone(), ws.decision.clone(), ws.feat.clone()
    ws2 = vad.StreamWorkspace(n, stream.device)
    vad.vad_stream(net, vad.synth_stream(0, n, seed=4), n, workspace=ws2)
    assert torch.equal(f_graph, ws2.feat) and torch.equal(p_graph, ws2.p) and torch.equal(d_graph, ws2.decision)


@pytest.mark.parametrize('hidden', [90, 40])
def test_sampled_windows_wide_gru(vad, hidden):
    ref = models_ref.make_model('gru_large', hidden=hidden)
    net = vad.Net(large=True, lstm=False, hidden=hidden)
    net.load_state_dict(ref.state_dict())
    rng = np.random.default_rng(hidden)
    feat = torch.tensor(rng.normal(0, 8, size=(300, 24)).astype(np.float32)).cuda()
    rows = rng.integers(0, 271, size=77)
    p, d = net.score_rows(feat, rows)
    batch = torch.stack([feat[r: r + 30] for r in rows]).contiguous()
    pb, db = net.score(batch)
    assert torch.equal(p, pb) and torch.equal(d, db)
    with torch.no_grad():
        y = ref(batch.cpu()).numpy()
    assert np.abs(p.cpu().numpy() - y).max() <= ATOL


def test_wide_gru_scratch_is_per_stream(vad):
    """A GRU with more than 32 units projects its feature rows into scratch memory before the steps read them.
    The module keeps one scratch buffer per (device, stream), so the same net scoring different inputs on two
    streams at once gives the results of one-at-a-time runs, in the float32 and in the split mode; a caller's
    own workspace tensor does the same."""
    ref = models_ref.make_model('gru_large', hidden=90)
    net = vad.Net(large=True, lstm=False, hidden=90)
    net.load_state_dict(ref.state_dict())
    n = 30000
    rng = np.random.default_rng(11)
    fa = torch.tensor(rng.normal(0, 8, size=(n, 24)).astype(np.float32)).cuda()
    fb = torch.tensor(rng.normal(0, 8, size=(n, 24)).astype(np.float32)).cuda()
    for mm in ('f32', 'bf16x8'):
        pa, _ = net.score(fa, window_stride=24, matrix_mode=mm)
        pb, _ = net.score(fb, window_stride=24, matrix_mode=mm)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = {}
        for _ in range(4):
            with torch.cuda.stream(s1):
                outs['a'] = net.score(fa, window_stride=24, matrix_mode=mm)[0]
            with torch.cuda.stream(s2):
                outs['b'] = net.score(fb, window_stride=24, matrix_mode=mm)[0]
        s1.synchronize()
        s2.synchronize()
        assert torch.equal(outs['a'], pa) and torch.equal(outs['b'], pb), mm
        own = torch.empty(net.workspace_bytes(n - 29), dtype=torch.uint8, device='cuda')
        assert torch.equal(net.score(fa, window_stride=24, matrix_mode=mm, workspace=own)[0], pa)
    assert len(net._ws) >= 3                           # default stream + the two side streams
    with pytest.raises(ValueError):
        net.score(fa, window_stride=24, workspace=torch.empty(64, dtype=torch.float32, device='cuda'))


def test_wide_gru_is_capturable_after_reserving_its_scratch(vad):
    """reserve_workspace() moves the wide GRU's one allocation out of a graph capture."""
    ref = models_ref.make_model('gru_large', hidden=90)
    net = vad.Net(large=True, lstm=False, hidden=90)
    net.load_state_dict(ref.state_dict())
    n = 5000
    rng = np.random.default_rng(12)
    feat = torch.tensor(rng.normal(0, 8, size=(n, 24)).astype(np.float32)).cuda()
    p_direct, d_direct = net.score(feat, window_stride=24)
    p = torch.empty_like(p_direct)
    d = torch.empty_like(d_direct)
    vad.init()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ws = torch.empty(net.workspace_bytes(n - 29), dtype=torch.uint8, device='cuda')
    with torch.cuda.graph(g, stream=side):
        net.score(feat, window_stride=24, out=p, decisions=d, workspace=ws)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(p, p_direct) and torch.equal(d, d_direct)
    assert net.reserve_workspace(n - 29).numel() >= net.workspace_bytes(n - 29)


def test_matrix_mode_is_a_property_of_the_call(vad):
    """Two Nets of one process score the same features in different matrix modes at the same time on two streams
    and each gets what a run in its own mode alone gives; the process-wide mode is only the default."""
    ref = models_ref.make_model('gru_large')
    n = 40000
    rng = np.random.default_rng(13)
    feat = torch.tensor(rng.normal(0, 8, size=(n, 24)).astype(np.float32)).cuda()
    expected = {}
    for mm in ('f32', 'bf16x8'):
        prev = vad.set_matrix_mode(mm)
        plain = vad.Net(large=True, lstm=False)
        plain.load_state_dict(ref.state_dict())
        expected[mm] = plain.score(feat, window_stride=24)[0].clone()
        vad.set_matrix_mode(prev)
    assert not torch.equal(expected['f32'], expected['bf16x8'])
    assert float((expected['f32'] - expected['bf16x8']).abs().max()) <= 1e-5
    strict = vad.Net(large=True, lstm=False, matrix_mode='f32')
    fast = vad.Net(large=True, lstm=False, matrix_mode='bf16x8')
    strict.load_state_dict(ref.state_dict())
    fast.load_state_dict(ref.state_dict())
    strict.score(feat[:64], window_stride=24)              # pack the weights outside the overlapped part
    fast.score(feat[:64], window_stride=24)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for globally in ('bf16x6', 'f32'):                         # whatever the process-wide mode says
        prev = vad.set_matrix_mode(globally)
        for _ in range(3):
            with torch.cuda.stream(s1):
                ps = strict.score(feat, window_stride=24)[0]
            with torch.cuda.stream(s2):
                pf = fast.score(feat, window_stride=24)[0]
        s1.synchronize()
        s2.synchronize()
        vad.set_matrix_mode(prev)
        assert torch.equal(ps, expected['f32']) and torch.equal(pf, expected['bf16x8'])
    # strict float32 stride-1 windows are the bits of the dense-batch forward()
    idx = torch.arange(0, 2048, device='cuda')[:, None] + torch.arange(30, device='cuda')[None, :]
    assert torch.equal(strict(feat[idx]), expected['f32'][:2048])
    # per call beats per model; LSTM takes the argument too
    assert torch.equal(strict.score(feat, window_stride=24, matrix_mode='bf16x8')[0], expected['bf16x8'])
    lref = models_ref.make_model('lstm_small')
    lstm = vad.Net(large=False, lstm=True)
    lstm.load_state_dict(lref.state_dict())
    a = lstm.score(feat, window_stride=24, matrix_mode='f32')[0]
    b = lstm.score(feat, window_stride=24, matrix_mode='bf16x9')[0]
    assert not torch.equal(a, b) and float((a - b).abs().max()) <= 1e-5
    with pytest.raises(ValueError):
        lstm.score(feat, window_stride=24, matrix_mode='fp8')


def test_float16_modes_take_outsized_weights(vad):
    """The float16 tables ride on a power-of-two scale chosen per model, so a recurrent weight far beyond float16's
    own range (65504) still goes through the float16 matrix modes: same bar against the float32 kernel."""
    ref = models_ref.make_model('gru_large')
    with torch.no_grad():
        ref.rnn.weight_hh_l0[3, 5] = 7.0e4
        ref.rnn.weight_hh_l0[40, 2] = -3.0e-7
    net = _gpu_model(vad, 'gru_large', ref)
    rng = np.random.default_rng(3)
    feat = torch.tensor(rng.normal(0, 8, size=(500, 24)).astype(np.float32)).cuda()
    p32, _ = net.score(feat, window_stride=24, matrix_mode='f32')
    for mode in ('f16x6', 'f16x4', 'bf16x8'):
        p, _ = net.score(feat, window_stride=24, matrix_mode=mode)
        assert bool(torch.isfinite(p).all()) and float((p - p32).abs().max()) <= ATOL, mode


@pytest.mark.parametrize('kind', list(NET_KINDS))
def test_forward_takes_every_input_shape_the_reference_does(vad, golden_dir, kind):
    """Net.forward on inputs that are not (B,30,24) windows, against outputs of the reference itself
    (tests/golden/make_seq_fixtures.py): 2-D rows as one-step sequences, sequences of 15 and 45 steps re-cut into
    rows of 900 hidden values, and RuntimeError where the reference's view() raises."""
    sd, _, _ = _fixture(golden_dir, kind)
    fx = np.load(os.path.join(golden_dir, 'model_seq_shapes.npz'))
    net = _make(vad, kind)
    net.load_state_dict(sd)
    for name in ('x2d', 'x15', 'x45', 'x1'):
        want = fx['%s.%s' % (kind, name)]
        got = net(torch.from_numpy(fx[name])).cpu().numpy()
        assert got.shape == want.shape, name
        assert np.abs(got - want).max() <= ATOL, (name, np.abs(got - want).max())
    # the window path and the sequence path agree on windows given as 30-step sequences of another batch shape
    x = torch.from_numpy(fx['x45'][:, :30].copy())
    assert np.abs(net(x).cpu().numpy() - net(x.reshape(4, 15, 24)).cpu().numpy()).max() > 0     # different cuts differ
    for bad in (torch.zeros(24), torch.zeros(1, 7, 24), torch.zeros(2, 30, 23)):
        with pytest.raises(RuntimeError):
            net(bad)
    assert net(torch.zeros(0, 5, 24)).shape == (0, 2)
