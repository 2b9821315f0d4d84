"""NickNet (conv trunk + GRU + embed head) on the GPU against the vectors captured from the reference's own
sub-modules (tests/golden/make_nicknet_fixtures.py) and against the CPU oracle (oracle/nicknet_ref.py).

Tolerance: 1e-5 absolute on trunk activations, GRU states (all in (-1, 1)) and posteriors; the embed logits of
the seeded random rows are compared at 1e-5 relative to the largest logit.  The whole-model posterior has no
reference counterpart (the reference's forward raises, core/models.py:166-174): `y_glue` is the reference's
sub-modules composed with the documented head input."""
import os

import numpy as np
import pytest
import torch

from oracle import nicknet_ref
from oracle.models_ref import load_state_dict_numpy

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _fixture(vad, golden_dir, size):
    fx = np.load(os.path.join(golden_dir, 'nicknet_%s.npz' % size))
    x = torch.from_numpy(np.load(os.path.join(golden_dir, 'model_inputs.npz'))['x'])
    net = load_state_dict_numpy(vad.NickNet(large=(size == 'large')), fx)
    return fx, x, net


@pytest.mark.parametrize('size', ['large', 'small'])
def test_trunk_gru_and_embed_match_the_reference_submodules(vad, golden_dir, size):
    fx, x, net = _fixture(vad, golden_dir, size)
    got = net.stages(x=x)
    torch.cuda.synchronize()
    assert np.abs(got['trunk'].cpu().numpy() - fx['trunk']).max() <= ATOL            # conv1 .. conv4
    assert np.abs(got['seq'].cpu().numpy() - fx['seq']).max() <= ATOL                # GRU outputs, end to end
    assert np.abs(got['hn'].cpu().numpy() - fx['hn']).max() <= ATOL
    assert np.abs(got['p'].cpu().numpy() - fx['y_glue']).max() <= ATOL
    # the GRU on its own, started from the reference's trunk output
    alone = net.stages(trunk=torch.from_numpy(fx['trunk']))
    assert 'trunk' not in alone
    assert np.abs(alone['seq'].cpu().numpy() - fx['seq']).max() <= ATOL
    assert np.abs(alone['hn'].cpu().numpy() - fx['hn']).max() <= ATOL
    # seq's last step is the last layer's final state
    assert torch.equal(alone['seq'][:, -1], alone['hn'][-1])
    # embed on its own, on the seeded rows the reference's embed was run on
    ev = net.embed(torch.from_numpy(fx['v'])).cpu().numpy()
    assert np.abs(ev - fx['ev']).max() <= ATOL * max(1.0, np.abs(fx['ev']).max())
    # forward() = the stages' posterior
    y = net(x)
    assert y.shape == (64, 2) and torch.equal(y, got['p'])
    assert float((y.sum(1) - 1).abs().max()) < 1e-6


@pytest.mark.parametrize('size', ['large', 'small'])
@pytest.mark.parametrize('batch', [1, 15, 16, 17, 127, 129, 2048])
def test_ragged_batches_match_oracle(vad, size, batch):
    large = size == 'large'
    ref = nicknet_ref.make_nicknet(large)
    net = vad.NickNet(large=large)
    net.load_state_dict(ref.state_dict())
    rng = np.random.default_rng(batch)
    x = torch.from_numpy(rng.normal(0, 6, size=(batch, 30, 24)).astype(np.float32))
    with torch.no_grad():
        y_ref = ref(x).numpy()
    p = torch.full((batch + 5, 2), -7.0, device='cuda')
    d = torch.full((batch + 5,), 9, dtype=torch.uint8, device='cuda')
    net.score(x.cuda(), out=p[:batch], decisions=d[:batch])
    torch.cuda.synchronize()
    assert np.abs(p[:batch].cpu().numpy() - y_ref).max() <= ATOL
    assert bool((p[batch:] == -7.0).all()) and bool((d[batch:] == 9).all())
    assert bool(((p[:batch, 1] > p[:batch, 0]).to(torch.uint8) == d[:batch]).all())


@pytest.mark.parametrize('size', ['large', 'small'])
def test_sliding_and_sampled_windows_equal_explicit_windows(vad, size):
    large = size == 'large'
    ref = nicknet_ref.make_nicknet(large)
    net = vad.NickNet(large=large)
    net.load_state_dict(ref.state_dict())
    rng = np.random.default_rng(5)
    n = 5000
    feat = torch.from_numpy(rng.normal(0, 6, size=(n, 24)).astype(np.float32)).cuda()
    p_slide, d_slide = net.score(feat, window_stride=24)
    assert p_slide.shape == (n - 29, 2)
    idx = torch.arange(n - 29, device='cuda')[:, None] + torch.arange(30, device='cuda')[None, :]
    p_batch, d_batch = net.score(feat[idx].contiguous())
    assert torch.equal(p_slide, p_batch) and torch.equal(d_slide, d_batch)
    rows = torch.from_numpy(rng.integers(0, n - 29, size=777))
    p_rows, d_rows = net.score_rows(feat, rows)
    assert torch.equal(p_rows, p_batch[rows.cuda()]) and torch.equal(d_rows, d_batch[rows.cuda()])
    with torch.no_grad():
        y = ref(feat[idx[:300]].cpu()).numpy()
    assert np.abs(p_slide[:300].cpu().numpy() - y).max() <= ATOL
    # thresholded decisions: p1 >= t in float32 (core/train.py:486-487)
    t = float(p_slide[:, 1].median())
    _, d_thr = net.score(feat, window_stride=24, threshold=t)
    assert torch.equal(d_thr, (p_slide[:, 1] >= np.float32(t)).to(torch.uint8))


@pytest.mark.parametrize('size', ['large', 'small'])
@pytest.mark.parametrize('n_rows', [30, 31, 45, 46, 61, 77, 158, 2077])
def test_stream_kernel_is_bit_identical_at_ragged_sizes(vad, size, n_rows):
    """Stride-1 windows take the kernel that evaluates the trunk once per row (interior positions) plus the
    window-specific edge positions; dense batches take the per-window kernel.  Same instruction chains on the same
    operands: equal bits, for every window count around the 16-window tiles and the 48-row span of a wave, with
    guard elements behind the outputs."""
    large = size == 'large'
    ref = nicknet_ref.make_nicknet(large)
    net = vad.NickNet(large=large)
    net.load_state_dict(ref.state_dict())
    rng = np.random.default_rng(n_rows)
    feat = torch.from_numpy(rng.normal(0, 6, size=(n_rows, 24)).astype(np.float32)).cuda()
    nw = n_rows - 29
    p = torch.full((nw + 7, 2), -7.0, device='cuda')
    d = torch.full((nw + 7,), 9, dtype=torch.uint8, device='cuda')
    net.score(feat, window_stride=24, out=p[:nw], decisions=d[:nw])
    idx = torch.arange(nw, device='cuda')[:, None] + torch.arange(30, device='cuda')[None, :]
    p_batch, d_batch = net.score(feat[idx].contiguous())
    assert torch.equal(p[:nw], p_batch) and torch.equal(d[:nw], d_batch)
    assert bool((p[nw:] == -7.0).all()) and bool((d[nw:] == 9).all())
    with torch.no_grad():
        y = ref(feat[idx[:64]].cpu()).numpy()
    assert np.abs(p[:min(nw, 64)].cpu().numpy() - y).max() <= ATOL


def test_saturated_inputs_stay_finite(vad):
    """Large pre-activations saturate tanh / sigmoid (exp2 overflows to inf, its reciprocal is 0): no NaN."""
    ref = nicknet_ref.make_nicknet(True)
    net = vad.NickNet(large=True)
    net.load_state_dict(ref.state_dict())
    rng = np.random.default_rng(9)
    x = torch.from_numpy((rng.normal(0, 6, size=(64, 30, 24)) * rng.choice([1, 1, 50, 2000], size=(64, 30, 1)))
                         .astype(np.float32))
    with torch.no_grad():
        y_ref = ref(x).numpy()
    y = net(x).cpu().numpy()
    assert np.isfinite(y).all() and np.abs(y - y_ref).max() <= ATOL


def test_bad_arguments_raise(vad):
    net = vad.NickNet(large=False)
    with pytest.raises(RuntimeError):
        net(torch.zeros(4, 24, 30))
    with pytest.raises(ValueError):
        net.stages()
    with pytest.raises(ValueError):
        net.stages(trunk=torch.zeros(4, 18, 30))              # the small model's trunk has 16 channels
    with pytest.raises(ValueError):
        net.embed(torch.zeros(4, 58))
    assert net.stages(x=torch.zeros(0, 30, 24))['p'].shape == (0, 2)
