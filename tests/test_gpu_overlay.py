"""GPU parity of vad_overlay_i16 / add_noise against the CPU restatement: bit-exact int16."""
import numpy as np
import pytest
import torch

from oracle import mfcc_ref, overlay_ref, synth_ref

pytestmark = pytest.mark.gpu

NOISE_LEVELS = [None, -15, -3]                 # core/common.py:25


@pytest.mark.parametrize('gain', NOISE_LEVELS + [0, 6, -60, 0.5])
def test_overlay_bit_exact(vad, gain):
    rng = np.random.default_rng(11)
    noise = rng.integers(-32768, 32768, size=(50, 480)).astype(np.int16)
    speech = rng.integers(-32768, 32768, size=(50, 480)).astype(np.int16)
    noise[0, :8] = [32767, -32768, 32767, -32768, 0, 1, -1, 12345]
    speech[0, :8] = [32767, -32768, -32768, 32767, 0, -1, 1, -12345]
    got = vad.overlay(noise, speech, gain)
    np.testing.assert_array_equal(got, overlay_ref.overlay(noise, speech, gain))
    assert got.dtype == np.int16 and got.shape == (50, 480)


@pytest.mark.parametrize('n_noise,n_speech,offset', [(4801, 4801, 0), (4800, 1000, 0), (1000, 4800, 0), (4807, 4807, 3),
                                                     (7, 7, 1), (4800, 0, 0)])
def test_overlay_ragged_and_unaligned(vad, n_noise, n_speech, offset):
    """Lengths that are not multiples of 8, unequal tracks, and pointers off the 16-byte grid."""
    rng = np.random.default_rng(n_noise + n_speech)
    noise = rng.integers(-32768, 32768, size=n_noise + offset).astype(np.int16)
    speech = rng.integers(-32768, 32768, size=n_speech + offset).astype(np.int16)
    nd = torch.from_numpy(noise).cuda()[offset:]
    sd = torch.from_numpy(speech).cuda()[offset:]
    out = torch.full((n_noise + offset,), 77, dtype=torch.int16, device='cuda')
    vad.overlay_device(nd, sd, -15, out=out[offset:])
    ref = overlay_ref.overlay(noise[offset:], speech[offset:], -15).reshape(-1)
    np.testing.assert_array_equal(out[offset:].cpu().numpy(), ref)
    assert bool((out[:offset] == 77).all())


def test_overlay_in_place_and_large(vad):
    n = 480 * 20000                                # 10 min
    noise = vad.synth_audio(0, n, seed=5)
    speech = vad.synth_audio(10 ** 7, n, seed=6)
    ref = overlay_ref.overlay(noise.cpu().numpy(), speech.cpu().numpy(), -3).reshape(-1)
    vad.overlay_device(noise, speech, -3, out=noise)
    np.testing.assert_array_equal(noise.cpu().numpy(), ref)


@pytest.mark.parametrize('gain', NOISE_LEVELS)
def test_add_noise_matches_oracle(vad, gain):
    speech = synth_ref.synth_frames(0, 60, seed=21)
    noise = synth_ref.synth_frames(5000, 60, seed=22)
    align = synth_ref.synth_frames(9000, 3, seed=23)
    frames, m, d = vad.add_noise(speech, noise, align, gain)
    mixed = overlay_ref.overlay(noise, speech, gain)
    np.testing.assert_array_equal(frames, mixed)
    _, rm, rd = mfcc_ref.process_test_frames(mixed, align)
    for got, ref in ((m, rm), (d, rd)):
        assert got.shape == (60, 12)
        assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)).max() <= 1e-4


def test_overlay_argument_errors(vad):
    a = torch.zeros(480, dtype=torch.int16, device='cuda')
    with pytest.raises(ValueError):
        vad.overlay_device(a.float(), a)
    with pytest.raises(ValueError):
        vad.overlay_device(a, a, out=torch.zeros(10, dtype=torch.int16, device='cuda'))
