"""vad_threshold_sweep_noalloc_f32 -- the threshold sweep that takes no workspace and allocates nothing (its
histogram is an array built into the library) -- against vad_threshold_sweep_f32 through the C ABI:
integer-exact equality, on two calls in a row of which the second has the smaller input, so that counts
left in the shared array by the first call would show."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(n, n_thr, seed):
    rng = np.random.default_rng(seed)
    y = (rng.random(n) < 0.4)
    z = rng.normal(0, 3, n) + np.where(y, 4.0, -6.0)
    s = (1.0 / (1.0 + np.exp(-z))).astype(np.float32)
    thr = np.sort(rng.random(n_thr))
    return (torch.from_numpy(s).cuda(), torch.from_numpy(y.astype(np.uint8)).cuda(), torch.from_numpy(thr).cuda())


def _both(L, stream, s, lab, thr):
    n, T = s.numel(), thr.numel()
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    hist = torch.empty(2 * (T + 1), dtype=torch.int64, device='cuda')
    ge_a = torch.full((T, 2), -1, dtype=torch.int64, device='cuda')
    tot_a = torch.full((2,), -1, dtype=torch.int64, device='cuda')
    ge_b, tot_b = ge_a.clone(), tot_a.clone()
    assert L.vad_threshold_sweep_f32(ptr(s), 1, ptr(lab), n, ptr(thr), T, ptr(hist), ptr(ge_a), ptr(tot_a), stream) == 0
    assert L.vad_threshold_sweep_noalloc_f32(ptr(s), 1, ptr(lab), n, ptr(thr), T, ptr(ge_b), ptr(tot_b), stream) == 0
    torch.cuda.synchronize()
    return (ge_a.cpu().numpy(), tot_a.cpu().numpy()), (ge_b.cpu().numpy(), tot_b.cpu().numpy())


def test_noalloc_sweep_equals_the_workspace_sweep_twice_in_a_row(vad):
    from voice_activity_detection_amd import _lib
    L = vad.lib()
    vad.init()
    stream = _lib.current_stream()
    # first call: large input, long table; second call: smaller input, shorter table (stale bins beyond its
    # table and stale counts inside it would both change the result)
    for n, T, seed in ((3_000_017, 2048, 1), (70_001, 57, 2)):
        s, lab, thr = _inputs(n, T, seed)
        (ge_a, tot_a), (ge_b, tot_b) = _both(L, stream, s, lab, thr)
        np.testing.assert_array_equal(ge_b, ge_a)
        np.testing.assert_array_equal(tot_b, tot_a)
        assert int(tot_a.sum()) == n and int(ge_a.min()) >= 0
        # and against numpy on the host
        sh, yh, th = s.cpu().numpy().astype(np.float64), lab.cpu().numpy(), thr.cpu().numpy()
        for k in (0, T // 2, T - 1):
            assert ge_b[k, 1] == int(np.sum((sh >= th[k]) & (yh == 1)))
            assert ge_b[k, 0] == int(np.sum((sh >= th[k]) & (yh == 0)))


def test_noalloc_sweep_strided_scores_empty_input_and_limits(vad):
    from voice_activity_detection_amd import _lib
    L = vad.lib()
    stream = _lib.current_stream()
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    s, lab, thr = _inputs(5000, 9, 3)
    p = torch.stack([1 - s, s], 1).contiguous()                       # column 1 of an (n,2) posterior array in place
    ge = torch.empty((9, 2), dtype=torch.int64, device='cuda')
    tot = torch.empty(2, dtype=torch.int64, device='cuda')
    assert L.vad_threshold_sweep_noalloc_f32(ctypes.c_void_p(p.data_ptr() + 4), 2, ptr(lab), 5000, ptr(thr), 9,
                                             ptr(ge), ptr(tot), stream) == 0
    (ge_a, tot_a), _ = _both(L, stream, s, lab, thr)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ge.cpu().numpy(), ge_a)
    np.testing.assert_array_equal(tot.cpu().numpy(), tot_a)
    # n = 0: everything zero, whatever the previous call left behind
    assert L.vad_threshold_sweep_noalloc_f32(None, 1, None, 0, ptr(thr), 9, ptr(ge), ptr(tot), stream) == 0
    torch.cuda.synchronize()
    assert int(ge.abs().sum()) == 0 and int(tot.abs().sum()) == 0
    # a table beyond the built-in array's size is refused with a message, nothing is launched
    big = torch.linspace(0, 1, 2049, dtype=torch.float64, device='cuda')
    ge_big = torch.empty((2049, 2), dtype=torch.int64, device='cuda')
    assert L.vad_threshold_sweep_noalloc_f32(ptr(s), 1, ptr(lab), 5000, ptr(big), 2049, ptr(ge_big), ptr(tot), stream) == -1
    assert b'n_thresholds' in L.vad_last_error()
