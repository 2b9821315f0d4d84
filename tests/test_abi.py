"""The C-ABI library loads on a GPU-less host and exports every symbol include/vad_amd.h declares.
No compute entry point is called here."""
import os
import re

import numpy as np
import pytest

import voice_activity_detection_amd as pkg
from voice_activity_detection_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    pkg.build()
    return pkg.lib()


def test_header_symbols_are_exported(L):
    header = open(os.path.join(ROOT, 'include', 'vad_amd.h')).read()
    declared = set(re.findall(r'\b(vad_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.vad_abi_version() == 1


def test_net_pack_layout(L):
    """Host-only: the packed blob places W_hh[g*H+m][u] where the kernel's A operand expects it.
    Tile rows 27 and 31 (the two rows of k-step 15) are the padding for H = 30; sigmoid gates are
    pre-scaled by -log2(e), tanh gates by 2 log2(e)."""
    H = 30
    rng = np.random.default_rng(0)
    f32 = lambda *s: rng.standard_normal(s).astype(np.float32)
    unit_of_row = lambda r: -1 if r in (27, 31) else (r - 1 if r > 27 else r)
    LOG2E = 1.4426950408889634
    for lstm in (0, 1):
        G = 4 if lstm else 3
        w_ih, w_hh, b_ih, b_hh = f32(G * H, 24), f32(G * H, H), f32(G * H), f32(G * H)
        lin1_w, lin1_b, lin2_w, lin2_b, lin_w, lin_b = f32(26, 900), f32(26), f32(2, 26), f32(2), f32(2, 900), f32(2)
        scale = [-LOG2E, -LOG2E, 2 * LOG2E, -LOG2E] if lstm else [-LOG2E, -LOG2E, 2 * LOG2E]
        for large in (0, 1):
            n = L.vad_net_blob_floats(lstm, large, H)
            blob = np.zeros(n, np.float32)
            hp = _lib.host_ptr
            rc = L.vad_pack_net_weights(lstm, large, H, hp(w_ih), hp(w_hh), hp(b_ih), hp(b_hh), hp(lin1_w),
                                        hp(lin1_b), hp(lin2_w), hp(lin2_b), hp(lin_w), hp(lin_b), hp(blob), n)
            assert rc == 0
            a_ih = blob[:G * 12 * 64].reshape(G, 12, 64)
            a_hh = blob[G * 12 * 64: G * 28 * 64].reshape(G, 16, 64)
            head = blob[G * 28 * 64 + 128: G * 28 * 64 + 128 + 30 * 16 * 64].reshape(30, 16, 64)
            hw = lin1_w if large else lin_w
            for lane in (0, 5, 26, 27, 28, 30, 31, 32, 40, 59, 63):
                m, hh = unit_of_row(lane & 31), lane >> 5
                for g in range(G):
                    for s in (0, 7, 11):
                        want = np.float32(scale[g] * float(w_ih[g * H + m, s + 12 * hh])) if m >= 0 else 0.0
                        assert a_ih[g, s, lane] == want
                    for i in (0, 3, 9, 14, 15):
                        u = unit_of_row((i & 3) + 8 * (i >> 2) + 4 * hh)
                        want = np.float32(scale[g] * float(w_hh[g * H + m, u])) if (m >= 0 and u >= 0) else 0.0
                        assert a_hh[g, i, lane] == want
                o = lane & 31
                for t in (0, 17, 29):
                    for i in (0, 6, 14, 15):
                        u = unit_of_row((i & 3) + 8 * (i >> 2) + 4 * hh)
                        want = hw[o, t * H + u] if (o < hw.shape[0] and u >= 0) else 0.0
                        assert head[t, i, lane] == want
            assert not a_hh[:, 15].any() and not head[:, 15].any()       # k-step 15 is all padding for H = 30
    # too-small blob is rejected with a message, not a crash
    small = np.zeros(8, np.float32)
    rc = L.vad_pack_net_weights(0, 1, H, hp(w_ih[:90]), hp(w_hh[:90]), hp(b_ih), hp(b_hh), hp(lin1_w), hp(lin1_b),
                                hp(lin2_w), hp(lin2_b), hp(lin_w), hp(lin_b), hp(small), 8)
    assert rc == -1 and b'blob too small' in L.vad_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libvad_amd.so')
    with pytest.raises(pkg.VadError):
        _lib.lib()
