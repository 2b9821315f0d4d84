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


def test_argument_checks_reject_before_any_device_work(L):
    """Every compute entry point validates its arguments first and answers VAD_EINVAL (-1) with a message;
    these calls never reach the HIP runtime, so they run on a GPU-less host."""
    import ctypes
    buf = np.zeros(64, np.float32)
    ok = buf.ctypes.data_as(ctypes.c_void_p)                 # 16-byte aligned by numpy
    odd = ctypes.c_void_p(buf.ctypes.data + 2)
    f0 = ctypes.c_float(0.0)
    cases = [
        ('vad_mfcc_delta_f32', (None, 1, 1, 1, 480, ok, 12, ok, 12, None), b'null pointer'),
        ('vad_mfcc_delta_f32', (ok, 0, 1, 1, 480, ok, 12, ok, 12, None), b'n_slices'),
        ('vad_mfcc_delta_f32', (ok, 1, 4, 5, 480, ok, 12, ok, 12, None), b'last_slice_frames'),
        ('vad_mfcc_delta_f32', (ok, 1, 1, 1, 481, ok, 12, ok, 12, None), b'slice_stride'),
        ('vad_mfcc_delta_f32', (ok, 1, 1, 1, 480, ok, 11, ok, 12, None), b'leading dimensions'),
        ('vad_mfcc_delta_f32', (odd, 1, 1, 1, 480, ok, 12, ok, 12, None), b'4-byte aligned'),
        ('vad_gru_forward_f32', (None, 1, 720, ok, 30, 1, ok, None, 0, f0, None), b'null pointer'),
        ('vad_gru_forward_f32', (ok, 1, 720, ok, 97, 1, ok, None, 0, f0, None), b'hidden'),
        ('vad_lstm_forward_f32', (ok, 1, 720, ok, 33, 1, ok, None, 0, f0, None), b'hidden'),
        ('vad_gru_forward_f32', (ok, -1, 720, ok, 30, 1, ok, None, 0, f0, None), b'n_windows'),
        ('vad_gru_forward_f32', (ok, 1, 722, ok, 30, 1, ok, None, 0, f0, None), b'window_stride'),
        ('vad_gru_forward_f32', (odd, 1, 720, ok, 30, 1, ok, None, 0, f0, None), b'16-byte aligned'),
        ('vad_gru_forward_f32', (ok, 1, 720, ok, 30, 1, ok, None, 7, f0, None), b'decision_mode'),
        ('vad_gru_forward_mode_f32', (ok, 1, 24, ok, 30, 1, ok, None, 0, f0, 6, None, 0, None), b'matrix mode'),
        ('vad_gru_forward_mode_f32', (ok, 1, 24, ok, 30, 1, ok, None, 0, f0, -2, None, 0, None), b'matrix mode'),
        ('vad_gru_forward_mode_f32', (ok, 1, 24, ok, 90, 1, ok, None, 0, f0, 0, None, 64, None), b'null workspace'),
        ('vad_lstm_forward_mode_f32', (ok, 1, 24, ok, 30, 1, ok, None, 0, f0, 17, None), b'matrix mode'),
        ('vad_lstm_forward_mode_f32', (ok, 1, 24, ok, 33, 1, ok, None, 0, f0, -1, None), b'hidden'),
        ('vad_densenet_forward_f32', (ok, 1, 720, None, 0, ok, None, 0, f0, None), b'null pointer'),
        ('vad_densenet_forward_f32', (ok, 1, 721, ok, 0, ok, None, 0, f0, None), b'window_stride'),
        ('vad_gru_forward_rows_f32', (ok, 29, ok, 1, ok, 30, 1, ok, None, 0, f0, None), b'at least 30 rows'),
        ('vad_lstm_forward_rows_f32', (ok, 100, None, 1, ok, 30, 1, ok, None, 0, f0, None), b'null pointer'),
        ('vad_densenet_forward_rows_f32', (ok, 10, ok, 1, ok, 0, ok, None, 0, f0, None), b'at least 30 rows'),
        ('vad_overlay_i16', (ok, -1, ok, 1, 0, 1.0, ok, None), b'negative'),
        ('vad_overlay_i16', (None, 8, ok, 8, 0, 1.0, ok, None), b'null pointer'),
        ('vad_overlay_i16', (ok, 8, ok, 8, 1, float('nan'), ok, None), b'gain factor'),
        ('vad_threshold_sweep_f32', (ok, 1, ok, 8, ok, 0, ok, ok, ok, None), b'n_thresholds'),
        ('vad_threshold_sweep_f32', (ok, 1, ok, 8, ok, 5000, ok, ok, ok, None), b'n_thresholds'),
        ('vad_threshold_sweep_f32', (ok, 0, ok, 8, ok, 4, ok, ok, ok, None), b'score_stride'),
        ('vad_threshold_sweep_f32', (None, 1, ok, 8, ok, 4, ok, ok, ok, None), b'null pointer'),
        ('vad_synth_audio_i16', (None, 0, 8, 1, None), b'null pointer'),
        ('vad_synth_audio_i16', (ok, -1, 8, 1, None), b'negative'),
    ]
    for name, args, needle in cases:
        rc = getattr(L, name)(*args)
        assert rc == -1, (name, args, rc)
        assert needle in L.vad_last_error(), (name, L.vad_last_error())
    assert L.vad_net_blob_floats(1, 1, 33) == 0 and L.vad_net_blob_floats(0, 1, 97) == 0
    assert L.vad_net_blob_floats(0, 1, 30) > 0 and L.vad_densenet_blob_floats(1) > L.vad_densenet_blob_floats(0)


def _unit_of_tile_row(r, nu):
    """Placement of nu < = 32 units on the 32 rows of a tile (rnn_device.h: unit_of_tile_row)."""
    ks = (nu + 1) // 2
    kstep = lambda q: (q & 3) + 4 * (q >> 3)
    if kstep(r) >= ks:
        return -1
    rank = sum(1 for q in range(r) if kstep(q) < ks)
    return rank if rank < nu else -1


def _bf16x3(entry_hi, entry_mid, entry_lo):
    """(64, 8) uint16 images of the three terms -> float64 sum of the three bfloat16 values."""
    f = lambda e: (e.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return f(entry_hi) + f(entry_mid) + f(entry_lo)


def _f16x3(entry_hi, entry_mid, entry_lo):
    """The same for three float16 terms."""
    f = lambda e: np.ascontiguousarray(e).view(np.float16).astype(np.float64)
    return f(entry_hi) + f(entry_mid) + f(entry_lo)


@pytest.mark.parametrize('hidden,lstm', [(30, 0), (30, 1), (17, 0), (90, 0), (40, 0)])
def test_split_operand_layout(L, hidden, lstm):
    """Host-only: the bf16 (hi, mid, lo) A operands of v_mfma_f32_16x16x32_bf16 that the packer writes sum back
    exactly to the scaled float32 weights, at the rows and k-slots the split kernels read them from: lane 16 q + m
    of entry (tile mt, row half mi, k-block kb) holds tile row 16 mi + m and k-slots 8 q + j = tile rows
    16 (j >> 2) + 4 q + (j & 3) of tile kb.  The float16 tables that follow have the same layout and hold the weights
    times a power of two 2^Sw (the largest that keeps them under 65504); their three terms sum back exactly for scaled
    magnitudes >= 0.5 and to within 2^-25 below (float16's subnormal grid)."""
    H = hidden
    UT = (H + 31) // 32
    G = 4 if lstm else 3
    rng = np.random.default_rng(H)
    f32 = lambda *s: rng.standard_normal(s).astype(np.float32)
    LOG2E = 1.4426950408889634
    scale = [-LOG2E, -LOG2E, 2 * LOG2E, -LOG2E] if lstm else [-LOG2E, -LOG2E, 2 * LOG2E]
    w_ih, w_hh, b_ih, b_hh = f32(G * H, 24), f32(G * H, H), f32(G * H), f32(G * H)
    lin1_w, lin1_b, lin2_w, lin2_b, lin_w, lin_b = f32(26, 30 * H), f32(26), f32(2, 26), f32(2), f32(2, 30 * H), f32(2)
    n = L.vad_net_blob_floats(lstm, 1, H)
    blob = np.zeros(n, np.float32)
    hp = _lib.host_ptr
    assert L.vad_pack_net_weights(lstm, 1, H, hp(w_ih), hp(w_hh), hp(b_ih), hp(b_hh), hp(lin1_w), hp(lin1_b),
                                  hp(lin2_w), hp(lin2_b), hp(lin_w), hp(lin_b), hp(blob), n) == 0
    # offsets as in net_layout (rnn_device.h)
    a_hh = G * UT * 12 * 64
    cb = a_hh + G * UT * UT * 16 * 64
    head = cb + 4 * UT * 2 * 16
    tail = head + 30 * UT * 16 * 64
    s_hh = tail + (2 * 16 + 2 * 2 * 16 + 4) + 4              # head tail, then 2^S, 2^-S, 2^Sw, pad
    up, inv, wscale = (float(v) for v in blob[s_hh - 4: s_hh - 1])
    assert up * inv == 1.0 and up == wscale * 2.0 ** 14 and np.log2(wscale) == int(np.log2(wscale))
    biggest = max(np.abs(np.asarray(scale)[:, None, None] * w_hh.reshape(G, H, H)).max(), np.abs(lin1_w).max())
    assert 32500.0 < biggest * wscale <= 65000.0 or wscale == 2.0 ** 13
    gate_entries = G * 2 * 3 if UT == 1 else 3 * UT * 2 * UT * 3
    s_head = s_hh + gate_entries * 256
    words = blob.view(np.uint16)
    entry = lambda base_floats, index: words[2 * (base_floats + index * 256): 2 * (base_floats + (index + 1) * 256)].reshape(64, 8)
    unit = lambda tile, r: (lambda u: -1 if u < 0 else 32 * tile + u)(_unit_of_tile_row(r, min(32, max(0, H - 32 * tile))))

    hh_floats = gate_entries * 256
    head_floats = 30 * 2 * 3 * 256 if UT == 1 else 30 * 2 * UT * 3 * 256
    assert n == s_head + head_floats + hh_floats + head_floats            # bf16 tables, then the float16 ones
    h_hh = s_head + head_floats
    h_head = h_hh + hh_floats
    def check(e_hi, e_mid, e_lo, value_at, f16=False):
        got = (_f16x3 if f16 else _bf16x3)(e_hi, e_mid, e_lo)
        for lane in range(64):
            for j in range(8):
                want = float(value_at(lane & 15, 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3)))
                if f16:
                    want = float(np.float32(want) * np.float32(wscale))
                if f16 and abs(want) < 0.5:
                    assert abs(got[lane, j] - want) <= 2.0 ** -25, (lane, j)
                else:
                    assert got[lane, j] == want, (lane, j)

    for g in range(G):
        for mt in range(UT):
            for mi in range(2):
                for kb in range(UT):
                    def gate_value(m16, p, g=g, mt=mt, mi=mi, kb=kb):
                        m, u = unit(mt, 16 * mi + m16), unit(kb, p)
                        return np.float32(0.0) if m < 0 or u < 0 else np.float32(scale[g] * float(w_hh[g * H + m, u]))
                    for off, f16 in ((s_hh, False), (h_hh, True)):
                        if UT == 1:
                            base = (g * 2 + mi) * 3
                            check(entry(off, base), entry(off, base + 1), entry(off, base + 2), gate_value, f16)
                        else:
                            base = ((g * UT + mt) * 2 + mi) * UT + kb
                            hm = 3 * UT * 2 * UT * 2
                            check(entry(off, 2 * base), entry(off, 2 * base + 1), entry(off, hm + base), gate_value, f16)
    for t in (0, 13, 29):
        for mi in range(2):
            for kb in range(UT):
                def head_value(m16, p, t=t, mi=mi, kb=kb):
                    o, u = 16 * mi + m16, unit(kb, p)
                    return np.float32(0.0) if o >= 26 or u < 0 else lin1_w[o, t * H + u]
                base = (t * 2 + mi) * 3 if UT == 1 else ((t * 2 + mi) * UT + kb) * 3
                check(entry(s_head, base), entry(s_head, base + 1), entry(s_head, base + 2), head_value)
                check(entry(h_head, base), entry(h_head, base + 1), entry(h_head, base + 2), head_value, True)
