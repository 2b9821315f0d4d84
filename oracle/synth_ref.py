"""numpy twin of the synthetic 16 kHz int16 audio generator (TEST INFRASTRUCTURE).

The generator is stateless and integer-only: sample ``n`` of stream ``seed`` is
a pure function of (seed, n), so any sub-range can be regenerated on the host
and compared bit-for-bit with what the device kernel wrote.  The reference
ships no audio (its datasets are LibriSpeech / QUT-NOISE / private recordings),
so this replaces core/prepare_files.py as the source of frames.

Signal model, per 250 ms segment (4000 samples) chosen by a hash of the
segment index: silence (exact zeros, which drive the front end's zero-energy
clamp), or approximately gaussian noise at one of three levels plus an optional
triangle-wave tone, saturated to int16.
"""
import numpy as np

SEGMENT = 4000
_NOISE_AMP = np.array([0, 0, 48, 48, 640, 640, 2560, 9000], dtype=np.int64)
_TONE_AMP = np.array([0, 0, 0, 1500, 0, 6000, 0, 12000], dtype=np.int64)


def _hash32(x):
    """lowbias32 integer finaliser on uint32 arrays."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def synth_audio(start, count, seed=1337):
    """int16 samples [start, start+count) of stream ``seed``."""
    with np.errstate(over='ignore'):
        n = np.arange(start, start + count, dtype=np.uint64)
        lo = (n & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (n >> np.uint64(32)).astype(np.uint32)
        s = np.uint32(seed)
        k1 = _hash32(lo ^ _hash32(hi + s * np.uint32(0x9E3779B9)))
        k2 = _hash32(k1 ^ np.uint32(0x68E31DA4))
        noise = ((k1 & np.uint32(0xFFFF)).astype(np.int64) + (k1 >> np.uint32(16)).astype(np.int64)
                 + (k2 & np.uint32(0xFFFF)).astype(np.int64) + (k2 >> np.uint32(16)).astype(np.int64)
                 - 131070)
        seg = n // np.uint64(SEGMENT)
        seg_lo = (seg & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        seg_hi = (seg >> np.uint64(32)).astype(np.uint32)
        hs = _hash32(seg_lo ^ _hash32(seg_hi ^ (s + np.uint32(0x51ED270B))))
        level = (hs & np.uint32(7)).astype(np.int64)
        fstep = (((hs >> np.uint32(8)) & np.uint32(0x3FF)) + np.uint32(16)).astype(np.uint32) * np.uint32(1 << 20)   # 62 Hz .. 4.06 kHz
        phase = lo * fstep                                  # uint32 wrap-around
        p = (phase >> np.uint32(16)).astype(np.int64)
        tri = 2 * np.where(p < 32768, p, 65535 - p) - 32767
        v = ((noise * _NOISE_AMP[level]) >> 16) + ((tri * _TONE_AMP[level]) >> 16)
        return np.clip(v, -32768, 32767).astype(np.int16)


def synth_frames(first_frame, n_frames, seed=1337, frame_size=480):
    """(n_frames, 480) int16 rows of the stream; row r holds samples [480 r, 480 r + 480)."""
    return synth_audio(first_frame * frame_size, n_frames * frame_size, seed).reshape(n_frames, frame_size)
