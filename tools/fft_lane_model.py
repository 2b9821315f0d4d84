"""Lane-level numpy model of the 1024-point complex FFT + real untangle used by the MFCC kernel.

Mirrors the kernel's data movement (64 lanes x 16 registers; radix 16 in registers, one LDS exchange,
radix 16 in registers, radix 4 across the four lanes of a quad; paired untangle with the partner value
fetched from another lane) so the index arithmetic can be checked against numpy.fft on the CPU before it
is written in HIP.
"""
import numpy as np


def dft4(a0, a1, a2, a3):
    t0, t1, t2, t3 = a0 + a2, a0 - a2, a1 + a3, (a1 - a3) * (-1j)
    return t0 + t2, t1 + t3, t0 - t2, t1 - t3


def dft16(v):
    """v: (..., 16) complex -> natural-order 16-point DFT via 4x4 decomposition."""
    w16 = np.exp(-2j * np.pi * np.arange(16) / 16)
    y = np.zeros_like(v)
    for b in range(4):
        outs = dft4(*(v[..., 4 * a + b] for a in range(4)))
        for c in range(4):
            y[..., 4 * c + b] = outs[c] * w16[(b * c) % 16]
    out = np.zeros_like(v)
    for c in range(4):
        outs = dft4(*(y[..., 4 * c + b] for b in range(4)))
        for d in range(4):
            out[..., c + 4 * d] = outs[d]
    return out


S = 68                                  # exchange row stride (complex)
REV = np.array([0, 2, 1, 3])            # the quad stage leaves output p = REV[a] in lane a


def fft1024_lanes(z):
    """Returns v[lane][q]: lane = 4 k1 + a holds Z[k1 + 16 q + 256 REV[a]]."""
    L = np.arange(64)
    v = z[L[:, None] + 64 * np.arange(16)[None, :]].copy()          # v[lane][jj], n = lane + 64 jj
    # pass A: DFT-16 over jj -> k1; twiddle W1024^(lane k1)
    v = dft16(v) * np.exp(-2j * np.pi * (L[:, None] * np.arange(16)[None, :]) / 1024)
    # the exchange: lds[k1 * S + n2]; lane (k1 = lane >> 2, a = lane & 3) reads n2 = a + 4 b
    lds = np.zeros(16 * S, complex)
    for k1 in range(16):
        lds[k1 * S + L] = v[:, k1]
    k1, a = L >> 2, L & 3
    v = lds[(k1 * S + a)[:, None] + 4 * np.arange(16)[None, :]]     # v[lane][b]
    # pass B: DFT-16 over b -> q; twiddle W64^(a q); lanes 1..3 of a quad hold their values negated
    sgn = np.array([1, -1, -1, -1])[a]
    v = dft16(v) * np.exp(-2j * np.pi * (a[:, None] * np.arange(16)[None, :]) / 64) * sgn[:, None]
    # pass C: radix 4 across the quad, r += c * (partner's r), lane 3 turned by i in between
    c1 = np.array([-1, 1, 1, -1])[a]
    c2 = np.array([-1, 1, -1, 1])[a]
    v = v + c1[:, None] * v[L ^ 2]
    v = np.where((a == 3)[:, None], 1j * v, v)
    v = v + c2[:, None] * v[L ^ 1]
    return v


def fft_bin(lane, u):
    """The bin k whose pair (k, 1024 - k) a lane works on in untangle step u (csrc/mfcc_device.h: fft_bin)."""
    k1, a = lane >> 2, lane & 3
    if k1 == 0 and u == 0 and a in (1, 3):
        return 128 + 256 * REV[a]
    return k1 + 16 * u + 256 * REV[a]


def power_from_lanes(v):
    """Paired untangle; returns the power spectrum and how many times each bin was written."""
    P = np.full(1025, np.nan)
    cnt = np.zeros(1025, int)
    for lane in range(64):
        k1, a = lane >> 2, lane & 3
        for u in range(8):
            k = fft_bin(lane, u)
            if k1:
                A, B = v[lane, u], v[4 * (16 - k1) + 3 - a, 15 - u]
            elif u:
                A, B = v[lane, u], v[3 - a, 16 - u]
            else:
                A = v[lane, 8 if a in (1, 3) else 0]
                B = [v[0, 0], v[2, 8], v[3, 0], v[0, 8]][a]
            E = A + np.conj(B)
            O = A - np.conj(B)
            wo = np.exp(-2j * np.pi * k / 2048) * O
            P[k] = ((E.real + wo.imag) ** 2 + (E.imag - wo.real) ** 2) * 0.25
            P[1024 - k] = ((E.real - wo.imag) ** 2 + (E.imag + wo.real) ** 2) * 0.25
            cnt[k] += 1
            cnt[1024 - k] += 1
    P[512] = abs(v[1, 0]) ** 2
    cnt[512] += 1
    return P, cnt


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    x = np.zeros(2048)
    x[:1920] = rng.normal(size=1920)
    z = x[0::2] + 1j * x[1::2]
    v = fft1024_lanes(z)
    Z = np.zeros(1024, complex)
    for lane in range(64):
        for q in range(16):
            Z[(lane >> 2) + 16 * q + 256 * REV[lane & 3]] = v[lane, q]
    print('fft err', np.abs(Z - np.fft.fft(z)).max())
    P, cnt = power_from_lanes(v)
    ref = np.abs(np.fft.rfft(x)) ** 2
    print('power err', np.abs(P - ref).max() / ref.max(), 'writes per bin', np.unique(cnt))
