"""Lane-level numpy model of the 1024-point complex FFT + real untangle used by the MFCC kernel.

Mirrors the kernel's data movement (64 lanes x 16 registers, radix 4 x 16 x 16 with two LDS
exchanges, transposed Z image, paired untangle) so the index arithmetic can be checked against
numpy.fft on the CPU before it is written in HIP.
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


S1 = 272    # exchange-1 row stride (complex)
S2 = 272    # exchange-2 plane stride (complex); row stride 17


def fft1024_lanes(z):
    """Returns Zt[q2][lane] with lane = 16 k1 + q1 holding Z[k1 + 4 q1 + 64 q2]."""
    L = np.arange(64)
    v = z[L[:, None] + 64 * np.arange(16)[None, :]].copy()          # v[lane][jj], n = lane + 64 jj
    # pass A: radix 4 over n1 = jj >> 2 for each b = jj & 3; result index 4 k1 + b; twiddle W1024^(n2 k1)
    for b in range(4):
        o = dft4(v[:, b], v[:, 4 + b], v[:, 8 + b], v[:, 12 + b])
        for k1 in range(4):
            v[:, 4 * k1 + b] = o[k1] * np.exp(-2j * np.pi * ((L + 64 * b) * k1) / 1024)
    # exchange 1: lds[k1 * S1 + n2]; lane (k1 = lane >> 4, c = lane & 15) reads n2 = c + 16 m
    lds = np.zeros(4 * S1, complex)
    for k1 in range(4):
        for b in range(4):
            lds[k1 * S1 + L + 64 * b] = v[:, 4 * k1 + b]
    k1 = L >> 4
    c = L & 15
    v = lds[(k1 * S1 + c)[:, None] + 16 * np.arange(16)[None, :]]  # v[lane][m]
    # pass B: DFT-16 over m -> q1; twiddle W256^(c q1)
    v = dft16(v) * np.exp(-2j * np.pi * (c[:, None] * np.arange(16)[None, :]) / 256)
    # exchange 2: lds[k1 * S2 + q1 * 17 + c]; lane (k1, q1 = lane & 15) reads c = 0..15
    lds = np.zeros(4 * S2 + 16, complex)
    for q1 in range(16):
        lds[k1 * S2 + q1 * 17 + c] = v[:, q1]
    q1 = L & 15
    v = lds[(k1 * S2 + q1 * 17)[:, None] + np.arange(16)[None, :]]  # v[lane][c]
    # pass C: DFT-16 over c -> q2 ; k = k1 + 4 q1 + 64 q2
    v = dft16(v)
    return v.T.copy()                                               # Zt[q2][lane]


def power_from_zt(Zt):
    """Paired untangle: lane handles k = k1 + 4 q1 + 64 jj (jj < 8) together with 1024 - k."""
    P = np.zeros(1025)
    L = np.arange(64)
    k1, q1 = L >> 4, L & 15
    lp = np.where(k1 != 0, 16 * (4 - k1) + 15 - q1, np.where(q1 != 0, 16 - q1, 0))
    d = (L == 0).astype(int)
    zt = Zt.reshape(-1)
    for jj in range(8):
        k = k1 + 4 * q1 + 64 * jj
        a = zt[jj * 64 + L]
        b = zt[((15 - jj + d) & 15) * 64 + lp]
        E = a + np.conj(b)
        O = a - np.conj(b)
        wo = np.exp(-2j * np.pi * k / 2048) * O
        P[k] = ((E.real + wo.imag) ** 2 + (E.imag - wo.real) ** 2) * 0.25
        P[1024 - k] = ((E.real - wo.imag) ** 2 + (E.imag + wo.real) ** 2) * 0.25
    P[512] = abs(zt[8 * 64 + 0]) ** 2
    return P


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    x = np.zeros(2048)
    x[:1920] = rng.normal(size=1920)
    z = x[0::2] + 1j * x[1::2]
    Zt = fft1024_lanes(z)
    Z = np.zeros(1024, complex)
    for q2 in range(16):
        for lane in range(64):
            Z[(lane >> 4) + 4 * (lane & 15) + 64 * q2] = Zt[q2, lane]
    print('fft err', np.abs(Z - np.fft.fft(z)).max())
    P = power_from_zt(Zt)
    ref = np.abs(np.fft.rfft(x)) ** 2
    print('power err', np.abs(P[1:1024] - ref[1:1024]).max() / ref.max(), 'P[512] err', abs(P[512] - ref[512]))
