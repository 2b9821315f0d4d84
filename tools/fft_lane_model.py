"""Lane-level numpy model of the 1024-point complex FFT + real untangle used by the MFCC kernel.

Mirrors the kernel's data movement (64 lanes x 16 registers, two exchanges) so the index
arithmetic can be checked against numpy.fft on the CPU before it is written in HIP.
"""
import numpy as np


def dft16(v):
    """v: (..., 16) complex -> natural-order 16-point DFT via 4x4 decomposition."""
    w16 = np.exp(-2j * np.pi * np.arange(16) / 16)
    y = np.zeros_like(v)
    for b in range(4):
        a0, a1, a2, a3 = (v[..., 4 * a + b] for a in range(4))
        t0, t1, t2, t3 = a0 + a2, a0 - a2, a1 + a3, (a1 - a3) * (-1j)
        outs = (t0 + t2, t1 + t3, t0 - t2, t1 - t3)
        for c in range(4):
            y[..., 4 * c + b] = outs[c] * w16[(b * c) % 16]        # y[c][b]
    out = np.zeros_like(v)
    for c in range(4):
        a0, a1, a2, a3 = (y[..., 4 * c + b] for b in range(4))
        t0, t1, t2, t3 = a0 + a2, a0 - a2, a1 + a3, (a1 - a3) * (-1j)
        outs = (t0 + t2, t1 + t3, t0 - t2, t1 - t3)
        for d in range(4):
            out[..., c + 4 * d] = outs[d]
    return out


def fft1024_lanes(z):
    L = np.arange(64)
    # pass 1: lane n2 holds z[n2 + 64 j]
    r = z[L[:, None] + 64 * np.arange(16)[None, :]]
    r = dft16(r)                                                   # r[n2][k1]
    r = r * np.exp(-2j * np.pi * (L[:, None] * np.arange(16)[None, :]) / 1024)
    # exchange 1: lds[k1][n2]; lane' = k1*4 + m2 reads [k1][4 m1 + m2]
    lds = r.T.copy()                                               # [k1][n2]
    k1 = L >> 2
    m2 = L & 3
    r = lds[k1[:, None], 4 * np.arange(16)[None, :] + m2[:, None]]  # r[lane'][m1]
    r = dft16(r)                                                   # r[lane'][q1]
    r = r * np.exp(-2j * np.pi * (m2[:, None] * np.arange(16)[None, :]) / 64)
    # quad DFT-4 across lanes: lane with q2 = lane&3 keeps output q2
    out = np.zeros_like(r)
    for lane in range(64):
        base = lane & ~3
        q2 = lane & 3
        acc = 0
        for mm in range(4):
            acc = acc + r[base + mm] * (-1j) ** (mm * q2)
        out[lane] = acc
    # lane' = (k1, q2), reg q1 -> k = k1 + 16 q1 + 256 q2
    Z = np.zeros(1024, complex)
    for lane in range(64):
        for q1 in range(16):
            Z[(lane >> 2) + 16 * q1 + 256 * (lane & 3)] = out[lane, q1]
    return Z


def untangle(Z):
    N = 1024
    k = np.arange(N)
    Zc = np.conj(Z[(N - k) % N])
    return 0.5 * (Z + Zc) - 0.5j * np.exp(-2j * np.pi * k / 2048) * (Z - Zc)


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    x = np.zeros(2048)
    x[:1920] = rng.normal(size=1920)
    z = x[0::2] + 1j * x[1::2]
    Z = fft1024_lanes(z)
    print('fft err', np.abs(Z - np.fft.fft(z)).max())
    X = untangle(Z)
    print('rfft err', np.abs(X - np.fft.rfft(x)[:1024]).max())
