"""Float64 CPU restatement of the MFCC/delta front end (TEST INFRASTRUCTURE).

PARITY UNPINNED.  The arithmetic of this stage lives in the third-party package
``python_speech_features`` (requirements.txt:9 of the reference, no version
pin; the only upstream release line is 0.6).  That package is neither vendored
under the reference tree nor installed in the build image, and the reference
holds no golden vectors for it, so this file restates the package's published
algorithm for exactly the call the reference makes

    python_speech_features.mfcc(frames_aligned, 16000, winstep=0.03,
                                winlen=0.12, nfft=2048)        # core/process_data.py:272-273
    mfcc = mfcc[:, 1:]                                          # core/process_data.py:276
    python_speech_features.delta(mfcc, 2)                       # core/process_data.py:277

and is anchored by known-answer tests (tests/test_oracle_mfcc.py), not by
reference output.  It uses the same primitives the package uses
(numpy.fft.rfft, scipy.fftpack.dct(type=2, norm='ortho'), numpy.dot).

Every stage is a separate function so that the tests can probe it:

    preemphasis_2d   psf.sigproc.preemphasis applied to the 2-D int16 array the
                     reference passes in  (row-wise => lag of one 480-sample row)
    frame_signal     psf.sigproc.framesig, rectangular window 1920 / hop 480
    power_spectrum   psf.sigproc.powspec, nfft 2048
    mel_filterbank   psf.get_filterbanks(26, 2048, 16000, 0, 8000)
    mfcc_slice       the full call incl. log, DCT-II ortho, lifter 22, drop c0
    delta            psf.delta(feat, 2)
"""
import math

import numpy as np
from scipy.fftpack import dct

SAMPLE_RATE = 16000          # core/common.py:17
FRAME_SIZE = 480             # core/process_data.py:56  (30 ms)
WINDOW_FRAMES = 4            # core/process_data.py:49  MFCC_WINDOW_FRAME_SIZE
ALIGN_FRAMES = WINDOW_FRAMES - 1
NFFT = 2048                  # core/process_data.py:273
NFILT = 26                   # psf.mfcc default
NUMCEP = 13                  # psf.mfcc default (c0 dropped by the caller)
PREEMPH = 0.97               # psf.mfcc default
CEPLIFTER = 22               # psf.mfcc default
EPS = np.finfo(float).eps    # psf.fbank zero-energy clamp


def hz2mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)


def mel2hz(mel):
    return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)


def mel_bin_edges(nfilt=NFILT, nfft=NFFT, samplerate=SAMPLE_RATE, lowfreq=0.0, highfreq=None):
    """The 28 FFT-bin edges of the 26 triangular filters (floor((nfft+1)*hz/fs))."""
    highfreq = highfreq or samplerate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    return np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)


def mel_filterbank(nfilt=NFILT, nfft=NFFT, samplerate=SAMPLE_RATE, lowfreq=0.0, highfreq=None):
    """(nfilt, nfft//2+1) float64 triangular filters, psf.get_filterbanks semantics."""
    edges = mel_bin_edges(nfilt, nfft, samplerate, lowfreq, highfreq)
    fb = np.zeros((nfilt, nfft // 2 + 1))
    for j in range(nfilt):
        for i in range(int(edges[j]), int(edges[j + 1])):
            fb[j, i] = (i - edges[j]) / (edges[j + 1] - edges[j])
        for i in range(int(edges[j + 1]), int(edges[j + 2])):
            fb[j, i] = (edges[j + 2] - i) / (edges[j + 2] - edges[j + 1])
    return fb


def preemphasis_2d(frames_aligned, coeff=PREEMPH):
    """psf.sigproc.preemphasis on the 2-D array the reference hands to psf.mfcc.

    ``numpy.append(signal[0], signal[1:] - coeff * signal[:-1])`` with a 2-D
    ``signal`` subtracts the previous ROW (480 samples earlier), leaves row 0
    untouched and flattens.  Returns float64, shape (rows*480,).
    """
    sig = np.asarray(frames_aligned)
    return np.append(sig[0], sig[1:] - coeff * sig[:-1])


def _round_half_up(x):
    return int(math.floor(x + 0.5))


def frame_signal(sig, frame_len, frame_step):
    """psf.sigproc.framesig with the default rectangular window."""
    slen = len(sig)
    frame_len = _round_half_up(frame_len)
    frame_step = _round_half_up(frame_step)
    if slen <= frame_len:
        numframes = 1
    else:
        numframes = 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    padded = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = np.arange(frame_len)[None, :] + frame_step * np.arange(numframes)[:, None]
    return padded[idx]


def power_spectrum(frames, nfft=NFFT):
    """psf.sigproc.powspec: |rfft(frame, nfft)|^2 / nfft."""
    return (1.0 / nfft) * np.square(np.absolute(np.fft.rfft(frames, nfft)))


def lifter_weights(ncoeff=NUMCEP, L=CEPLIFTER):
    n = np.arange(ncoeff)
    return 1.0 + (L / 2.0) * np.sin(np.pi * n / L)


_FB = None


def log_mel_energies(frames_aligned):
    """(rows-3, 26) float64 log filterbank energies of one aligned slice."""
    global _FB
    if _FB is None:
        _FB = mel_filterbank()
    winstep = 30 / 1000                      # FRAME_SIZE_MS / 1000
    winlen = WINDOW_FRAMES * (30 / 1000)
    sig = preemphasis_2d(frames_aligned)
    frames = frame_signal(sig, winlen * SAMPLE_RATE, winstep * SAMPLE_RATE)
    pspec = power_spectrum(frames, NFFT)
    feat = np.dot(pspec, _FB.T)
    feat = np.where(feat == 0, EPS, feat)
    return np.log(feat)


def mfcc_slice(frames_aligned):
    """psf.mfcc(...)[:, 1:] for one aligned slice: (N+3, 480) int16 -> (N, 12) float64."""
    feat = log_mel_energies(frames_aligned)
    feat = dct(feat, type=2, axis=1, norm='ortho')[:, :NUMCEP]
    feat = lifter_weights() * feat
    # psf overwrites column 0 with log(frame energy); the caller drops it
    # (core/process_data.py:276), so it is not computed here.
    return feat[:, 1:]


def delta(feat, N=2):
    """psf.delta: regression over +-N rows with edge replication."""
    feat = np.asarray(feat, dtype=np.float64)
    numframes = len(feat)
    denominator = 2 * sum(i ** 2 for i in range(1, N + 1))
    out = np.empty_like(feat)
    padded = np.pad(feat, ((N, N), (0, 0)), mode='edge')
    w = np.arange(-N, N + 1)
    for t in range(numframes):
        out[t] = np.dot(w, padded[t: t + 2 * N + 1]) / denominator
    return out


def process_test_frames(data_frames, align_frames):
    """Restatement of core/process_data.py:257-279.

    The pydub round trip at :259-267 is the identity on int16 samples.
    Returns (frames int16 (N,480), mfcc float64 (N,12), delta float64 (N,12)).
    """
    data_frames = np.asarray(data_frames, dtype=np.int16)
    frames = data_frames.reshape(-1, FRAME_SIZE)
    aligned = np.concatenate((np.asarray(align_frames, dtype=np.int16).reshape(-1, FRAME_SIZE), frames))
    m = mfcc_slice(aligned)
    return frames, m, delta(m, 2)
