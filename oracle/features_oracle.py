"""TEST INFRASTRUCTURE ONLY (imported by tests/): NumPy restatement of the reference's log-mel front end
(utils/preprocessing.py:48-94), written independently of rnnt-speech-recognition_amd/features.py: explicit framing loop,
np.fft, scalar mel-filter construction.  Restates TensorFlow's documented definitions (tf.signal.stft: periodic Hann,
fft_length = next power of two, no end padding; tf.signal.linear_to_mel_weight_matrix: HTK mel, DC bin excluded).
Parity unpinned by the reference itself: TensorFlow is absent from this image, so no TF output could be captured."""
import math

import numpy as np


def mel_matrix(n_mel, n_bins, sr, f_lo, f_hi):
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    edges = np.linspace(mel(f_lo), mel(f_hi), n_mel + 2)
    W = np.zeros((n_bins, n_mel))
    for k in range(1, n_bins):  # bin 0 (DC) stays zero
        m = mel(k * (sr / 2.0) / (n_bins - 1))
        for j in range(n_mel):
            lo, ce, hi = edges[j], edges[j + 1], edges[j + 2]
            W[k, j] = max(0.0, min((m - lo) / (ce - lo), (hi - m) / (hi - ce)))
    return W


def log_mel(audio, sr, n_mel=80, frame_length=0.025, frame_step=0.01, f_lo=125.0, f_hi=7600.0):
    audio = np.asarray(audio, np.float64)
    L, S = int(round(sr * frame_length)), int(round(sr * frame_step))
    nfft = 1
    while nfft < L:
        nfft *= 2
    win = np.array([0.5 - 0.5 * math.cos(2.0 * math.pi * i / L) for i in range(L)])
    n_frames = 0 if len(audio) < L else 1 + (len(audio) - L) // S
    mag = np.zeros((n_frames, nfft // 2 + 1))
    for i in range(n_frames):
        mag[i] = np.abs(np.fft.rfft(audio[i * S : i * S + L] * win, n=nfft))
    lm = np.log(mag @ mel_matrix(n_mel, nfft // 2 + 1, float(sr), f_lo, f_hi) + 1e-6)
    return lm - (lm.mean(axis=0) + 1e-8)


def downsample(spec, n=3):
    t = (spec.shape[0] // n) * n
    return spec[:t].reshape(-1, spec.shape[1] * n)
