"""``loud_norm`` of ``process_utterance`` (preprocessing/data_gen_utils.py:117-122): the waveform is brought to -22 LUFS before the STFT --

    meter = pyln.Meter(sample_rate); loudness = meter.integrated_loudness(wav)
    wav = pyln.normalize.loudness(wav, loudness, -22.0);  if abs(wav).max() > 1: wav /= abs(wav).max()

The arithmetic lives in a third-party package, **pyloudnorm==0.1.0** (requirements.txt:67), which is absent from this image (no network).  This file
restates its published algorithm -- ITU-R BS.1770-4 integrated loudness as pyloudnorm implements it -- on the host, where the reference runs it
too (numpy + scipy.signal.lfilter on one utterance; not a device kernel: it precedes the STFT of an offline front-end):

  * K-weighting = two biquads designed for the signal's own rate: a high shelf (+4 dB, Q = 1/sqrt 2, 1500 Hz) and a high pass (Q = 0.5, 38 Hz)
    (pyloudnorm.iirfilter.IIRfilter.generate_coefficients: the audio-EQ-cookbook forms, normalised by a0);
  * 400 ms blocks with 75 % overlap, mean square per block, l_j = -0.691 + 10 log10(sum_c G_c z_cj) (G = 1 for the first three channels);
  * absolute gate -70 LUFS, relative gate 10 LU below the loudness of the blocks that pass the absolute gate;
  * normalize.loudness: gain = 10^((target - measured) / 20).

**Parity unpinned at this dependency** (like librosa's filterbank): pinned to the standard instead -- BS.1770's own conformance point, a
997 Hz full-scale sine measures -3.01 LKFS (tests/test_host.py).  The reference's shipped configs set ``loud_norm: false``
(training/config.yaml:76, config_nsf.yaml:76)."""
import numpy as np


def _biquad(kind, G, Q, fc, rate):
    A = 10.0 ** (G / 40.0)
    w0 = 2.0 * np.pi * (fc / rate)
    alpha = np.sin(w0) / (2.0 * Q)
    c = np.cos(w0)
    if kind == "high_shelf":
        b = np.array([A * ((A + 1) + (A - 1) * c + 2 * np.sqrt(A) * alpha), -2 * A * ((A - 1) + (A + 1) * c), A * ((A + 1) + (A - 1) * c - 2 * np.sqrt(A) * alpha)])
        a = np.array([(A + 1) - (A - 1) * c + 2 * np.sqrt(A) * alpha, 2 * ((A - 1) - (A + 1) * c), (A + 1) - (A - 1) * c - 2 * np.sqrt(A) * alpha])
    elif kind == "high_pass":
        b = np.array([(1 + c) / 2, -(1 + c), (1 + c) / 2])
        a = np.array([1 + alpha, -2 * c, 1 - alpha])
    else:
        raise ValueError(kind)
    return b / a[0], a / a[0]


def integrated_loudness(wav, rate, block_size=0.400):
    """pyloudnorm.Meter(rate).integrated_loudness(wav): wav [N] or [N, channels] float -> LUFS (-inf for digital silence)."""
    from scipy.signal import lfilter
    x = np.asarray(wav, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    n, ch = x.shape
    if ch > 5:
        raise ValueError("audio must have five channels or less")
    if n < block_size * rate:
        raise ValueError("audio must have length greater than the block size")
    for kind, G, Q, fc in (("high_shelf", 4.0, 1.0 / np.sqrt(2.0), 1500.0), ("high_pass", 0.0, 0.5, 38.0)):
        b, a = _biquad(kind, G, Q, fc, rate)
        x = lfilter(b, a, x, axis=0)
    gains = np.array([1.0, 1.0, 1.0, 1.41, 1.41])[:ch]
    T_g, gamma_a, step = block_size, -70.0, 0.25            # overlap 0.75
    T = n / rate
    n_blocks = int(np.round((T - T_g) / (T_g * step)) + 1)
    z = np.zeros((ch, n_blocks))
    for j in range(n_blocks):
        lo, hi = int(T_g * (j * step) * rate), int(T_g * (j * step + 1) * rate)
        z[:, j] = np.sum(np.square(x[lo:hi]), axis=0) / (T_g * rate)
    with np.errstate(divide="ignore"):
        l = -0.691 + 10.0 * np.log10(np.sum(gains[:, None] * z, axis=0))
        keep = l >= gamma_a
        z_avg = np.array([np.mean(z[c, keep]) if keep.any() else 0.0 for c in range(ch)])
        gamma_r = -0.691 + 10.0 * np.log10(np.sum(gains * z_avg)) - 10.0
        keep = (l > gamma_r) & (l > gamma_a)
        z_avg = np.array([np.mean(z[c, keep]) if keep.any() else 0.0 for c in range(ch)])
        return float(-0.691 + 10.0 * np.log10(np.sum(gains * z_avg)))


def loud_norm(wav, rate, target=-22.0):
    """The loud_norm branch of process_utterance (data_gen_utils.py:117-122): normalise to `target` LUFS, then rescale only if that clips."""
    wav = np.asarray(wav)
    gain = 10.0 ** ((target - integrated_loudness(wav, rate)) / 20.0)
    out = (gain * wav).astype(wav.dtype, copy=False) if np.issubdtype(wav.dtype, np.floating) else gain * wav
    peak = np.abs(out).max() if out.size else 0.0
    return out / peak if peak > 1 else out
