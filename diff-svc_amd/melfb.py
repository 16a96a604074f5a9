"""Mel filterbank for the STFT front-end: the Slaney-scale, Slaney-normalised triangular filters that
``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` (librosa 0.9.1 defaults, htk=False, norm='slaney')
returns -- the call the reference makes at modules/nsf_hifigan/nvSTFT.py:88.  librosa is an un-vendored
third-party dependency (requirements.txt) that is absent from this image, so the published algorithm
(Slaney's Auditory Toolbox mel scale: linear below 1 kHz, log above with step ln(6.4)/27) is restated
here, vectorised.  Host-side, numpy only; the result is handed to dsvc_melspec_create."""
import numpy as np

_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / _F_SP
    with np.errstate(divide="ignore", invalid="ignore"):
        log = _MIN_LOG_MEL + np.log(np.maximum(f, 1e-300) / _MIN_LOG_HZ) / _LOGSTEP
    return np.where(f >= _MIN_LOG_HZ, log, lin)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _MIN_LOG_MEL, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), _F_SP * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """float32 [n_mels, n_fft//2+1]."""
    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    rising = -ramps[:-2] / width[:-1, None]
    falling = ramps[2:] / width[1:, None]
    fb = np.maximum(0.0, np.minimum(rising, falling)).astype(np.float32)
    enorm = 2.0 / (edges[2:] - edges[:-2])
    return (fb.astype(np.float64) * enorm[:, None]).astype(np.float32)     # librosa: float32 *= float64
