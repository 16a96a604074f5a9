"""Silence slicer -- the host-side step in front of the hot path (``infer_tools/slicer.py:40-125`` ``Slicer.slice``,
driven by ``cut`` :128-146 and consumed as ``{"<i>": {"slice": bool, "split_time": "begin,end"}}`` by
``infer.py:27-32`` / ``infer_tools/slicer.py:149-158``).  It stays on the CPU as it does in the reference
(0.135 s for 22.6 s of audio) and its sample indices are integer work: the bar is bit-exact, pinned by
``tests/golden/slicer_kat.json`` (minted from the real ``Slicer`` by oracle/make_golden.py).

Formulation used here: the windowed peak level is thresholded once, the maximal runs of quiet windows are found with one
``np.flatnonzero(np.diff(...))``, and each run is then turned into a cut pair -- instead of the reference's
sample-by-sample two-pointer scan.  The arithmetic that decides an index (scipy's running-sum uniform filter on float32
input, sqrt / clip / log10, first-minimum argmin) is kept call for call so that ties and rounding fall the same way.
"""
import numpy as np
from scipy.ndimage import maximum_filter1d, uniform_filter1d


def _valid(filtered, n, win):
    """the part of a centred length-``win`` filter output whose window lies fully inside the signal"""
    return filtered[win // 2: win // 2 + n - win + 1]


def _db(levels):
    return 20 * np.log10(np.clip(levels, a_min=1e-12, a_max=1))


class Slicer:
    def __init__(self, sr, db_threshold=-40, min_length=5000, win_l=300, win_s=20, max_silence_kept=500):
        self.db_threshold = db_threshold
        self.min_samples = round(sr * min_length / 1000)
        self.win_ln = round(sr * win_l / 1000)
        self.win_sn = round(sr * win_s / 1000)
        self.max_silence = round(sr * max_silence_kept / 1000)
        if not self.min_samples >= self.win_ln >= self.win_sn:
            raise ValueError("The following condition must be satisfied: min_length >= win_l >= win_s")
        if not self.max_silence >= self.win_sn:
            raise ValueError("The following condition must be satisfied: max_silence_kept >= win_s")

    # quietest point of ``seg`` (absolute offset ``base``): the short window with the lowest RMS level, then the sample with
    # the smallest |amplitude| inside it
    def _quiet_point(self, samples, abs_amp, base, seg):
        ws = self.win_sn
        var = uniform_filter1d(np.power(seg, 2), ws) - np.power(uniform_filter1d(seg, ws), 2)
        w0 = base + int(np.argmin(_db(_valid(np.sqrt(var), seg.shape[0], ws))))
        return w0 + int(np.argmin(abs_amp[w0: w0 + ws]))

    def silence_tags(self, samples):
        """[(begin, end)] sample ranges tagged silent, in order."""
        n_s = samples.shape[0]
        wl = self.win_ln
        abs_amp = np.abs(samples - np.mean(samples))
        peak_db = _db(_valid(maximum_filter1d(abs_amp, size=wl), n_s, wl))
        n = peak_db.shape[0]
        quiet = np.concatenate(([False], peak_db < self.db_threshold, [False]))
        edges = np.flatnonzero(quiet[1:] != quiet[:-1])
        tags = []
        with np.errstate(invalid="ignore", divide="ignore"):
            for lo, hi in zip(edges[0::2].tolist(), edges[1::2].tolist()):     # windows lo..hi-1 are quiet
                span = min(self.max_silence, (hi + wl - lo) // 2)
                if hi == n:                                                     # quiet up to the end of the signal
                    tags.append((self._quiet_point(samples, abs_amp, lo, samples[lo: lo + span]), n_s))
                    break
                begin = 0 if lo == 0 else self._quiet_point(samples, abs_amp, lo, samples[lo: lo + span])
                if tags and begin - tags[-1][1] < self.min_samples and hi < n - 1:
                    continue                                                    # voiced piece in front would be too short
                if hi == n - 1:
                    end = hi + wl
                else:
                    base = hi + wl - span
                    end = self._quiet_point(samples, abs_amp, base, samples[base: hi + wl])
                tags.append((begin, end))
        return tags

    def slice(self, audio):
        n_s = len(audio)
        whole = {"0": {"slice": False, "split_time": "0,%d" % n_s}}
        if audio.shape[0] <= self.min_samples:
            return whole
        tags = self.silence_tags(audio)
        if not tags:
            return whole
        cuts = []
        pos = 0
        for k, (b, e) in enumerate(tags):
            if k or b:
                cuts.append((False, pos, b))
            cuts.append((True, b, e))
            pos = e
        if pos != n_s:
            cuts.append((False, pos, n_s))
        return {str(i): {"slice": s, "split_time": "%d,%d" % (b, e)} for i, (s, b, e) in enumerate(cuts)}


def cut_samples(audio, sr, db_thresh=-30, min_len=5000, win_l=300, win_s=20, max_sil_kept=500):
    """``cut`` (infer_tools/slicer.py:128-146) for audio already in memory: [N] or [channels, N] float -> chunk dict."""
    audio = np.asarray(audio)
    if audio.ndim == 2:
        audio = audio.mean(axis=0) if audio.shape[0] >= 2 else audio[0]
    return Slicer(sr=sr, db_threshold=db_thresh, min_length=min_len, win_l=win_l, win_s=win_s,
                  max_silence_kept=max_sil_kept).slice(audio)


def chunks_of(chunks, audio):
    """``chunks2audio`` (infer_tools/slicer.py:149-158) without the file read: [(is_silent, samples)] per chunk."""
    out = []
    for v in chunks.values():
        b, e = (int(x) for x in v["split_time"].split(","))
        if b != e:
            out.append((v["slice"], audio[b:e]))
    return out
