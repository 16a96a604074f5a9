"""Disk and wire formats on either side of the hot path, byte-compatible with the reference:

* ``IndexedDataset`` / ``IndexedDatasetBuilder`` -- the binarised training set (utils/indexed_datasets.py:7-54): ``<path>.data`` is the
  concatenation of one pickle per item, ``<path>.idx`` an ``np.save`` of ``{'offsets': [0, end_0, end_1, ...]}``.
* ``save_singer_features`` -- the ``_mel.npy`` / ``_f0.npy`` pair ``Svc.after_infer`` leaves beside a clip in singer mode
  (infer_tools/infer_tool.py:192-198).
* ``decode_voice_change_request`` / ``encode_voice_change_response`` -- the payload of the VST bridge (flask_api.py:19-38): a wav file
  in the multipart field ``sample`` plus the form fields ``fPitchChange`` / ``sampleRate`` / ``sSpeakId`` in, a PCM-16 wav at the
  DAW's rate out.  Only the payload codec lives here; the HTTP server is outside the path."""
import io
import os
import pickle
import wave
from collections import OrderedDict
from copy import deepcopy

import numpy as np


class IndexedDataset:
    """Random access to the items of ``<path>.data`` through ``<path>.idx``.  ``num_cache`` most recently read items are kept (the
    reference keeps a move-to-front list of the same length) and handed out as they are stored, like the reference does."""

    def __init__(self, path, num_cache=1):
        self.path = path
        self.data_offsets = np.load(path + ".idx", allow_pickle=True).item()["offsets"]
        self._fd = os.open(path + ".data", os.O_RDONLY)
        self.num_cache = num_cache
        self._recent = OrderedDict()

    def __len__(self):
        return len(self.data_offsets) - 1

    def check_index(self, i):
        if not 0 <= i < len(self):
            raise IndexError("index out of range")

    def __getitem__(self, i):
        self.check_index(i)
        if i in self._recent:
            self._recent.move_to_end(i)
            return self._recent[i]
        start, stop = int(self.data_offsets[i]), int(self.data_offsets[i + 1])
        item = pickle.loads(os.pread(self._fd, stop - start, start))            # positional read: safe under DataLoader forks
        if self.num_cache > 0:
            self._recent[i] = deepcopy(item)
            while len(self._recent) > self.num_cache:
                self._recent.popitem(last=False)
        return item

    def close(self):
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IndexedDatasetBuilder:
    def __init__(self, path):
        self.path = path
        self._out = open(path + ".data", "wb")
        self.byte_offsets = [0]

    def add_item(self, item):
        n = self._out.write(pickle.dumps(item))
        self.byte_offsets.append(self.byte_offsets[-1] + n)

    def finalize(self):
        self._out.close()
        with open(self.path + ".idx", "wb") as f:
            np.save(f, {"offsets": self.byte_offsets})


def save_singer_features(in_path, mel_pred, f0_pred):
    """``batch/x.wav`` -> ``singer_data/x_mel.npy`` + ``singer_data/x_f0.npy`` (infer_tool.py:192-198); returns the two paths."""
    data_path = in_path.replace("batch", "singer_data")
    mel_path, f0_path = data_path[:-4] + "_mel.npy", data_path[:-4] + "_f0.npy"
    os.makedirs(os.path.dirname(mel_path) or ".", exist_ok=True)
    np.save(mel_path, np.asarray(mel_pred))
    np.save(f0_path, np.asarray(f0_pred))
    return mel_path, f0_path


def decode_voice_change_request(form, sample_bytes):
    """(wav BytesIO for ``Svc.infer``, key in semitones, DAW sample rate, speaker id) from the form fields and the uploaded file
    (flask_api.py:21-30; numbers arrive as strings, possibly with a decimal point)."""
    key = float(form.get("fPitchChange", 0))
    daw_sr = int(float(form.get("sampleRate", 0)))
    speaker = int(float(form.get("sSpeakId", 0)))
    return io.BytesIO(sample_bytes), key, daw_sr, speaker


def encode_voice_change_response(audio, model_sr, daw_sr):
    """The reply body: ``audio`` (float, the model's rate) resampled to the DAW's rate and written as a mono PCM-16 wav
    (flask_api.py:34-38: librosa.resample + soundfile.write(format='wav'), whose default subtype is PCM_16)."""
    from .vocoder import resample
    a = np.asarray(audio, dtype=np.float32).reshape(-1)
    if daw_sr and daw_sr != model_sr:
        a = resample(a, model_sr, daw_sr)
    pcm = np.clip(np.rint(a.astype(np.float64) * 32767.0), -32768, 32767).astype("<i2")       # libsndfile scales float -> PCM_16 by 0x7FFF
    out = io.BytesIO()
    with wave.open(out, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(daw_sr or model_sr))
        w.writeframes(pcm.tobytes())
    out.seek(0)
    return out
