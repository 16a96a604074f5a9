"""CPU, container only: the drop-in boundary driven by the REAL reference's own seam code -- the vocoder registry
(network/vocoders/base_vocoder.py:5-19), the strict checkpoint loader (utils/__init__.py:178-209), the process-global hparams dict
-- and the slicer against the reference's shipped demo input.  Skipped where /root/reference does not exist (the GPU box)."""
import json
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DSVC_REFERENCE_ROOT", "/root/reference")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "network", "diff")), reason="reference tree not present")


@needs_ref
def test_drop_ins_through_the_real_reference_seams():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_seams_driver.py")], capture_output=True, text=True, timeout=600,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    d = json.loads(line[-1][7:])
    assert d["vocoder_cls"] == "NsfHifiGANHip" and d["vocoder_module"].endswith("vocoder")
    for k in ("registered_bare", "registered_lower", "is_base_vocoder", "by_short_name", "has_contract", "ckpt_keys_equal",
              "ckpt_values_equal", "same_keys_as_reference_model", "dir_load_ok", "strict_rejects_missing", "shares_global_hparams"):
        assert d[k] is True, (k, d)


@needs_ref
def test_reference_svc_driver_runs_wav_to_wav_over_the_drop_ins():
    """BASELINE configs[0] (plumbing, no GPU): the REAL ``Svc.infer`` / ``pre`` / ``after_infer`` of infer_tools/infer_tool.py:104-345 over
    GaussianDiffusionHip(DiffNetHip) + NsfHifiGANHip, checkpoint loaded by the reference's own strict loader, vocoder resolved through
    the reference's registry -- 20-step-class PLMS (pndm_speedup 10 over a 40-step schedule), wav in -> wav out.  The C-ABI handles are
    oracle-backed stand-ins here (no device in this container; tests/ref_infer_driver.py), so the equality below is about the host
    glue: the driver's output equals the same conversion written as plain oracle calls."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_infer_driver.py")], capture_output=True, text=True, timeout=900,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    d = json.loads(line[-1][7:])
    assert d["sampler_calls"][0] == {"t_start": 40, "speedup": 10, "seed": 5, "T": d["frames"]} and d["pndm_speedup_set_by_pre"]
    # diffsvc_amd.svc_chunks.infer_chunks: four chunks of the utterance through the reference's Svc object in fewer model calls than chunks,
    # every chunk's (f0_gt, f0_pred, wav) equal to what the loop of Svc.infer calls returns
    assert d["chunks_model_calls"] < 4 and d["chunks_equal_loop"] and d["chunks_unbatched_equal_loop"], d
    assert len(set(d["chunks_lens"])) >= 3, d
    assert d["wav_len_ok"] and d["f0_gt_is_shifted_input"] and d["wav_rms"] > 0.05
    assert d["wav_max_abs_diff"] < 1e-6 and d["f0_pred_max_abs_diff"] == 0.0


@needs_ref
def test_slicer_on_the_reference_demo_input_matches_the_real_slicer():
    """raw/test_input.wav (mono 16-bit 22 050 Hz, 22.6 s -- the only real audio the reference ships) through diffsvc_amd.slicer at six
    parameter sets against chunk dicts minted from the real infer_tools/slicer.py (tests/golden/slicer_test_input.json, written by
    oracle/make_golden.py --slicer-only).  Sample indices: bit-exact."""
    from diffsvc_amd.slicer import Slicer
    with open(os.path.join(ROOT, "tests", "golden", "slicer_test_input.json")) as f:
        kat = json.load(f)
    with wave.open(os.path.join(REF, "raw", "test_input.wav"), "rb") as w:
        sr, n = w.getframerate(), w.getnframes()
        audio = np.frombuffer(w.readframes(n), dtype="<i2").astype(np.float32) / 32768.0
    assert sr == kat["sr"] and n == kat["n_samples"]
    for case in kat["cases"]:
        got = Slicer(sr=sr, **case["args"]).slice(audio)
        assert got == case["chunks"], case["args"]


@needs_ref
def test_indexed_dataset_is_byte_compatible_with_the_reference(tmp_path):
    """utils/indexed_datasets.py: files written by our builder are read by the REAL IndexedDataset, files written by the real builder
    are read by ours, and the two builders produce identical bytes for the same items."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_indexed_datasets", os.path.join(REF, "utils", "indexed_datasets.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from diffsvc_amd.formats import IndexedDataset, IndexedDatasetBuilder
    g = np.random.default_rng(0)
    items = [{"item_name": "clip%d" % i, "mel": g.standard_normal((5 + i, 8)).astype(np.float32), "f0": g.random(5 + i).astype(np.float32),
              "hubert": g.standard_normal((3 + i, 4)).astype(np.float32), "mel2ph": np.arange(5 + i) // 2 + 1} for i in range(7)]
    ours, theirs = str(tmp_path / "ours"), str(tmp_path / "theirs")
    b = IndexedDatasetBuilder(ours)
    rb = ref.IndexedDatasetBuilder(theirs)
    for it in items:
        b.add_item(it); rb.add_item(it)
    b.finalize(); rb.finalize()
    assert open(ours + ".data", "rb").read() == open(theirs + ".data", "rb").read()
    assert open(ours + ".idx", "rb").read() == open(theirs + ".idx", "rb").read()
    for reader, path in ((ref.IndexedDataset, ours), (IndexedDataset, theirs)):
        ds = reader(path)
        assert len(ds) == len(items)
        for i in (3, 0, 6, 3, 1):
            for k, v in items[i].items():
                assert np.array_equal(ds[i][k], v) if isinstance(v, np.ndarray) else ds[i][k] == v
        with pytest.raises(IndexError):
            ds[len(items)]
        with pytest.raises(IndexError):
            ds[-1]
