"""Host logic of the replay path (cache reader, MOT formatting) -- no GPU."""
import numpy as np
import pytest

from boxmot_amd.replay import CachedSequence, format_for_mot, load_cached_sequence, write_mot_results


def _cache(tmp_path, name="MOT17-02", frames=(1, 2, 4), n=3, dim=8, seed=0):
    rng = np.random.default_rng(seed)
    rows, embs = [], []
    for f in frames:
        d = np.c_[np.full(n, f), rng.uniform(0, 100, (n, 2)), rng.uniform(120, 300, (n, 2)), rng.uniform(0.2, 0.9, n), np.zeros(n)]
        rows.append(d)
        embs.append(rng.standard_normal((n, dim)))
    dets = np.concatenate(rows).astype(np.float32)             # cache.py:283-300: [frame, x1, y1, x2, y2, conf, cls] fp32
    embs = np.concatenate(embs).astype(np.float32)
    (tmp_path / "dets").mkdir()
    (tmp_path / "embs").mkdir()
    np.save(tmp_path / "dets" / f"{name}.npy", dets)
    np.save(tmp_path / "embs" / f"{name}.npy", embs)
    return dets, embs


def test_cache_reader_slices_frames_like_the_reference_dataset(tmp_path):
    dets, embs = _cache(tmp_path)
    seq = load_cached_sequence("MOT17-02", tmp_path / "dets" / "MOT17-02.npy", tmp_path / "embs" / "MOT17-02.npy")
    assert list(seq.frame_ids) == [1, 2, 3, 4]
    d, e = seq.frame(2)
    assert np.array_equal(d, dets[3:6, 1:]) and np.array_equal(e, embs[3:6])
    d, e = seq.frame(3)                                          # frame without detections (dataset.py:418-422)
    assert d.shape == (0, 6) and e.shape == (0, 8)
    np.save(tmp_path / "embs" / "bad.npy", embs[:-1])
    with pytest.raises(ValueError, match="Row mismatch"):
        load_cached_sequence("x", tmp_path / "dets" / "MOT17-02.npy", tmp_path / "embs" / "bad.npy")


def test_mot_formatting_and_file(tmp_path):
    rows = np.array([[10.4, 20.6, 50.5, 81.5, 3, 0.87654321, 0, 5], [0.5, 1.5, 2.5, 4.5, 12, 0.5, 2, 0]], dtype=np.float32)
    m = format_for_mot(rows, 7)
    assert m.shape == (2, 9)
    assert m[0].tolist()[:6] == [7, 3, 10, 21, 40, 61]           # l, t, w, h rounded half-to-even like np.round (mot.py:267)
    assert m[1].tolist()[:6] == [7, 12, 0, 2, 2, 3]
    assert m[0, 7] == 1 and m[1, 7] == 3 and m[0, 8] == 5         # class + 1, det_ind
    assert format_for_mot(np.empty((0, 8)), 1).shape == (0, 9)
    p = tmp_path / "out" / "seq.txt"
    write_mot_results(p, m)
    lines = p.read_text().strip().splitlines()
    assert lines[0] == "7,3,10,21,40,61,0.876543,1,5"
    write_mot_results(tmp_path / "out" / "empty.txt", np.empty((0, 9)))
    assert (tmp_path / "out" / "empty.txt").read_text() == ""


def test_mot_formatting_equals_reference_when_available():
    from oracle import ref_harness
    if not ref_harness.reference_available():
        pytest.skip("/root/reference is not mounted")
    ref_harness.install_standins()
    try:
        from boxmot.engine.tracking.mot import convert_to_mot_format
    except Exception as exc:                                     # heavy optional imports of the engine package
        pytest.skip(f"reference engine not importable offline: {exc}")
    rng = np.random.default_rng(3)
    rows = np.c_[rng.uniform(0, 500, (20, 2)), rng.uniform(500, 900, (20, 2)), np.arange(1, 21), rng.uniform(0, 1, 20),
                 rng.integers(0, 3, 20), rng.integers(0, 30, 20)].astype(np.float32)
    assert np.array_equal(format_for_mot(rows, 11), convert_to_mot_format(rows, 11))
