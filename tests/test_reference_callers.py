"""The reference's OWN per-frame callers driving a boxmot_amd tracker unchanged (build container only: needs /root/reference).

``TrackerRuntime`` (boxmot/engine/tracking/runtime.py:15-128) and ``Results._run_tracker`` (boxmot/engine/tracking/results.py:467-496)
are imported from the reference tree and handed ``boxmot_amd.BotSort`` -- the real host class; its C-ABI calls are answered by the
emulated device step (tests/emu_lib.py: botsort_step.hpp on CPU threads) because this container has no GPU.  The same callers
drive the reference's own ``BotSort`` beside it: rows must agree frame by frame (ids / conf / cls / det_ind exact, boxes to 1e-3).
The package carries no re-typed copy of these callers; a drop-in is shown by the originals accepting it."""
import logging
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")


@pytest.fixture()
def emulated_abi(monkeypatch):
    from boxmot_amd import _lib
    from emu_lib import EmuHipLib
    lib = EmuHipLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(_lib, "last_error", lambda: lib.boxmot_hip_last_error().decode())
    return lib


def _pair(**kw):
    from boxmot_amd.botsort import BotSort as HipBotSort
    RefBotSort = ref_harness.load_botsort()
    ours = HipBotSort(use_cmc=False, max_tracks=128, max_dets=64, emb_dim=32, **kw)
    ref = RefBotSort(reid_model=None, use_cmc=False, **({"with_reid": True} | kw))
    return ours, ref


def _same(a, b, t):
    a, b = np.asarray(a, dtype=np.float32).reshape(-1, 8), np.asarray(b, dtype=np.float32).reshape(-1, 8)
    assert a.shape == b.shape, (t, a.shape, b.shape)
    assert np.array_equal(a[:, 4:], b[:, 4:]), t
    assert np.allclose(a[:, :4], b[:, :4], rtol=0, atol=1e-3), t


def test_reference_tracker_runtime_drives_boxmot_amd_botsort(emulated_abi):
    from boxmot_amd.scenario import stress_frames
    logging.disable(logging.CRITICAL)
    TrackerRuntime, _ = ref_harness.load_engine_callers()
    ours, ref = _pair()
    rt_ours, rt_ref = TrackerRuntime(ours), TrackerRuntime(ref)
    assert rt_ours._accepts_embs and rt_ours._accepts_masks            # the signature the reference inspects (runtime.py:25-35)
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for t, (d, e) in enumerate(stress_frames(40, seed=4)):
        got, ms = rt_ours.update(d.copy(), img, embs=e.copy())
        want, _ = rt_ref.update(d.copy(), img, embs=e.copy())
        assert got.dtype == np.float32 and got.ndim == 2 and ms >= 0
        _same(got, want, t)
        if len(want):
            # the reference's MOT formatter accepts our rows as it accepts its own (mot.py:239-344)
            assert np.array_equal(TrackerRuntime.format_for_mot(got, t + 1)[:, :2], TrackerRuntime.format_for_mot(want, t + 1)[:, :2])
    none, no_embs = np.empty((0, 6), dtype=np.float32), np.empty((0, 32), dtype=np.float32)     # a frame without detections
    got, _ = rt_ours.update(none, img, embs=no_embs)
    want, _ = rt_ref.update(none, img, embs=no_embs)
    assert got.shape == want.shape
    ours.close()


def test_reference_results_run_tracker_drives_boxmot_amd_botsort(emulated_abi):
    from boxmot_amd.scenario import stress_frames
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    logging.disable(logging.CRITICAL)
    _, Results = ref_harness.load_engine_callers()
    from boxmot.trackers.track_results import TrackResults as RefTrackResults
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    ours, ref = _pair(**kw)
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for t, (d, e) in enumerate(stress_frames(40, seed=7)):
        # Results._run_tracker only touches self.tracker: call the reference method on a holder of our tracker
        got = Results._run_tracker(SimpleNamespace(tracker=ours), d.copy(), img, e.copy())
        want = Results._run_tracker(SimpleNamespace(tracker=ref), d.copy(), img, e.copy())
        assert isinstance(got, RefTrackResults) and isinstance(want, RefTrackResults)
        _same(got, want, t)
        assert Results._extract_track_ids(got) == Results._extract_track_ids(want)
    ours.close()


@pytest.mark.parametrize("name", ["bytetrack", "ocsort", "botsort"])
def test_reference_tracker_runtime_drives_the_oriented_trackers(emulated_abi, name):
    """The same caller with ORIENTED detections (7 columns): our ByteTrack / OcSort / BotSort and the reference's own class under the
    reference's TrackerRuntime, 9-column rows through its MMOT formatter (runtime.py:81-88 -> convert_to_mmot_obb_format)."""
    import boxmot_amd
    from common import obb_frames
    logging.disable(logging.CRITICAL)
    TrackerRuntime, _ = ref_harness.load_engine_callers()
    if name == "bytetrack":
        ours, ref = boxmot_amd.ByteTrack(max_tracks=128, max_dets=64), ref_harness.load_bytetrack()()
    elif name == "ocsort":
        ours, ref = boxmot_amd.OcSort(max_tracks=128, max_dets=64), ref_harness.load_ocsort()()
    else:
        ours = boxmot_amd.BotSort(reid_model=None, with_reid=False, use_cmc=False, max_tracks=128, max_dets=64)
        ref = ref_harness.load_botsort()(reid_model=None, with_reid=False, use_cmc=False)
    rt_ours, rt_ref = TrackerRuntime(ours), TrackerRuntime(ref)
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    rows = 0
    for t, d in enumerate(obb_frames(40, seed=4)):
        got, _ = rt_ours.update(d.copy(), img)
        want, _ = rt_ref.update(d.copy(), img)
        assert got.shape == want.shape, (t, got.shape, want.shape)
        if len(want):
            assert got.shape[1] == 9 and np.array_equal(got[:, 5:], want[:, 5:]), t
            assert np.allclose(got[:, :5], want[:, :5], rtol=0, atol=1e-3), t
            a, b = TrackerRuntime.format_for_mot(got, t + 1), TrackerRuntime.format_for_mot(want, t + 1)
            assert a.shape == b.shape and np.array_equal(a[:, :2], b[:, :2])
            rows += len(want)
    assert rows > 50 and ours.is_obb and ref.is_obb
    ours.close()


def test_mmot_rows_of_oriented_tracks_match_the_reference_formatter(tmp_path):
    """boxmot_amd.replay.format_for_mmot_obb / write_mot_results against convert_to_mmot_obb_format / write_mot_results
    (boxmot/engine/tracking/mot.py:297-344) on the reference's own oriented rows (tests/golden/obb_golden.npz)."""
    from boxmot_amd.replay import format_for_mmot_obb, write_mot_results
    from common import obb_golden_rows
    ref_harness.install_standins()
    from boxmot.engine.tracking import mot as ref_mot
    rows, _, _ = obb_golden_rows("bytetrack")
    ours_txt, ref_txt = tmp_path / "a.txt", tmp_path / "b.txt"
    n = 0
    for t, r in enumerate(rows[:40]):
        got, want = format_for_mmot_obb(r, t + 1), ref_mot.convert_to_mmot_obb_format(r.copy(), t + 1)
        assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want), t
        write_mot_results(ours_txt, got)
        ref_mot.write_mot_results(ref_txt, want)
        n += len(r)
    assert n > 100 and ours_txt.read_text() == ref_txt.read_text()
    with pytest.raises(ValueError, match="at least 9 columns"):
        format_for_mmot_obb(np.zeros((1, 8), np.float32), 1)
