"""Caller-side adapters (boxmot_amd.runtime): TrackerRuntime (boxmot/engine/tracking/runtime.py:15-128) and run_tracker
(Results._run_tracker, boxmot/engine/tracking/results.py:467-496) -- keyword forwarding, positional fallbacks, result shapes.
CPU tests drive them with stand-in trackers; the GPU test with the HIP BoT-SORT against the oracle."""
import numpy as np
import pytest


class _KwTracker:
    def __init__(self):
        self.calls = []

    def update(self, dets, img, embs=None, masks=None):
        self.calls.append((len(dets), embs is not None, masks is not None))
        return np.c_[dets[:, :4], np.arange(1, len(dets) + 1), dets[:, 4:6], np.arange(len(dets))].astype(np.float32)


class _PlainTracker:
    def update(self, dets, img):
        return np.empty((0, 8), dtype=np.float32)


class _PositionalEmbTracker:
    def update(self, dets, img, features):
        return np.zeros((1, 8), dtype=np.float32)


def test_tracker_runtime_forwards_only_accepted_keywords():
    from boxmot_amd.runtime import TrackerRuntime
    dets = np.array([[10, 10, 50, 90, 0.9, 0], [100, 40, 160, 200, 0.8, 1]], dtype=np.float32)
    img = np.zeros((240, 320, 3), dtype=np.uint8)
    t = _KwTracker()
    rt = TrackerRuntime(t)
    tracks, ms = rt.update(dets, img, embs=np.ones((2, 4), np.float32))
    assert tracks.shape == (2, 8) and tracks.dtype == np.float32 and ms >= 0 and t.calls[-1] == (2, True, False)
    rt.update(dets, img)
    assert t.calls[-1] == (2, False, False)
    rt2 = TrackerRuntime(_PlainTracker())
    assert not rt2._accepts_embs and not rt2._accepts_masks
    tracks, _ = rt2.update(dets, img, embs=np.ones((2, 4), np.float32), masks=np.zeros((2, 4, 4)))
    assert tracks.shape == (0, 8)
    mot = TrackerRuntime.format_for_mot(np.array([[10, 20, 50, 100, 3, 0.9, 0, 1]], dtype=np.float32), 7)
    assert mot.shape == (1, 9) and mot[0, :6].tolist() == [7, 3, 10, 20, 40, 80] and mot[0, 7] == 1       # class + 1 (mot.py:268)
    assert TrackerRuntime.format_for_mot(np.array([]), 1).shape == (0, 0)
    with pytest.raises(ValueError, match="not supported"):
        TrackerRuntime.create("boosttrack")


def test_run_tracker_fallback_chain():
    from boxmot_amd.runtime import run_tracker
    from boxmot_amd.track_results import TrackResults
    dets = np.array([[10, 10, 50, 90, 0.9, 0]], dtype=np.float32)
    img = np.zeros((240, 320, 3), dtype=np.uint8)
    feats = np.ones((1, 4), np.float32)
    assert isinstance(run_tracker(_KwTracker(), dets, img, feats), TrackResults)
    assert run_tracker(_PositionalEmbTracker(), dets, img, feats).shape == (1, 8)      # TypeError on embs= -> positional
    assert run_tracker(_PlainTracker(), dets, img, feats).shape == (0, 8)              # -> no embeddings at all
    assert run_tracker(_PlainTracker(), dets, img).shape == (0, 8)


@pytest.mark.gpu
def test_tracker_runtime_create_drives_the_hip_botsort():
    from boxmot_amd.runtime import TrackerRuntime, run_tracker
    from boxmot_amd.scenario import stress_frames
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    from oracle.botsort import BotSortOracle
    rt = TrackerRuntime.create("botsort", per_class=False, use_cmc=False, max_tracks=128, max_dets=64, emb_dim=32)
    orc = BotSortOracle(**{k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")})
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for t, (d, e) in enumerate(stress_frames(40, seed=4)):
        tracks, ms = rt.update(d, img, embs=e) if t % 2 else (np.asarray(run_tracker(rt.tracker, d, img, e)), 0.0)
        want = orc.update(d.copy(), img, e.copy())
        assert tracks.reshape(-1, 8).shape == want.shape and np.array_equal(tracks.reshape(-1, 8)[:, 4:], want[:, 4:]), t
    rt.tracker.close()
