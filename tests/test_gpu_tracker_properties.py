"""The reference's own tracker behaviour tests (tests/unit/test_trackers.py, tests/performance/test_tracking_p.py),
restated for the HIP backends: property / shape tests, no golden vectors.  The reference downloads ReID checkpoints for
them; here the ReID model is the HIP OSNet-x0.25 with the reference's random initialisation."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL = ["botsort", "deepocsort", "strongsort"]
NO_CMC = {"botsort": dict(use_cmc=False), "deepocsort": dict(cmc_off=True), "strongsort": {}}


@pytest.fixture(scope="module")
def reid():
    from boxmot_amd.reid import HipReID
    from boxmot_amd.reid_weights import reference_init_state_dict
    return HipReID(reference_init_state_dict("osnet_x0_25", seed=0), mode=1)


def _make(kind, reid, per_class=False, **kw):
    from boxmot_amd import create_tracker
    return create_tracker(kind, reid_model=reid, per_class=per_class, max_tracks=64, max_dets=16, **NO_CMC[kind], **kw)


@pytest.mark.parametrize("kind", ALL)
def test_tracker_output_size(kind, reid):                      # test_trackers.py:71-92
    trk = _make(kind, reid)
    rgb = np.random.default_rng(0).integers(0, 255, (640, 640, 3), dtype=np.uint8)
    det = np.array([[144, 212, 400, 480, 0.92, 0], [425, 281, 576, 472, 0.91, 65]])
    out = np.empty((0,))
    for _ in range(10):
        out = trk.update(det, rgb)
        if out.shape == (2, 8):
            break
    assert out.shape == (2, 8)
    trk.close()


@pytest.mark.parametrize("kind", ALL)
@pytest.mark.parametrize("dets", [None, np.array([])])
def test_tracker_with_no_detections(kind, dets, reid):          # :517-534
    trk = _make(kind, reid)
    out = trk.update(dets, np.zeros((640, 640, 3), np.uint8), np.random.random(size=(0, 512)))
    assert out.size == 0
    trk.close()


@pytest.mark.parametrize("kind", ["botsort", "deepocsort"])     # PER_CLASS_TRACKERS of tests/test_config.py that exist here
def test_per_class_isolation(kind, reid):                       # :537-557
    trk = _make(kind, reid, per_class=True, nr_classes=3)
    det = np.array([[100, 100, 150, 150, 0.9, 1], [102, 102, 152, 152, 0.9, 2]])
    out = trk.update(det, np.zeros((640, 640, 3), np.uint8), np.random.rand(2, 512))
    assert len(set(out[:, 4].tolist())) == 2, "Each class should get a separate track even if overlapping"
    trk.close()


@pytest.mark.parametrize("kind", ALL)
def test_emb_trackers_requires_matching_embeddings(kind, reid):  # :560-574
    trk = _make(kind, reid)
    with pytest.raises(AssertionError):
        trk.update(np.array([[10, 10, 20, 20, 0.7, 0]]), np.zeros((640, 640, 3), np.uint8), np.random.rand(2, 512))
    trk.close()


@pytest.mark.parametrize("kind", ALL)
def test_invalid_det_array_shape(kind, reid):                   # :577-591
    trk = _make(kind, reid)
    with pytest.raises(AssertionError):
        trk.update(np.random.rand(2, 5), np.zeros((640, 640, 3), np.uint8), np.random.rand(2, 512))
    trk.close()


@pytest.mark.parametrize("kind", ALL)
def test_track_id_stable_over_frames(kind, reid):               # :600-636
    trk = _make(kind, reid)
    det = np.array([[50, 50, 100, 100, 0.95, 3]])
    rgb = np.zeros((640, 640, 3), np.uint8)
    rng = np.random.default_rng(1)
    out = np.empty((0,))
    for _ in range(10):
        out = trk.update(det, rgb, rng.random((1, 512)))
        if out.shape == (1, 8):
            break
    assert out.shape == (1, 8), "Track was not confirmed after warm-up"
    tid = out[0, 4]
    out2 = trk.update(det, rgb, rng.random((1, 512)))
    assert out2.shape == (1, 8) and out2[0, 4] == tid
    trk.close()


def test_dynamic_max_obs_based_on_max_age():                     # :95-98 (BaseTracker behaviour, any tracker)
    from boxmot_amd import DeepOcSort
    with pytest.raises(RuntimeError):                            # max_age 400 exceeds what the device step supports ...
        DeepOcSort(cmc_off=True, max_age=400, embedding_off=True)
    from boxmot_amd import StrongSort
    trk = StrongSort(max_age=400, emb_dim=8, max_tracks=64, max_dets=16)
    assert trk.max_obs == 405                                    # ... but the rule itself is BaseTracker's
    trk.close()


@pytest.mark.parametrize("kind", ALL)
def test_update_time_with_embeddings_supplied(kind, reid):       # test_tracking_p.py:16-58 (5 ms per iteration)
    trk = _make(kind, reid)
    rgb = np.random.default_rng(0).integers(0, 255, (640, 640, 3), dtype=np.uint8)
    det = np.array([[144, 212, 578, 480, 0.82, 0], [425, 281, 576, 472, 0.56, 65]])
    emb = np.random.default_rng(2).random((2, 512))
    trk.update(det, rgb, emb)
    t0 = time.perf_counter()
    for _ in range(100):
        trk.update(det, rgb, emb)
    per_iter = (time.perf_counter() - t0) / 100
    assert per_iter < 0.005, f"{kind}: {per_iter * 1e3:.2f} ms per update"
    trk.close()


@pytest.mark.parametrize("kind", ["botsort", "bytetrack", "deepocsort", "ocsort", "strongsort"])
def test_no_capacity_limits_like_the_reference(kind):
    """The reference's lists have no maximum size; the device tables are created at max_tracks / max_dets and grow when a frame does
    not fit -- one detection more than max_dets, more simultaneous tracks than max_tracks -- with the same rows as a tracker created
    large (tests/test_gpu_capacity.py compares against the oracles).  Beyond what the assignment solver's LDS state holds the
    update fails with a message, never with a silent truncation."""
    from boxmot_amd import BotSort, ByteTrack, DeepOcSort, OcSort, StrongSort
    mk = {"botsort": lambda **kw: BotSort(use_cmc=False, with_reid=False, **kw), "bytetrack": ByteTrack,
          "deepocsort": lambda **kw: DeepOcSort(cmc_off=True, embedding_off=True, **kw), "ocsort": OcSort,
          "strongsort": lambda **kw: StrongSort(emb_dim=8, **kw)}[kind]
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    rng = np.random.default_rng(0)

    def dets(n, x0=0.0):
        x = x0 + 40.0 * (np.arange(n) % 14)
        y = 60.0 * (np.arange(n) // 14)
        return np.stack([x, y, x + 30, y + 50, np.full(n, 0.9), np.zeros(n)], 1).astype(np.float32)

    small, large = mk(max_tracks=16, max_dets=16), mk(max_tracks=512, max_dets=64)
    seq = [dets(16)] * 3 + [dets(17)] + [dets(16) + np.array([3.0, 700.0 * t, 3.0, 700.0 * t, 0, 0], np.float32) for t in range(1, 5)]
    for d in seq:                                  # 16 confirmed tracks, a 17th detection, then 16 new objects elsewhere per frame
        e = rng.standard_normal((len(d), 8)).astype(np.float32)
        a, b = np.asarray(small.update(d, img, e)).reshape(-1, 8), np.asarray(large.update(d, img, e)).reshape(-1, 8)
        assert np.array_equal(a, b)
    cap, nd, grows = small.capacity()
    assert grows >= 1 and nd >= 17 and cap > 16 and large.capacity()[2] == 0
    with pytest.raises(RuntimeError, match="beyond what the assignment solver"):
        small.reserve(max_tracks=200000)
    small.close(); large.close()
