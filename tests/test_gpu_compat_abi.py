"""The reference's own FFI binding, pointed at libboxmot_hip.so.

The ctypes declarations below are the reference's (fixture copies of ``_BotSortCConfig`` and the argtypes of
boxmot/native/trackers/botsort.py:94-146, ``LIVE_UPDATE_WITH_EMBS_ARGTYPES`` / ``LIVE_UPDATE_ARGTYPES`` of
boxmot/native/trackers/_common.py, and the signatures of boxmot/native/cpp/trackers/base/include/boxmot/trackers/base/
reid_capi.h:36-94) -- deliberately NOT taken from boxmot_amd._lib, so this test fails if the exported names, struct layout
or argument order drift from what a maintainer's unchanged binding expects.  Results are compared with the oracle.
"""
import ctypes
from pathlib import Path

import numpy as np
import pytest

from common import assert_rows_match

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


class _BotSortCConfig(ctypes.Structure):                 # native/trackers/botsort.py:94-110
    _fields_ = [
        ("track_high_thresh", ctypes.c_float),
        ("track_low_thresh", ctypes.c_float),
        ("new_track_thresh", ctypes.c_float),
        ("track_buffer", ctypes.c_int),
        ("match_thresh", ctypes.c_float),
        ("proximity_thresh", ctypes.c_float),
        ("appearance_thresh", ctypes.c_float),
        ("cmc_method", ctypes.c_char_p),
        ("frame_rate", ctypes.c_int),
        ("fuse_first_associate", ctypes.c_int),
        ("with_reid", ctypes.c_int),
        ("max_obs", ctypes.c_int),
        ("reid_model_path", ctypes.c_char_p),
        ("reid_preprocess", ctypes.c_char_p),
    ]


class _ByteTrackCConfig(ctypes.Structure):               # native/trackers/bytetrack.py
    _fields_ = [("min_conf", ctypes.c_float), ("track_thresh", ctypes.c_float), ("match_thresh", ctypes.c_float),
                ("track_buffer", ctypes.c_int), ("frame_rate", ctypes.c_int), ("max_obs", ctypes.c_int)]


class _OcSortCConfig(ctypes.Structure):                  # native/trackers/ocsort.py
    _fields_ = [("min_conf", ctypes.c_float), ("det_thresh", ctypes.c_float), ("iou_threshold", ctypes.c_float),
                ("max_age", ctypes.c_int), ("min_hits", ctypes.c_int), ("delta_t", ctypes.c_int), ("use_byte", ctypes.c_int),
                ("inertia", ctypes.c_float), ("q_xy_scaling", ctypes.c_float), ("q_s_scaling", ctypes.c_float),
                ("max_obs", ctypes.c_int)]


_F, _I, _U8 = ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.POINTER(ctypes.c_uint8)
_IP = ctypes.POINTER(ctypes.c_int)
LIVE_UPDATE_WITH_EMBS_ARGTYPES = [ctypes.c_void_p, _F, _I, _I, _F, _I, _I, _U8, _I, _I, _I, _F, _I, _I, _IP, _IP]
LIVE_UPDATE_ARGTYPES = [ctypes.c_void_p, _F, _I, _I, _U8, _I, _I, _I, _F, _I, _I, _IP, _IP]


@pytest.fixture(scope="module")
def lib():
    import torch  # noqa: F401  -- the reference imports torch long before it binds a native tracker; the bundled HIP runtime of
    # the torch wheel must be in the process before a library linked against the system runtime (boxmot_amd/_lib.py)
    import __graft_entry__ as g
    g.build()
    L = ctypes.CDLL(str(ROOT / "boxmot_amd" / "libboxmot_hip.so"))
    L.boxmot_botsort_create.argtypes = [ctypes.POINTER(_BotSortCConfig)]
    L.boxmot_botsort_create.restype = ctypes.c_void_p
    L.boxmot_botsort_destroy.argtypes = [ctypes.c_void_p]
    L.boxmot_botsort_destroy.restype = None
    L.boxmot_botsort_reset.argtypes = [ctypes.c_void_p]
    L.boxmot_botsort_reset.restype = ctypes.c_int
    L.boxmot_botsort_update.argtypes = LIVE_UPDATE_WITH_EMBS_ARGTYPES
    L.boxmot_botsort_update.restype = ctypes.c_int
    for sym in ("boxmot_botsort_last_reid_time_ms", "boxmot_botsort_last_reid_preprocess_time_ms",
                "boxmot_botsort_last_reid_process_time_ms", "boxmot_botsort_last_reid_postprocess_time_ms"):
        fn = getattr(L, sym)
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        fn.restype = ctypes.c_int
    L.boxmot_botsort_last_error.argtypes = []
    L.boxmot_botsort_last_error.restype = ctypes.c_char_p
    for name, cfg in (("bytetrack", _ByteTrackCConfig), ("ocsort", _OcSortCConfig)):
        getattr(L, f"boxmot_{name}_create").argtypes = [ctypes.POINTER(cfg)]
        getattr(L, f"boxmot_{name}_create").restype = ctypes.c_void_p
        getattr(L, f"boxmot_{name}_destroy").argtypes = [ctypes.c_void_p]
        getattr(L, f"boxmot_{name}_destroy").restype = None
        getattr(L, f"boxmot_{name}_reset").argtypes = [ctypes.c_void_p]
        getattr(L, f"boxmot_{name}_update").argtypes = LIVE_UPDATE_ARGTYPES
        getattr(L, f"boxmot_{name}_update").restype = ctypes.c_int
        getattr(L, f"boxmot_{name}_last_error").restype = ctypes.c_char_p
    L.boxmot_reid_capi_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    L.boxmot_reid_capi_create.restype = ctypes.c_int
    L.boxmot_reid_capi_destroy.argtypes = [ctypes.c_void_p]
    L.boxmot_reid_capi_destroy.restype = None
    L.boxmot_reid_capi_feature_dim.argtypes = [ctypes.c_void_p, _IP]
    L.boxmot_reid_capi_compute_features.argtypes = [ctypes.c_void_p, _F, _I, _U8, _I, _I, _I, _F, _I]
    L.boxmot_reid_capi_preprocess.argtypes = [ctypes.c_void_p, _F, _I, _U8, _I, _I, _I]
    L.boxmot_reid_capi_process.argtypes = [ctypes.c_void_p]
    L.boxmot_reid_capi_postprocess.argtypes = [ctypes.c_void_p, _F, _I]
    L.boxmot_reid_capi_last_error.restype = ctypes.c_char_p
    return L


def _fp(a):
    return a.ctypes.data_as(_F)


def _call_update(fn, handle, dets, embs, img, with_embs=True):
    """call_update of native/trackers/_common.py:158-221: (max(N,1), 9) fp32 out buffer, rows [x1,y1,x2,y2,id,conf,cls,det_ind,angle]."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = len(dets)
    out = np.zeros((max(n, 1), 9), dtype=np.float32)
    out_rows, out_is_obb = ctypes.c_int(0), ctypes.c_int(0)
    img = np.ascontiguousarray(img)
    im = img.ctypes.data_as(_U8)
    if with_embs:
        e = None if embs is None else np.ascontiguousarray(embs, dtype=np.float32)
        ok = fn(handle, _fp(dets) if n else None, n, 6, _fp(e) if e is not None and n else None, n if e is not None else 0,
                e.shape[1] if e is not None else 0, im, img.shape[0], img.shape[1], 3, _fp(out), out.shape[0], 9,
                ctypes.byref(out_rows), ctypes.byref(out_is_obb))
    else:
        ok = fn(handle, _fp(dets) if n else None, n, 6, im, img.shape[0], img.shape[1], 3, _fp(out), out.shape[0], 9,
                ctypes.byref(out_rows), ctypes.byref(out_is_obb))
    return ok, out[: out_rows.value, :8].copy(), out_is_obb.value


def _f32(x):
    return float(np.float32(x))


def test_reference_botsort_binding_drives_the_hip_library(lib):
    from boxmot_amd.scenario import stress_frames
    from oracle.botsort import BotSortOracle
    kw = dict(track_high_thresh=0.55, track_low_thresh=0.12, new_track_thresh=0.62, track_buffer=20, match_thresh=0.78,
              proximity_thresh=0.55, appearance_thresh=0.3, frame_rate=30, fuse_first_associate=1, with_reid=1)
    cfg = _BotSortCConfig(kw["track_high_thresh"], kw["track_low_thresh"], kw["new_track_thresh"], kw["track_buffer"],
                          kw["match_thresh"], kw["proximity_thresh"], kw["appearance_thresh"], b"none", kw["frame_rate"],
                          kw["fuse_first_associate"], kw["with_reid"], 50, None, None)
    h = lib.boxmot_botsort_create(ctypes.byref(cfg))
    assert h, lib.boxmot_botsort_last_error()
    # the struct carries float thresholds: the oracle gets the same (fp32-rounded) values; the knobs the C++ reference
    # hard-codes (tracker.cpp:435,465,476) are the Python constructor defaults
    okw = {k: (_f32(v) if isinstance(v, float) else v) for k, v in kw.items()}
    okw["fuse_first_associate"], okw["with_reid"] = True, True
    orc = BotSortOracle(**okw)
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for t, (dets, embs) in enumerate(stress_frames(120, seed=7)):
        ok, got, is_obb = _call_update(lib.boxmot_botsort_update, h, dets, embs, img)
        assert ok == 1, lib.boxmot_botsort_last_error()
        assert is_obb == 0
        assert_rows_match(got, orc.update(dets.copy(), img, embs.copy()), t)
    v = ctypes.c_double(-1.0)
    assert lib.boxmot_botsort_last_reid_time_ms(h, ctypes.byref(v)) == 1 and v.value >= 0.0
    assert lib.boxmot_botsort_reset(h) == 1
    orc2 = BotSortOracle(**okw)
    for t, (dets, embs) in enumerate(stress_frames(20, seed=3)):
        ok, got, _ = _call_update(lib.boxmot_botsort_update, h, dets, embs, img)
        assert ok == 1
        assert_rows_match(got, orc2.update(dets.copy(), img, embs.copy()), t)
    # errors: wrong column count -> 0 + message (c_api.cpp GuardCall convention), handle stays usable
    bad = np.zeros((2, 5), dtype=np.float32)
    out = np.zeros((2, 9), dtype=np.float32)
    r, o = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.boxmot_botsort_update(h, _fp(bad), 2, 5, None, 0, 0, img.ctypes.data_as(_U8), 480, 640, 3, _fp(out), 2, 9,
                                     ctypes.byref(r), ctypes.byref(o)) == 0
    assert lib.boxmot_botsort_last_error()
    lib.boxmot_botsort_destroy(h)
    # an estimator the library does not have: create fails loudly, as the header says
    cfg.cmc_method = b"orb"
    assert not lib.boxmot_botsort_create(ctypes.byref(cfg))
    assert b"camera-motion" in lib.boxmot_botsort_last_error()


@pytest.mark.parametrize("method", ["ecc", "sof"])
def test_reference_botsort_binding_with_cmc_method_ecc(lib, method):
    """cmc_method = "ecc" / "sof" (the YAML default) in the reference's config struct: the library estimates the warp itself on every
    frame (on the device, include/boxmot_hip.h; SOF masked by the frame's detections, botsort.py:142) and applies it -- rows equal the
    oracle tracker fed with the oracle estimator's warps on a panning camera."""
    from boxmot_amd.scenario import Scenario
    from oracle.botsort import BotSortOracle
    from oracle.ecc import EccOracle
    from oracle.sof import SofOracle
    from scipy.ndimage import gaussian_filter
    kw = dict(track_high_thresh=0.5, track_low_thresh=0.1, new_track_thresh=0.6, track_buffer=30, match_thresh=0.8,
              proximity_thresh=0.5, appearance_thresh=0.25, frame_rate=30, fuse_first_associate=0, with_reid=1)
    cfg = _BotSortCConfig(kw["track_high_thresh"], kw["track_low_thresh"], kw["new_track_thresh"], kw["track_buffer"],
                          kw["match_thresh"], kw["proximity_thresh"], kw["appearance_thresh"], method.encode(), kw["frame_rate"],
                          kw["fuse_first_associate"], kw["with_reid"], 50, None, None)
    h = lib.boxmot_botsort_create(ctypes.byref(cfg))
    assert h, lib.boxmot_botsort_last_error()
    okw = {k: (_f32(v) if isinstance(v, float) else v) for k, v in kw.items()}
    okw["fuse_first_associate"], okw["with_reid"] = False, True
    orc, ecc = BotSortOracle(**okw), (EccOracle() if method == "ecc" else SofOracle())
    rng = np.random.default_rng(2)
    base = gaussian_filter(rng.integers(0, 255, (700, 1200, 3)).astype(np.float32), (6, 6, 0))
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    sc = Scenario(10, 16, width=960, height=540, emb_dim=32, random_image=False)
    for t in range(12):
        dets, embs = sc.frame(t)
        ox, oy = 100 + 3 * t, 80 - 2 * t
        dets = dets.copy(); dets[:, [0, 2]] -= 3 * t; dets[:, [1, 3]] += 2 * t
        frame = np.ascontiguousarray(base[oy:oy + 540, ox:ox + 960])
        ok, got, _ = _call_update(lib.boxmot_botsort_update, h, dets, embs, frame)
        assert ok == 1, lib.boxmot_botsort_last_error()
        want = orc.update(dets.copy(), frame, embs.copy(), warp=ecc.apply(frame, None if method == "ecc" else dets).astype(np.float64))
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
        assert np.allclose(got[:, :4], want[:, :4], atol=2e-2), t
    lib.boxmot_botsort_destroy(h)


def test_reference_botsort_binding_with_reid_weights(lib, tmp_path):
    """with_reid = 1, reid_model_path given, embs = NULL: the library runs its own ReID like the reference's native tracker."""
    from boxmot_amd.reid_weights import pack_osnet, reference_init_state_dict, save_blob
    from boxmot_amd.scenario import Scenario
    from oracle.botsort import BotSortOracle
    from oracle.osnet import OracleReID
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    path = save_blob(pack_osnet(sd), tmp_path / "osnet_x0_25.osn1")
    cfg = _BotSortCConfig(0.5, 0.1, 0.6, 30, 0.8, 0.5, 0.25, None, 30, 0, 1, 50, str(path).encode(), b"resize")
    h = lib.boxmot_botsort_create(ctypes.byref(cfg))
    assert h, lib.boxmot_botsort_last_error()
    orc = BotSortOracle(reid=OracleReID(sd), track_low_thresh=_f32(0.1), new_track_thresh=_f32(0.6), match_thresh=_f32(0.8))
    sc = Scenario(12, 24, width=640, height=480, random_image=True)
    for t in range(8):
        dets, _ = sc.frame(t)
        ok, got, _ = _call_update(lib.boxmot_botsort_update, h, dets, None, sc.image)
        assert ok == 1, lib.boxmot_botsort_last_error()
        want = orc.update(dets, sc.image)
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
    v = ctypes.c_double(0.0)
    assert lib.boxmot_botsort_last_reid_process_time_ms(h, ctypes.byref(v)) == 1 and v.value > 0.0
    lib.boxmot_botsort_destroy(h)


def test_reference_bytetrack_and_ocsort_bindings(lib):
    from boxmot_amd.scenario import stress_frames
    from oracle.bytetrack import ByteTrackOracle
    from oracle.deepocsort import OcSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    bcfg = _ByteTrackCConfig(0.1, 0.45, 0.8, 25, 30, 50)
    h = lib.boxmot_bytetrack_create(ctypes.byref(bcfg))
    assert h, lib.boxmot_bytetrack_last_error()
    orc = ByteTrackOracle(min_conf=_f32(0.1), track_thresh=_f32(0.45), match_thresh=_f32(0.8), track_buffer=25, frame_rate=30)
    for t, (dets, _) in enumerate(stress_frames(100, seed=5)):
        ok, got, _ = _call_update(lib.boxmot_bytetrack_update, h, dets, None, img, with_embs=False)
        assert ok == 1, lib.boxmot_bytetrack_last_error()
        assert_rows_match(got, orc.update(dets.copy(), img), t)
    lib.boxmot_bytetrack_destroy(h)
    ocfg = _OcSortCConfig(0.1, 0.3, 0.3, 30, 3, 3, 1, 0.2, 0.01, 0.0001, 50)
    h = lib.boxmot_ocsort_create(ctypes.byref(ocfg))
    assert h, lib.boxmot_ocsort_last_error()
    orc = OcSortOracle(min_conf=_f32(0.1), use_byte=True, det_thresh=_f32(0.3), iou_threshold=_f32(0.3), max_age=30, min_hits=3,
                       delta_t=3, inertia=_f32(0.2), Q_xy_scaling=_f32(0.01), Q_s_scaling=_f32(0.0001))
    for t, (dets, _) in enumerate(stress_frames(100, seed=5)):
        ok, got, _ = _call_update(lib.boxmot_ocsort_update, h, dets, None, img, with_embs=False)
        assert ok == 1, lib.boxmot_ocsort_last_error()
        want = np.asarray(orc.update(dets.copy(), img), dtype=np.float32).reshape(-1, 8)
        assert_rows_match(got, want, t, box_atol=1e-3)
    lib.boxmot_ocsort_destroy(h)


@pytest.mark.parametrize("mode", ["0", "1"])
def test_reference_reid_capi_binding(lib, tmp_path, monkeypatch, mode):
    """boxmot_reid_capi_*: create(path, preprocess, &handle) -> feature_dim -> compute_features, and the staged
    preprocess -> process -> postprocess calls give the same rows (reid_capi.h:36-94)."""
    from boxmot_amd.reid_weights import pack_osnet, reference_init_state_dict, save_blob
    from oracle.osnet import OracleReID
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    path = save_blob(pack_osnet(sd), tmp_path / "osnet_x0_25.osn1")
    monkeypatch.setenv("BOXMOT_HIP_REID_MODE", mode)
    monkeypatch.setenv("BOXMOT_HIP_REID_MAX_CROPS", "8")       # forces compute_features to walk its boxes in chunks
    img = np.random.default_rng(3).integers(0, 255, (540, 961, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [-10, -5, 60, 120], [900, 400, 961, 540], [100, 100, 100, 150], [5, 5, 300, 40],
                      [400, 20, 480, 300], [10, 10, 138, 266], [600.4, 430.2, 700, 500], [7, 3, 9, 400], [50, 60, 110, 170],
                      [200, 200, 260, 380]], dtype=np.float32)
    for pre, want_pre in ((None, "resize_pad"), (b"resize", "resize")):      # NULL -> the reference's default, resize_pad
        h = ctypes.c_void_p()
        assert lib.boxmot_reid_capi_create(str(path).encode(), pre, ctypes.byref(h)) == 1, lib.boxmot_reid_capi_last_error()
        dim = ctypes.c_int(0)
        assert lib.boxmot_reid_capi_feature_dim(h, ctypes.byref(dim)) == 1 and dim.value == 512
        out = np.zeros((len(boxes), 512), dtype=np.float32)
        assert lib.boxmot_reid_capi_compute_features(h, _fp(boxes), len(boxes), img.ctypes.data_as(_U8), 540, 961, 3, _fp(out),
                                                     out.size) == 1, lib.boxmot_reid_capi_last_error()
        want = OracleReID(sd, preprocess=want_pre).get_features(boxes, img)
        assert np.abs(out - want).max() < 1e-3
        assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-4)
        staged = np.zeros((8, 512), dtype=np.float32)
        assert lib.boxmot_reid_capi_preprocess(h, _fp(boxes), 8, img.ctypes.data_as(_U8), 540, 961, 3) == 1
        assert lib.boxmot_reid_capi_process(h) == 1
        assert lib.boxmot_reid_capi_postprocess(h, _fp(staged), staged.size) == 1
        assert np.array_equal(staged, out[:8])
        assert lib.boxmot_reid_capi_process(h) == 0                        # out of order -> error, not garbage
        assert b"preprocess" in lib.boxmot_reid_capi_last_error()
        small = np.zeros(16, dtype=np.float32)
        assert lib.boxmot_reid_capi_compute_features(h, _fp(boxes), 2, img.ctypes.data_as(_U8), 540, 961, 3, _fp(small), 16) == 0
        lib.boxmot_reid_capi_destroy(h)
    h = ctypes.c_void_p()
    assert lib.boxmot_reid_capi_create(str(path).encode(), b"letterbox", ctypes.byref(h)) == 0 and not h.value


# The reference's native trackers take a 6- or 7-column table per call (live_c_api.hpp:22-60) and write 9-column oriented rows with
# out_is_obb = 1 (:118-149); its Python wrappers fix the layout with the first table (native/trackers/*.py: "cannot switch between AABB
# and OBB inputs").  The same calls on this library's reference-named entry points:
def _call_update_obb(fn, handle, dets, img, cols):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = len(dets)
    out = np.zeros((max(n, 1), 9), dtype=np.float32)
    out_rows, out_is_obb = ctypes.c_int(0), ctypes.c_int(0)
    img = np.ascontiguousarray(img)
    ok = fn(handle, _fp(dets) if n else None, n, cols, img.ctypes.data_as(_U8), img.shape[0], img.shape[1], 3, _fp(out), out.shape[0], 9,
            ctypes.byref(out_rows), ctypes.byref(out_is_obb))
    return ok, out[: out_rows.value].copy(), out_is_obb.value


def test_reference_bytetrack_and_ocsort_bindings_with_oriented_tables(lib):
    """ByteTrack and OC-SORT against the oriented oracles (float-typed configuration like the C structs), an empty 0 x 0 table before and
    after, the refused layout switch, reset."""
    from common import obb_frames
    from oracle.bytetrack_obb import ByteTrackObbOracle
    from oracle.ocsort_obb import OcSortObbOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)

    def rows_match(got, want, t):
        want = np.asarray(want, dtype=np.float32).reshape(-1, 9)
        assert got.shape == want.shape and np.array_equal(got[:, 5:], want[:, 5:]), t
        assert np.allclose(got[:, :5], want[:, :5], rtol=0, atol=1e-3), t

    bcfg = _ByteTrackCConfig(0.1, 0.45, 0.8, 25, 30, 50)
    h = lib.boxmot_bytetrack_create(ctypes.byref(bcfg))
    assert h, lib.boxmot_bytetrack_last_error()
    orc = ByteTrackObbOracle(min_conf=_f32(0.1), track_thresh=_f32(0.45), match_thresh=_f32(0.8), track_buffer=25, frame_rate=30)
    ok, got, is_obb = _call_update_obb(lib.boxmot_bytetrack_update, h, np.empty((0, 0), np.float32), img, 0)      # before the layout is known
    assert ok == 1 and len(got) == 0 and is_obb == 0
    orc.update(np.empty((0, 7), np.float32), img)
    for t, d in enumerate(obb_frames(80, seed=5)):
        ok, got, is_obb = _call_update_obb(lib.boxmot_bytetrack_update, h, d, img, 7)
        assert ok == 1, lib.boxmot_bytetrack_last_error()
        assert is_obb == 1
        rows_match(got, orc.update(d.copy(), img), t)
    ok, got, is_obb = _call_update_obb(lib.boxmot_bytetrack_update, h, np.empty((0, 0), np.float32), img, 0)      # keeps the layout
    assert ok == 1 and is_obb == 1
    ok, _, _ = _call_update_obb(lib.boxmot_bytetrack_update, h, np.array([[10, 10, 40, 60, 0.9, 0]], np.float32), img, 6)
    assert ok == 0 and b"cannot switch between AABB and OBB inputs" in lib.boxmot_bytetrack_last_error()
    assert lib.boxmot_bytetrack_reset(h) == 1                                       # after a reset the next table decides again
    ok, got, is_obb = _call_update_obb(lib.boxmot_bytetrack_update, h, np.array([[10, 10, 40, 60, 0.9, 0]], np.float32), img, 6)
    assert ok == 1 and is_obb == 0 and got.shape == (1, 9) and got[0, 8] == 0
    lib.boxmot_bytetrack_destroy(h)

    ocfg = _OcSortCConfig(0.1, 0.3, 0.3, 30, 3, 3, 1, 0.2, 0.01, 0.0001, 50)
    h = lib.boxmot_ocsort_create(ctypes.byref(ocfg))
    assert h, lib.boxmot_ocsort_last_error()
    orc = OcSortObbOracle(min_conf=_f32(0.1), use_byte=True, det_thresh=_f32(0.3), iou_threshold=_f32(0.3), max_age=30, min_hits=3,
                          delta_t=3, inertia=_f32(0.2), Q_xy_scaling=_f32(0.01), Q_s_scaling=_f32(0.0001))
    for t, d in enumerate(obb_frames(80, seed=5)):
        ok, got, is_obb = _call_update_obb(lib.boxmot_ocsort_update, h, d, img, 7)
        assert ok == 1, lib.boxmot_ocsort_last_error()
        assert is_obb == 1
        rows_match(got, orc.update(d.copy(), img), t)
    ok, _, _ = _call_update_obb(lib.boxmot_ocsort_update, h, np.zeros((1, 5), np.float32), img, 5)
    assert ok == 0 and b"6 (AABB) or 7 (OBB) columns" in lib.boxmot_ocsort_last_error()
    lib.boxmot_ocsort_destroy(h)
