"""The sparse-optical-flow camera-motion estimator on the device (boxmot_amd.cmc.HipSOF -> boxmot_hip_sof_*, csrc/cmc_sof.hpp) against
the oracle (oracle/sof.py; tests/test_sof.py pins it on the reference's SOF class): every branch of sof.py:55-129, the reference's
MOT17-mini frames, multi-stream handles on frames resident in HBM, and the trackers created with the reference's defaults."""
import ctypes

import numpy as np
import pytest

from common import GOLDEN, assert_rows_match
from test_sof import _sequence, _textured

pytestmark = pytest.mark.gpu


def _check(est, orc, frames, tol=2e-6):
    for t, (fr, dets) in enumerate(frames):
        want, got = orc.apply(fr, dets), est.apply(fr, dets)
        assert got.dtype == np.float32 and got.shape == (2, 3)
        assert np.allclose(got, want, rtol=0, atol=tol * max(1.0, float(np.abs(want).max()))), (t, got, want)
        kp, ok = est.keypoints(), orc.prev_keypoints
        assert len(kp) == (0 if ok is None else len(ok)), t
        if ok is not None and len(ok):
            assert np.abs(kp - ok).max() <= 1e-4, t
        info = est.last_info
        assert bool(info["initialized"]) == bool(orc.initialized), t
        if "inliers" in orc.last:
            assert (info["tracked"], info["inliers"]) == (orc.last["matches"], orc.last["inliers"]), t
        elif "status" in orc.last:
            assert info["mode"] == 2 and info["tracked"] == int((orc.last["status"] == 1).sum()), t


def test_every_branch_vs_oracle():
    from boxmot_amd.cmc import HipSOF
    from oracle.sof import SofOracle
    est = HipSOF()
    _check(est, SofOracle(), _sequence())
    est.reset()                                                      # forgets the previous frame: the next call initialises again
    fr, dets = _sequence()[1]
    assert np.array_equal(est.apply(fr, dets), np.eye(2, 3, dtype=np.float32)) and est.last_info["mode"] == 0
    est.close()


def test_full_hd_frames_and_known_translation():
    from boxmot_amd.cmc import HipSOF
    from oracle.sof import SofOracle
    base = _textured(1200, 2100, seed=11, sigma=7)
    dets = np.array([[300, 200, 520, 800, 0.9, 0], [1200, 300, 1400, 900, 0.8, 0], [-40, 900, 200, 1200, 0.7, 0]], dtype=np.float32)
    frames = [(np.ascontiguousarray(base[60 + dy:1140 + dy, 90 + dx:2010 + dx]), dets) for dx, dy in ((0, 0), (-21, 8), (-21, 8), (14, -30))]
    est = HipSOF()
    _check(est, SofOracle(), frames)
    est2 = HipSOF()
    est2.apply(*frames[0])
    # the corner detector's images of that frame, bit for bit: scaled gray frame, minimum-eigenvalue map, detection mask
    from oracle.ecc import preprocess
    from oracle.sof import generate_mask, min_eigen_map
    gray = preprocess(frames[0][0], 0.15)
    assert gray.shape == (162, 288) and np.array_equal(est2.debug_map(2), gray)
    assert np.array_equal(est2.debug_map(0), min_eigen_map(gray))
    assert np.array_equal(est2.debug_map(1), generate_mask(162, 288, dets[:, :4], 0.15))
    w = est2.apply(*frames[1])
    assert abs(w[0, 2] - 21) < 0.5 and abs(w[1, 2] + 8) < 0.5
    est.close(); est2.close()


def test_mot17_frames_golden():
    from boxmot_amd.cmc import HipSOF
    g = np.load(GOLDEN / "sof_golden.npz")
    for seq in ("02", "04"):
        small = np.load(GOLDEN / "ecc_golden.npz")[f"small_{seq}"]
        est = HipSOF(scale=1.0)
        for k in range(len(small)):
            w = est.apply(np.repeat(small[k][:, :, None], 3, axis=2), g[f"dets_{seq}"][k])
            assert np.allclose(w, g[f"warp_{seq}"][k], rtol=0, atol=2e-6), (seq, k)
            assert len(est.keypoints()) == int(g[f"nkps_{seq}"][k])
        assert np.abs(est.keypoints() - g[f"kps_last_{seq}"]).max() <= 1e-4
        est.close()


def test_multi_stream_handle_on_device_frames():
    """boxmot_hip_sof_apply_device: all streams of a handle in one kernel sequence, frames and detections resident in HBM."""
    import torch
    from boxmot_amd import _lib
    from oracle.sof import SofOracle
    lib = _lib.load()
    S, R, C = 3, 360, 520
    seqs = [_sequence(R, C, seed=20 + s)[1:5] for s in range(S)]
    h = lib.boxmot_hip_sof_create(S, R, C, 0.15, 8, 0.2, 3.0)
    assert h, _lib.last_error()
    orcs = [SofOracle() for _ in range(S)]
    for t in range(4):
        frames = [torch.from_numpy(seqs[s][t][0]).cuda() for s in range(S)]
        table = torch.tensor([f.data_ptr() for f in frames], dtype=torch.int64).cuda()
        nd = 4
        dets = torch.zeros((S, nd, 6), dtype=torch.float32)
        n = torch.zeros(S, dtype=torch.int32)
        for s in range(S):
            d = seqs[s][t][1]
            dets[s, :len(d), :d.shape[1]] = torch.from_numpy(np.ascontiguousarray(d))
            n[s] = len(d)
        dets, n = dets.cuda(), n.cuda()
        torch.cuda.synchronize()
        warps, info = np.zeros((S, 6)), np.zeros((S, 8), np.int32)
        _lib.check(lib.boxmot_hip_sof_apply_device(h, table.data_ptr(), dets.data_ptr(), n.data_ptr(), nd, 6, warps.ctypes.data, info.ctypes.data))
        for s in range(S):
            want = orcs[s].apply(*seqs[s][t])
            assert np.allclose(warps[s].reshape(2, 3), want, rtol=0, atol=2e-6 * max(1.0, float(np.abs(want).max()))), (t, s)
            assert bool(info[s, 7]) == bool(orcs[s].initialized)
    lib.boxmot_hip_sof_destroy(h)


def test_trackers_with_the_reference_defaults_on_a_panning_camera():
    """create_tracker("botsort") / ("deepocsort") with the reference's YAML / constructor defaults (cmc_method "sof"; DeepOCSORT's
    built-in estimator): rows equal the oracle trackers fed with the oracle estimator's warps."""
    from boxmot_amd import create_tracker
    from boxmot_amd.cmc import HipSOF
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS, DEEPOCSORT_YAML_DEFAULTS
    from oracle.botsort import BotSortOracle
    from oracle.deepocsort import DeepOcSortOracle
    from oracle.sof import SofOracle
    base = _textured(760, 1260, seed=31, sigma=6)
    for kind in ("botsort", "deepocsort"):
        sc = Scenario(10, 16, width=960, height=540, emb_dim=32, random_image=False)
        trk = create_tracker(kind, emb_dim=32, max_tracks=128, max_dets=32)
        assert isinstance(trk.cmc, HipSOF)
        if kind == "botsort":
            kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
            orc = BotSortOracle(**kw)
        else:
            y = DEEPOCSORT_YAML_DEFAULTS
            orc = DeepOcSortOracle(iou_threshold=y["iou_thresh"], **{k: v for k, v in y.items() if k not in ("cmc_off", "iou_thresh", "asso_func")})
        sof = SofOracle()
        moved = 0
        for t in range(14):
            dets, embs = sc.frame(t)
            ox, oy = 100 + 4 * t, 90 - 3 * t
            dets = dets.copy(); dets[:, [0, 2]] -= 4 * t; dets[:, [1, 3]] += 3 * t
            frame = np.ascontiguousarray(base[oy:oy + 540, ox:ox + 960])
            got = np.asarray(trk.update(dets, frame, embs)).reshape(-1, 8)
            if kind == "botsort":
                warp = sof.apply(frame, np.hstack([dets, np.arange(len(dets)).reshape(-1, 1)]))          # botsort.py:142: the whole table
            else:
                keep = dets[:, 4] > DEEPOCSORT_YAML_DEFAULTS["det_thresh"]
                warp = sof.apply(frame, dets[keep, :4])                                                 # deepocsort.py:347
            moved += int(not np.array_equal(warp, np.eye(2, 3, dtype=np.float32)))
            want = np.asarray(orc.update(dets.copy(), frame, embs.copy(), warp=warp.astype(np.float64)), dtype=np.float32).reshape(-1, 8)
            assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), (kind, t)
            assert np.allclose(got[:, :4], want[:, :4], atol=2e-2), (kind, t)
        assert moved >= 10
        trk.close()
