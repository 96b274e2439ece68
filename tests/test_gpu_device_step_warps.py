"""Camera-motion warps through the DEVICE-RESIDENT steps of StrongSORT and DeepOCSORT: a warp set with ``*_set_warp`` is consumed by
the next ``step_device`` (as it is by the host ``update`` and by BoT-SORT's ``step_device``), not left pending.  Compared with the
oracles driven with the same scheduled warps (Track.camera_update, sort/track.py:139-148; KalmanBoxTracker.apply_affine_correction,
deepocsort.py:197-211)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(name, n_frames=40, seed=11):
    import torch

    from boxmot_amd import _lib
    from boxmot_amd.scenario import camera_warps, stress_frames
    lib = _lib.load()
    frames = stress_frames(n_frames, seed=seed, max_objects=18)
    warps = camera_warps(n_frames, seed=seed, every=2)          # every other frame: both "pending" and "none pending" steps
    nd, cap, dim = 64, 256, 32
    dev = torch.device("cuda:0")
    if name == "strongsort":
        from oracle.strongsort import StrongSortOracle
        cfg = _lib.StrongSortConfig()
        lib.boxmot_hip_strongsort_default_config(ctypes.byref(cfg))
        cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim = 1, cap, nd, dim
        h = lib.boxmot_hip_strongsort_create(ctypes.byref(cfg))
        step, set_warp, sync, destroy = (lib.boxmot_hip_strongsort_step_device, lib.boxmot_hip_strongsort_set_warp,
                                          lib.boxmot_hip_strongsort_synchronize, lib.boxmot_hip_strongsort_destroy)
        orc = StrongSortOracle(dot_rule="device")
    else:
        from oracle.deepocsort import DeepOcSortOracle
        cfg = _lib.DeepOcSortConfig()
        lib.boxmot_hip_deepocsort_default_config(ctypes.byref(cfg))
        cfg.cmc_off = 0
        cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim = 1, cap, nd, dim
        h = lib.boxmot_hip_deepocsort_create(ctypes.byref(cfg))
        step, set_warp, sync, destroy = (lib.boxmot_hip_deepocsort_step_device, lib.boxmot_hip_deepocsort_set_warp,
                                          lib.boxmot_hip_deepocsort_synchronize, lib.boxmot_hip_deepocsort_destroy)
        orc = DeepOcSortOracle()       # the warp is supplied per frame (cmc_off is a wrapper-level switch)
    assert h, _lib.last_error()
    d_dets = torch.zeros((nd, 6), dtype=torch.float32, device=dev)
    d_embs = torch.zeros((nd, dim), dtype=torch.float32, device=dev)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)
    d_out = torch.zeros((cap, 8), dtype=torch.float32, device=dev)
    d_out_n = torch.zeros(1, dtype=torch.int32, device=dev)
    identity = np.eye(2, 3)
    try:
        for t, (d, e) in enumerate(frames):
            n = len(d)
            d_dets[:n] = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float32)).to(dev)
            d_embs[:n] = torch.from_numpy(np.ascontiguousarray(e, dtype=np.float32)).to(dev)
            d_n[0] = n
            torch.cuda.synchronize()
            w = np.ascontiguousarray(warps[t], dtype=np.float64)
            pending = t % 2 == 0
            if pending:
                _lib.check(set_warp(h, 0, w.ctypes.data))
            _lib.check(step(h, d_dets.data_ptr(), d_n.data_ptr(), d_embs.data_ptr(), d_out.data_ptr(), d_out_n.data_ptr()))
            _lib.check(sync(h))
            m = int(d_out_n.cpu()[0])
            got = d_out[:m].cpu().numpy()
            want = np.asarray(orc.update(d.copy(), None, e.copy(), warp=w if pending else (identity if name == "strongsort" else None)),
                              dtype=np.float32).reshape(-1, 8)
            assert got.shape == want.shape, (name, t, got.shape, want.shape)
            order_g, order_w = np.argsort(got[:, 4], kind="stable"), np.argsort(want[:, 4], kind="stable")
            assert np.array_equal(got[order_g][:, 4:], want[order_w][:, 4:]), (name, t)
            assert np.allclose(got[order_g][:, :4], want[order_w][:, :4], rtol=0, atol=1e-2), (name, t)
    finally:
        destroy(h)


def test_strongsort_step_device_consumes_pending_warps():
    _run("strongsort")


def test_deepocsort_step_device_consumes_pending_warps():
    _run("deepocsort")


@pytest.mark.parametrize("method", ["sof", "ecc"])
def test_botsort_device_resident_step_estimates_camera_motion_itself(method):
    """cmc_method = "sof" / "ecc" on the handle: the estimator runs inside the DEVICE-RESIDENT step too (on the frames resident in
    HBM), not only in the host updates -- a handle stepped with step_device returns the rows of a handle updated through the host
    API with the same frames, on a panning camera (tracks stay compensated; before round 4 the step silently skipped the estimate)."""
    import torch
    from scipy.ndimage import gaussian_filter

    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    rng = np.random.default_rng(2)
    base = gaussian_filter(rng.integers(0, 255, (700, 1200, 3)).astype(np.float32), (6, 6, 0))
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    sc = Scenario(10, 16, width=960, height=540, emb_dim=32, random_image=False)
    nd = 32
    host = MultiStreamBotSort(1, max_tracks=64, max_dets=nd, emb_dim=32, cmc_method=method, **kw)
    devh = MultiStreamBotSort(1, max_tracks=64, max_dets=nd, emb_dim=32, cmc_method=method, **kw)
    dev = torch.device("cuda:0")
    moved = 0.0
    for t in range(12):
        dets, embs = sc.frame(t)
        ox, oy = 100 + 3 * t, 80 - 2 * t
        dets = dets.copy(); dets[:, [0, 2]] -= 3 * t; dets[:, [1, 3]] += 2 * t
        frame = np.ascontiguousarray(base[oy:oy + 540, ox:ox + 960])
        want = host.update_batch([dets], [frame], [embs])[0]
        d_frame = torch.from_numpy(frame).to(dev)
        ptrs = torch.tensor([d_frame.data_ptr()], dtype=torch.int64, device=dev)
        d_dets = torch.zeros((1, nd, 6), dtype=torch.float32, device=dev)
        d_embs = torch.zeros((1, nd, 32), dtype=torch.float32, device=dev)
        d_dets[0, : len(dets)] = torch.from_numpy(dets).to(dev)
        d_embs[0, : len(dets)] = torch.from_numpy(embs).to(dev)
        d_n = torch.tensor([len(dets)], dtype=torch.int32, device=dev)
        d_out = torch.zeros((1, nd, 8), dtype=torch.float32, device=dev)
        d_out_n = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        devh.step_device(d_dets.data_ptr(), d_n.data_ptr(), d_embs.data_ptr(), ptrs.data_ptr(), 540, 960, d_out.data_ptr(), d_out_n.data_ptr())
        devh.synchronize()
        got = d_out[0, : int(d_out_n[0])].cpu().numpy()
        assert got.shape == np.asarray(want).shape and np.array_equal(got[:, 4:], np.asarray(want)[:, 4:]), t
        assert np.allclose(got[:, :4], np.asarray(want)[:, :4], atol=1e-3), t
        moved = max(moved, float(np.abs(got[:, :4]).max()))
    assert moved > 0
    host.close(); devh.close()
