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
