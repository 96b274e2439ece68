"""GPU test of the pinned-host frame ingest ring (include/boxmot_hip.h, boxmot_amd/ingest.py): tracking from ring slots gives
the rows of the plain host-frame update_batch, with the upload of frame t + 1 queued before frame t is tracked."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ring_update_batch_equals_host_frame_update_batch():
    from boxmot_amd.ingest import FrameRing
    from boxmot_amd.reid_weights import reference_init_state_dict
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    S, T, H, W = 3, 8, 480, 640
    scs = [Scenario(10, 20, width=W, height=H, random_image=True, stream=s) for s in range(S)]
    rng = np.random.default_rng(0)
    # a different frame per step, so a slot that was not (re)uploaded in time would change the embeddings and the ids
    frames = [np.stack([np.roll(sc.image, 7 * t, axis=1) for sc in scs]) for t in range(T)]
    dets = [[sc.frame(t, with_embs=False)[0] for sc in scs] for t in range(T)]
    a = MultiStreamBotSort(S, max_tracks=64, max_dets=32, emb_dim=512, reid_weights=sd)
    b = MultiStreamBotSort(S, max_tracks=64, max_dets=32, emb_dim=512, reid_weights=sd)
    a.set_reid_mode(1); b.set_reid_mode(1)
    ring = FrameRing(3, S, H, W)
    assert ring.host_view(1).shape == (S, H, W, 3)
    ring.host_view(0)[...] = frames[0]
    ring.submit(0)
    for t in range(T):
        k, k1 = t % 3, (t + 1) % 3
        if t + 1 < T:
            ring.host_done(k1)
            ring.host_view(k1)[...] = frames[t + 1]
            ring.submit(k1)
        got = b.update_batch(dets[t], ring=ring, slot=k)
        want = a.update_batch(dets[t], imgs=list(frames[t]))
        for s in range(S):
            assert np.array_equal(np.asarray(got[s]), np.asarray(want[s])), (t, s)
    with pytest.raises(RuntimeError):
        ring.submit(7)
    held = ring.host_view(2)[0]                 # a slice keeps the slot's view alive: close() must not free the memory under it
    with pytest.raises(RuntimeError):
        ring.close()
    del held
    ring.close(); a.close(); b.close()
    with pytest.raises(RuntimeError):
        FrameRing(1, 1, 10, 10)


def test_ecc_on_device_vs_oracle_golden_and_shifts():
    """boxmot_hip_ecc_* (csrc/cmc_ecc.hpp) against oracle/ecc.py: the small images of the reference's MOT17-mini frames (golden
    fixture) are reproduced from synthetic full-resolution frames, so here the comparison runs on textured full-HD frames with
    known shifts and on a stateful sequence; same iteration counts, warps within 1e-3 px."""
    from boxmot_amd.cmc import HipECC
    from oracle.ecc import EccOracle
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(0)
    base = gaussian_filter(rng.integers(0, 255, (1200, 2100, 3)).astype(np.float32), (8, 8, 0))
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    dev, orc = HipECC(), EccOracle()
    x0, y0 = 90, 60
    for k, (dx, dy) in enumerate([(0, 0), (-27, 14), (5, -3), (0, 0), (40, 40), (-6, 2)]):
        x0, y0 = x0 + dx, y0 + dy
        frame = np.ascontiguousarray(base[y0:y0 + 1080, x0:x0 + 1920])
        got, want = dev.apply(frame), orc.apply(frame)
        assert got.shape == (2, 3) and got.dtype == np.float32
        print(f"frame {k}: device {got[0, 2]:.4f}, {got[1, 2]:.4f} ({dev.last_iterations} it), oracle {want[0, 2]:.4f}, {want[1, 2]:.4f} ({orc.last_iterations} it)")
        assert np.abs(got - want).max() < 1e-3, (k, got, want)
        if k:
            assert dev.last_iterations == orc.last_iterations
    # uncorrelated frame: OpenCV's StsNoConv exit -> identity, and the estimator carries on from the new frame (ecc.py:67-76)
    flat = np.full((1080, 1920, 3), 90, np.uint8)
    assert np.array_equal(dev.apply(flat), np.eye(2, 3, dtype=np.float32)) and np.array_equal(orc.apply(flat), np.eye(2, 3, dtype=np.float32))
    dev.close()


def test_strongsort_with_device_ecc_matches_oracle_with_oracle_ecc():
    """StrongSort(cmc="ecc"): the reference's default pairing (strongsort.py:67,83-86) -- the estimator is asked only while tracks
    exist; ids exact against the oracle tracker fed by the oracle estimator on a panning camera."""
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.strongsort import StrongSort
    from oracle.ecc import EccOracle
    from oracle.strongsort import StrongSortOracle
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(1)
    base = gaussian_filter(rng.integers(0, 255, (700, 1200, 3)).astype(np.float32), (6, 6, 0))
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    sc = Scenario(10, 16, width=960, height=540, emb_dim=32, random_image=False)
    trk = StrongSort(cmc="ecc", max_tracks=64, max_dets=32, emb_dim=32)
    orc, ecc = StrongSortOracle(), EccOracle()
    n_tracks = 0
    for t in range(12):
        dets, embs = sc.frame(t)
        ox, oy = 100 + 3 * t, 80 - 2 * t                                   # the camera pans: boxes move with the background
        dets = dets.copy(); dets[:, [0, 2]] -= 3 * t; dets[:, [1, 3]] += 2 * t
        frame = np.ascontiguousarray(base[oy:oy + 540, ox:ox + 960])
        got = np.asarray(trk.update(dets, frame, embs))
        warp = ecc.apply(frame, None).astype(np.float64) if n_tracks >= 1 else None
        want = orc.update(dets, frame, embs.copy(), warp=warp)
        n_tracks = len(orc.tracks)
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
        assert np.allclose(got[:, :4], want[:, :4], atol=2e-2), t
    trk.close()
