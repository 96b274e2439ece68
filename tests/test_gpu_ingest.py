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
    ring.close(); a.close(); b.close()
    with pytest.raises(RuntimeError):
        FrameRing(1, 1, 10, 10)
