"""GPU test of the replay path: several cached sequences of different lengths (with frames without detections)
advance as streams of one handle; every sequence must equal its own oracle run fed like the reference's
process_sequence feeds a tracker (replay.py:306-341)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sequences(dim=32):
    from boxmot_amd.replay import CachedSequence
    from boxmot_amd.scenario import stress_frames
    seqs = []
    for k, (n_frames, seed) in enumerate(((70, 3), (45, 8), (90, 12))):
        rows, embs = [], []
        for t, (d, e) in enumerate(stress_frames(n_frames, seed=seed, emb_dim=dim)):       # every 37th frame is empty
            rows.append(np.c_[np.full(len(d), t + 1), d])
            embs.append(e)
        dets = np.concatenate(rows).astype(np.float32)
        seqs.append(CachedSequence(f"seq{k}", np.arange(1, n_frames + 1), dets, np.concatenate(embs).astype(np.float32)))
    return seqs


@pytest.mark.parametrize("kind", ["botsort", "deepocsort", "strongsort"])
def test_replay_all_sequences_at_once_equals_per_sequence_oracle(kind, tmp_path):
    from boxmot_amd.replay import format_for_mot, replay_to_dir
    from oracle.botsort import BotSortOracle
    from oracle.deepocsort import DeepOcSortOracle
    from oracle.strongsort import StrongSortOracle
    seqs = _sequences()
    conf = 0.15
    got = replay_to_dir(seqs, tmp_path, tracker_type=kind, conf_threshold=conf, max_tracks=256, max_dets=64)
    make = {"botsort": BotSortOracle, "deepocsort": lambda: DeepOcSortOracle(), "strongsort": StrongSortOracle}[kind]
    for s in seqs:
        orc, want = make(), []
        for fid in s.frame_ids:
            d, e = s.frame(int(fid))
            keep = d[:, 4] >= conf
            d, e = d[keep], e[keep]
            if not d.size:
                continue
            rows = np.asarray(orc.update(d, None, e.copy())).reshape(-1, 8)
            if rows.size:
                want.append(format_for_mot(rows, int(fid)))
        want = np.vstack(want)
        g = got[s.name]
        assert g.shape == want.shape, s.name
        assert np.array_equal(g[:, [0, 1, 7, 8]], want[:, [0, 1, 7, 8]]), s.name           # frame, id, class, det_ind
        assert np.abs(g[:, 2:6] - want[:, 2:6]).max() <= 1                                  # integer-rounded boxes
        assert np.allclose(g[:, 6], want[:, 6])
        lines = (tmp_path / f"{s.name}.txt").read_text().strip().splitlines()
        assert len(lines) == len(want) and lines[0].split(",")[0] == str(int(want[0, 0]))
