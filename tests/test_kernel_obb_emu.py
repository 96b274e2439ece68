"""The ORIENTED-box copy of the frame step (boxmot_amd/csrc/botsort_step_body.hpp compiled with BM_OBB, namespace bm::obb) on CPU
threads through tests/host_emu, against the OBB oracles (oracle/botsort_obb.py, oracle/bytetrack_obb.py -- pinned on the reference
BotSort / ByteTrack fed 7-column detections): 9-column rows, ids, the 10-state filter.  Test infrastructure for the kernel logic."""
import numpy as np
import pytest

from boxmot_amd.scenario import stress_frames
from common import obb_frames
from emu_util import EmuBotSort
from oracle.botsort import DEFAULTS


def _run(n_frames, seed, kind, threads=64, **kw):
    from oracle.botsort_obb import BotSortObbOracle
    from oracle.bytetrack_obb import ByteTrackObbOracle
    cfg = dict(DEFAULTS)
    if kind == 1:       # ByteTrack on the shared step (boxmot_amd/bytetrack.py): its thresholds in BoT-SORT's fields
        bt = dict(min_conf=0.1, track_thresh=0.45, match_thresh=0.8, track_buffer=25, frame_rate=30)
        bt.update(kw)
        orc = ByteTrackObbOracle(**kw)
        cfg.update(track_low_thresh=bt["min_conf"], track_high_thresh=bt["track_thresh"], new_track_thresh=bt["track_thresh"],
                   match_thresh=bt["match_thresh"], track_buffer=bt["track_buffer"], frame_rate=bt["frame_rate"], with_reid=False,
                   second_match_thresh=0.5, unconfirmed_match_thresh=0.7, fuse_first_associate=True, removed_stracks_buffer=0, kind=1)
    else:
        cfg.update(kw)
        orc = BotSortObbOracle(**kw)
    emu = EmuBotSort(cfg, cap=128, nd=64, dim=32, threads=threads, obb=True)
    embs = [e for _, e in stress_frames(n_frames, seed=seed)]
    with_emb = kind == 0 and cfg["with_reid"]
    try:
        for t, d in enumerate(obb_frames(n_frames, seed=seed)):
            e = embs[t] if with_emb else None
            want = np.asarray(orc.update(d.copy(), None, None if e is None else e.copy()), dtype=np.float32).reshape(-1, 9)
            got = emu.update(d, e if e is not None else np.zeros((len(d), 32), np.float32))
            assert got.shape == want.shape, (t, got.shape, want.shape)
            assert np.array_equal(got[:, 5:], want[:, 5:]), t                       # id, conf, cls, det_ind and the row order: exact
            assert np.allclose(got[:, :5], want[:, :5], rtol=0, atol=1e-4), (t, np.abs(got[:, :5] - want[:, :5]).max())
        for which, recs in ((0, orc.active), (1, orc.lost)):
            d = emu.dump(which)
            assert list(d["ints"][:, 0]) == [r.id for r in recs]
            if d["n"]:
                ref = np.concatenate([np.array([r.mean for r in recs]), np.array([r.cov for r in recs]).reshape(-1, 100)], 1)
                assert np.allclose(d["kf"], ref, rtol=1e-8, atol=1e-10)
    finally:
        emu.close()


@pytest.mark.parametrize("kw", [dict(with_reid=False), dict(with_reid=True), dict(with_reid=True, track_buffer=4, fuse_first_associate=True)])
def test_emulated_obb_step_matches_the_botsort_oracle(kw):
    _run(80, 4, 0, **kw)


@pytest.mark.parametrize("kw", [{}, dict(track_buffer=4, track_thresh=0.6, match_thresh=0.7)])
def test_emulated_obb_step_matches_the_bytetrack_oracle(kw):
    _run(80, 4, 1, **kw)


def test_emulated_obb_step_four_wavefronts():
    _run(50, 9, 0, threads=256, with_reid=True)


@pytest.mark.parametrize("with_reid", [False, True])
def test_emulated_obb_step_with_camera_motion(with_reid):
    """STrack.multi_gmc_obb in the oriented step (kf_warp_wave of the bm::obb layout) against the oracle's restatement under scheduled
    warps: rows exact, the fp64 filter state to 1e-8 (the refit goes through fp32 corner points and back -- the same roundings on both sides)."""
    from boxmot_amd.scenario import camera_warps
    from oracle.botsort_obb import BotSortObbOracle
    cfg = dict(DEFAULTS)
    cfg.update(with_reid=with_reid)
    orc = BotSortObbOracle(with_reid=with_reid)
    emu = EmuBotSort(cfg, cap=128, nd=64, dim=32, obb=True)
    n = 80
    warps = camera_warps(n, seed=4)
    embs = [e for _, e in stress_frames(n, seed=4)]
    try:
        for t, d in enumerate(obb_frames(n, seed=4)):
            e = embs[t] if with_reid else None
            want = np.asarray(orc.update(d.copy(), None, None if e is None else e.copy(), warp=warps[t]), dtype=np.float32).reshape(-1, 9)
            got = emu.update(d, e if e is not None else np.zeros((len(d), 32), np.float32), warp=warps[t])
            assert got.shape == want.shape, (t, got.shape, want.shape)
            assert np.array_equal(got[:, 5:], want[:, 5:]), t
            assert np.allclose(got[:, :5], want[:, :5], rtol=0, atol=2e-4), (t, np.abs(got[:, :5] - want[:, :5]).max())
        for which, recs in ((0, orc.active), (1, orc.lost)):
            dd = emu.dump(which)
            assert list(dd["ints"][:, 0]) == [r.id for r in recs]
            if dd["n"]:
                ref = np.concatenate([np.array([r.mean for r in recs]), np.array([r.cov for r in recs]).reshape(-1, 100)], 1)
                assert np.allclose(dd["kf"], ref, rtol=1e-8, atol=1e-10), np.abs(dd["kf"] - ref).max()
    finally:
        emu.close()


def _config2_golden(key):
    from common import GOLDEN
    g = np.load(GOLDEN / "obb_config2_golden.npz")
    rows, counts, out, o = g[key + "_rows"], g[key + "_counts"], [], 0
    for n in counts:
        out.append(rows[o:o + n])
        o += n
    return out, int(g["frames"])


@pytest.mark.parametrize("key", ["bytetrack", "botsort_reid", "ocsort"])
def test_oriented_steps_at_the_configuration_2_shape_reproduce_the_reference_rows(key):
    """64 oriented detections per frame on 256 tracks, 1080p (all 256 in the first three frames) -- BASELINE configuration 2's shape:
    the emulated oriented steps and the oracles against rows of the reference classes themselves (tests/golden/obb_config2_golden.npz,
    tests/golden/make_obb_golden.py), 40 of its 60 frames."""
    from common import obb_config2_frames
    from emu_util import EmuDeepOcSort
    from oracle.botsort_obb import BotSortObbOracle
    from oracle.bytetrack_obb import ByteTrackObbOracle
    from oracle.deepocsort import DEFAULTS as DD
    from oracle.ocsort_obb import OcSortObbOracle
    want, _ = _config2_golden(key)
    n = 40
    cfg = dict(DEFAULTS)
    if key == "bytetrack":
        cfg.update(track_low_thresh=0.1, track_high_thresh=0.45, new_track_thresh=0.45, match_thresh=0.8, track_buffer=25, frame_rate=30,
                   with_reid=False, second_match_thresh=0.5, unconfirmed_match_thresh=0.7, fuse_first_associate=True, removed_stracks_buffer=0, kind=1)
        orc, emu = ByteTrackObbOracle(), EmuBotSort(cfg, cap=512, nd=256, dim=32, obb=True)
    elif key == "botsort_reid":
        cfg.update(with_reid=True)
        orc, emu = BotSortObbOracle(with_reid=True), EmuBotSort(cfg, cap=512, nd=256, dim=32, obb=True)
    else:
        c = {**DD, "embedding_off": 1, "use_byte": 1, "min_conf": 0.1, "frame_wh": (1920, 1080)}
        orc, emu = OcSortObbOracle(use_byte=True), EmuDeepOcSort(c, cap=512, nd=256, dim=1, obb=True)
    try:
        for t, (d, e) in enumerate(obb_config2_frames(n)):
            if key == "ocsort":
                got, o = emu.update(d, None), orc.update(d.copy())
            elif key == "bytetrack":
                got, o = emu.update(d, np.zeros((len(d), 32), np.float32)), orc.update(d.copy(), None, None)
            else:
                got, o = emu.update(d, e), orc.update(d.copy(), None, e.copy())
            o = np.asarray(o, dtype=np.float32).reshape(-1, 9)
            assert o.shape == want[t].shape and np.array_equal(o, want[t]), (key, t)                   # the oracle: the reference's rows, bit for bit
            assert got.shape == want[t].shape and np.array_equal(got[:, 5:], want[t][:, 5:]), (key, t)
            assert np.allclose(got[:, :5], want[t][:, :5], rtol=0, atol=2e-4), (key, t)
    finally:
        emu.close()
