"""The ORIENTED copy of the DeepOCSORT / OC-SORT frame step (boxmot_amd/csrc/deepocsort_step_body.hpp compiled with BM_OBB, namespace
bm::obb) on CPU threads through tests/host_emu, against OcSortObbOracle (oracle/ocsort_obb.py -- pinned bit for bit on the reference
OcSort fed 7-column detections): 9-column rows, ids, ages, the 9-state filter of every track.  Test infrastructure for the kernel logic."""
import numpy as np
import pytest

from common import obb_frames
from emu_util import EmuDeepOcSort
from oracle.deepocsort import DEFAULTS as DOCS_DEFAULTS
from oracle.ocsort_obb import OcSortObbOracle


def _run(n_frames, seed, threads=64, check_every=0, **kw):
    cfg = {**DOCS_DEFAULTS, **{k: v for k, v in kw.items() if k in DOCS_DEFAULTS}, "embedding_off": 1,
           "use_byte": int(kw.get("use_byte", False)), "min_conf": kw.get("min_conf", 0.1)}
    orc, emu = OcSortObbOracle(**kw), EmuDeepOcSort(cfg, cap=128, nd=64, dim=1, threads=threads, obb=True)

    def check_state(t):
        od, dd = orc.dump(), emu.dump()
        assert np.array_equal(dd["ints"][:, 0], od["id"]), t
        assert np.array_equal(dd["ints"][:, 1], od["age"]) and np.array_equal(dd["ints"][:, 2], od["time_since_update"]), t
        assert np.array_equal(dd["ints"][:, 3], od["hit_streak"]), t
        if dd["n"]:
            assert np.allclose(dd["kf"][:, :9], od["x"], rtol=1e-8, atol=1e-9), (t, np.abs(dd["kf"][:, :9] - od["x"]).max())
            assert np.allclose(dd["kf"][:, 9:].reshape(-1, 9, 9), od["P"], rtol=1e-7, atol=1e-8), t
        assert dd["counters"][1] == od["count"]
    try:
        thawed = 0
        for t, d in enumerate(obb_frames(n_frames, seed=seed)):
            frozen = {k.id for k in orc.tracks if not k.kf.observed and k.kf.saved is not None}
            want = np.asarray(orc.update(d.copy()), dtype=np.float32).reshape(-1, 9)
            thawed += sum(1 for k in orc.tracks if k.id in frozen and k.kf.observed)
            got = emu.update(d, None)
            assert got.shape == want.shape, (t, got.shape, want.shape)
            assert np.array_equal(got[:, 5:], want[:, 5:]), t                   # id, conf, cls, det_ind and the row order: exact
            assert np.allclose(got[:, :5], want[:, :5], rtol=0, atol=1e-4), (t, np.abs(got[:, :5] - want[:, :5]).max())
            if check_every and t % check_every == 0:
                check_state(t)
        check_state(n_frames)
        return thawed
    finally:
        emu.close()


@pytest.mark.parametrize("kw", [{}, dict(use_byte=True), dict(max_age=5, min_hits=1, delta_t=2, inertia=0.4, iou_threshold=0.2),
                                dict(use_byte=True, max_age=8, min_hits=1)])
def test_emulated_oriented_ocsort_step_matches_the_oracle(kw):
    thawed = _run(100, 4, check_every=10, **kw)
    assert thawed > 3          # the observation-centric re-update (interpolated boxes incl. the angle) ran


def test_emulated_oriented_ocsort_step_four_wavefronts():
    _run(60, 9, threads=256, use_byte=True)
