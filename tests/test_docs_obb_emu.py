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
           "use_byte": int(kw.get("use_byte", False)), "min_conf": kw.get("min_conf", 0.1), "frame_wh": kw.get("frame_wh", (640, 480))}
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
                                dict(use_byte=True, max_age=8, min_hits=1),
                                dict(asso_func="centroid", frame_wh=(640, 480), iou_threshold=0.9, use_byte=True)])
def test_emulated_oriented_ocsort_step_matches_the_oracle(kw):
    thawed = _run(100, 4, check_every=10, **kw)
    assert thawed > 3          # the observation-centric re-update (interpolated boxes incl. the angle) ran


def test_emulated_oriented_ocsort_step_four_wavefronts():
    _run(60, 9, threads=256, use_byte=True)


def test_oriented_steps_under_the_address_and_ub_sanitizers():
    """Both oriented frame steps (OC-SORT's with the BYTE round, BoT-SORT's with embeddings) compiled with -fsanitize=address,undefined
    -- the BoT-SORT one with a camera-motion warp per frame (kf_warp_wave of the oriented layout) -- and run on small tables that fill up: no out-of-bounds access of the 6-wide observation /
    5-wide box / 90- and 110-double filter rows, no undefined behaviour; results still equal to the oracles."""
    import glob
    import os
    import subprocess
    import sys
    libasan = sorted(glob.glob("/usr/lib/gcc/x86_64-linux-gnu/*/libasan.so"))
    if not libasan:
        pytest.skip("libasan.so not found")
    code = ("import sys; sys.path[:0]=['.', 'tests']\n"
            "import numpy as np\n"
            "from common import obb_frames\n"
            "from boxmot_amd.scenario import stress_frames\n"
            "from emu_util import EmuBotSort, EmuDeepOcSort\n"
            "from oracle.botsort import DEFAULTS as BD\n"
            "from oracle.botsort_obb import BotSortObbOracle\n"
            "from oracle.deepocsort import DEFAULTS as DD\n"
            "from oracle.ocsort_obb import OcSortObbOracle\n"
            "kw = dict(use_byte=True, max_age=6, min_hits=1)\n"
            "cfg = {**DD, **{k: v for k, v in kw.items() if k in DD}, 'embedding_off': 1, 'use_byte': 1, 'min_conf': 0.1, 'frame_wh': (640, 480)}\n"
            "emu, orc = EmuDeepOcSort(cfg, cap=64, nd=32, dim=1, sanitize=True, obb=True), OcSortObbOracle(**kw)\n"
            "for d in obb_frames(18, seed=7):\n"
            "    g, w = emu.update(d[:32], None), np.asarray(orc.update(d[:32].copy()), dtype=np.float32).reshape(-1, 9)\n"
            "    assert g.shape == w.shape and np.array_equal(g[:, 5:], w[:, 5:])\n"
            "emu.close()\n"
            "cfg = dict(BD); cfg.update(with_reid=True)\n"
            "emu, orc = EmuBotSort(cfg, cap=64, nd=32, dim=32, sanitize=True, obb=True), BotSortObbOracle(with_reid=True)\n"
            "embs = [e for _, e in stress_frames(18, seed=7)]\n"
            "from boxmot_amd.scenario import camera_warps\n"
            "warps = camera_warps(18, seed=7)\n"
            "for t, d in enumerate(obb_frames(18, seed=7)):\n"
            "    g = emu.update(d[:32], embs[t][:32], warp=warps[t])\n"
            "    w = np.asarray(orc.update(d[:32].copy(), None, embs[t][:32].copy(), warp=warps[t]), dtype=np.float32).reshape(-1, 9)\n"
            "    assert g.shape == w.shape and np.array_equal(g[:, 5:], w[:, 5:])\n"
            "emu.close()\nprint('ASAN-OK')\n")
    env = dict(os.environ, LD_PRELOAD=libasan[-1], ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "ASAN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
