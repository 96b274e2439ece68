"""Distinct handles on distinct threads (include/boxmot_hip.h "Threads" / "Devices"; the reference's contract, reid_capi.h:61-70):
two trackers driven concurrently from two Python threads return, each, exactly the rows of a single-threaded run; the per-thread
error slot (thread_local g_last_error) stays per thread; a handle reports the device it was created on and keeps it when the
call comes from another thread."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(kind, seed, frames, barrier=None, out=None, key=None):
    from boxmot_amd import BotSort, DeepOcSort
    from boxmot_amd.scenario import Scenario
    sc = Scenario(24, 48, width=640, height=480, emb_dim=64, stream=seed, random_image=False)
    trk = BotSort(use_cmc=False, max_tracks=128, max_dets=64, emb_dim=64) if kind == "botsort" else \
        DeepOcSort(cmc_off=True, emb_dim=64, max_tracks=128, max_dets=64)
    rows = []
    if barrier is not None:
        barrier.wait()              # both threads enter the frame loop together
    for t in range(frames):
        d, e = sc.frame(t)
        rows.append(np.array(trk.update(d, sc.image, e), dtype=np.float32, copy=True))
    trk.close()
    if out is not None:
        out[key] = rows
    return rows


@pytest.mark.parametrize("kinds", [("botsort", "botsort"), ("botsort", "deepocsort")])
def test_two_handles_on_two_threads_equal_the_single_thread_runs(kinds):
    T = 60
    want = [_run(k, s, T) for s, k in enumerate(kinds)]
    got, errs = {}, []
    bar = threading.Barrier(2)

    def work(i):
        try:
            _run(kinds[i], i, T, bar, got, i)
        except BaseException as exc:         # surfaced below: a thread's exception must fail the test
            errs.append(exc)
            try:
                bar.abort()
            except Exception:
                pass
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errs, errs
    for i in range(2):
        assert len(got[i]) == T
        for t in range(T):
            assert got[i][t].shape == want[i][t].shape and np.array_equal(got[i][t], want[i][t]), (kinds[i], t)
        assert sum(len(r) for r in got[i]) > 0


def test_last_error_is_per_thread_and_handle_keeps_its_device():
    import ctypes

    from boxmot_amd import _lib
    lib = _lib.load()
    cfg = _lib.BotSortConfig()
    lib.boxmot_hip_botsort_default_config(ctypes.byref(cfg))
    cfg.with_reid = 0
    cfg.max_tracks, cfg.max_dets, cfg.emb_dim = 64, 32, 1
    h = lib.boxmot_hip_botsort_create(ctypes.byref(cfg))
    assert h
    assert lib.boxmot_hip_botsort_device(h) == 0 and lib.boxmot_hip_botsort_device(None) == -1
    seen = {}

    def other():
        # a failing call on this thread sets THIS thread's message only; the handle created on the main thread still runs on device 0
        seen["rc"] = lib.boxmot_hip_botsort_reserve(None, 1, 1)
        seen["err"] = lib.boxmot_hip_last_error()
        seen["dev"] = lib.boxmot_hip_botsort_device(h)
        seen["ok"] = lib.boxmot_hip_botsort_reset(h)
    t = threading.Thread(target=other)
    t.start()
    t.join(60)
    assert seen["rc"] == 0 and b"null" in seen["err"].lower()
    assert seen["dev"] == 0 and seen["ok"] == 1
    assert lib.boxmot_hip_last_error() in (b"", None)            # the main thread saw no failure
    lib.boxmot_hip_botsort_destroy(h)
