"""TEST-ONLY: the part of the C ABI that ``boxmot_amd.BotSort`` calls (include/boxmot_hip.h), answered by the emulated device step
(tests/host_emu: botsort_step.hpp executed unchanged on CPU threads).  It lets the build container -- which has the reference under
/root/reference but no GPU -- run the real host class ``boxmot_amd.BotSort`` under the reference's own callers
(tests/test_reference_callers.py).  Never imported by the package; the shipped library has no CPU path."""
from __future__ import annotations

import ctypes

import numpy as np

from emu_util import CFG_D, CFG_I, EmuBotSort


class EmuHipLib:
    def __init__(self, threads: int = 64):
        self._handles = {}
        self._next = 1
        self._threads = threads
        self._err = b""

    # ---- config / lifetime ----
    def boxmot_hip_botsort_default_config(self, ref):
        pass        # BotSort.__init__ sets every field it uses

    def boxmot_hip_botsort_create(self, ref):
        c = ref._obj
        if c.n_streams != 1 or c.n_class_lists != 1:
            raise NotImplementedError("emulated ABI: one stream, one class list")
        cfg = {k: getattr(c, k) for k in CFG_D}
        cfg.update({k: getattr(c, k) for k in CFG_I if k != "kind"})
        cfg["kind"] = c.tracker_kind
        h = self._next
        self._next += 1
        obb = bool(c.is_obb)
        self._handles[h] = dict(cfg=cfg, cap=c.max_tracks, nd=c.max_dets, dim=c.emb_dim, obb=obb,
                                emu=EmuBotSort(cfg, cap=c.max_tracks, nd=c.max_dets, dim=c.emb_dim, threads=self._threads, obb=obb), warp=None)
        return h

    def boxmot_hip_botsort_destroy(self, h):
        rec = self._handles.pop(h, None)
        if rec:
            rec["emu"].close()

    def boxmot_hip_botsort_reset(self, h):
        rec = self._handles[h]
        rec["emu"].close()
        rec["emu"] = EmuBotSort(rec["cfg"], cap=rec["cap"], nd=rec["nd"], dim=rec["dim"], threads=self._threads, obb=rec["obb"])
        return 1

    def boxmot_hip_botsort_set_warp(self, h, stream, ptr):
        rec = self._handles[h]
        assert not rec["obb"], "the library rejects warps on an oriented-box handle"
        rec["warp"] = None if not ptr else np.ctypeslib.as_array((ctypes.c_double * 6).from_address(ptr)).copy().reshape(2, 3)
        return 1

    # ---- the per-frame call (boxmot_hip_botsort_update_stream) ----
    def boxmot_hip_botsort_update_stream(self, h, stream, class_list, frame_count_set, dets, n, det_cols, embs, emb_rows, emb_cols,
                                         img, rows, cols, ch, out, out_cap, out_cols, out_rows_ref, out_is_obb_ref):
        rec = self._handles[h]
        dc, oc = (7, 9) if rec["obb"] else (6, 8)
        if n and det_cols != dc:        # the library's wording (boxmot_hip.hip host_update)
            self._err = b"boxmot_hip live tracking supports AABB detections with 6 columns (7 with is_obb)."
            return 0
        assert out_cols == 9 and class_list == 0 and frame_count_set < 0
        d = np.ctypeslib.as_array((ctypes.c_float * (n * dc)).from_address(dets)).reshape(n, dc).copy() if n else np.empty((0, dc), np.float32)
        e = None
        if embs and emb_rows:
            e = np.ctypeslib.as_array((ctypes.c_float * (emb_rows * emb_cols)).from_address(embs)).reshape(emb_rows, emb_cols).copy()
        try:
            got = rec["emu"].update(d, e, warp=rec["warp"])
        except RuntimeError as exc:
            self._err = str(exc).encode()
            return 0
        rec["warp"] = None
        m = len(got)
        if m > out_cap:
            self._err = b"output capacity"
            return 0
        o = np.ctypeslib.as_array((ctypes.c_float * (out_cap * 9)).from_address(out)).reshape(out_cap, 9)
        o[:m, 8] = 0
        o[:m, :oc] = got
        out_rows_ref._obj.value = m
        out_is_obb_ref._obj.value = int(rec["obb"])
        return 1

    # ---- read-only introspection ----
    def boxmot_hip_botsort_capacity(self, h, a, b, c):
        rec = self._handles[h]
        a._obj.value, b._obj.value, c._obj.value = rec["cap"], rec["nd"], 0
        return 1

    def boxmot_hip_botsort_state_dump(self, h, stream, which, class_list, ints, kf, smooth, misc, rows, fc, ic):
        rec = self._handles[h]
        d = rec["emu"].dump(which)
        for ptr, arr, ct in ((ints, d["ints"], ctypes.c_int32), (kf, d["kf"], ctypes.c_double), (smooth, d["smooth"], ctypes.c_float),
                             (misc, d["misc"], ctypes.c_float)):
            if ptr and arr.size:
                np.ctypeslib.as_array((ct * arr.size).from_address(ptr))[:] = arr.reshape(-1)
        rows._obj.value = d["n"]
        fc._obj.value, ic._obj.value = int(d["counters"][0]), int(d["counters"][1])
        return 1

    def boxmot_hip_last_error(self):
        return self._err
