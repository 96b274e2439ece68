"""TEST-ONLY: the part of the C ABI that ``boxmot_amd.BotSort`` / ``ByteTrack`` / ``OcSort`` call (include/boxmot_hip.h), answered by the emulated device steps
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
        """the constructor defaults of BotSort, as boxmot_hip_botsort_default_config fills them (boxmot_hip.hip; botsort.py:66-86)"""
        c = ref._obj
        c.track_high_thresh, c.track_low_thresh, c.new_track_thresh, c.track_buffer = 0.5, 0.1, 0.6, 30
        c.match_thresh, c.proximity_thresh, c.appearance_thresh, c.cmc_method = 0.8, 0.5, 0.25, None
        c.frame_rate, c.fuse_first_associate, c.with_reid, c.max_obs = 30, 0, 1, 50
        c.second_match_thresh, c.unconfirmed_match_thresh, c.unconfirmed_emb_scale, c.removed_stracks_buffer = 0.5, 0.7, 2.0, 100
        c.n_streams, c.max_tracks, c.max_dets, c.emb_dim, c.n_class_lists, c.tracker_kind, c.is_obb = 1, 1024, 256, 512, 1, 0, 0

    def boxmot_hip_botsort_create(self, ref):
        c = ref._obj
        cfg = {k: getattr(c, k) for k in CFG_D}
        cfg.update({k: getattr(c, k) for k in CFG_I if k != "kind"})
        cfg["kind"] = c.tracker_kind
        h = self._next
        self._next += 1
        obb = bool(c.is_obb)
        nl = int(c.n_class_lists)
        mk = lambda: EmuBotSort(cfg, cap=c.max_tracks, nd=c.max_dets, dim=c.emb_dim, threads=self._threads, obb=obb, n_lists=nl)
        # streams of a handle are independent trackers: one emulated step each (`emu` = stream 0, the single-stream entry points)
        streams = [mk() for _ in range(c.n_streams)]
        self._handles[h] = dict(cfg=cfg, cap=c.max_tracks, nd=c.max_dets, dim=c.emb_dim, obb=obb, nl=nl, emu=streams[0], streams=streams, warp=None)
        return h

    def boxmot_hip_botsort_destroy(self, h):
        rec = self._handles.pop(h, None)
        if rec:
            for e in rec["streams"]:
                e.close()

    def boxmot_hip_botsort_update_batch(self, h, n_streams, det_ptrs, rows_ptr, emb_ptrs, emb_cols, img_ptrs, ir, ic, ch, out_ptrs, out_cap, out_rows_ptr):
        """one frame of each of the first n_streams streams (include/boxmot_hip.h boxmot_hip_botsort_update_batch)"""
        rec = self._handles[h]
        dc, oc = (7, 9) if rec["obb"] else (6, 8)
        rows = np.ctypeslib.as_array((ctypes.c_int32 * n_streams).from_address(rows_ptr))
        out_rows = np.ctypeslib.as_array((ctypes.c_int32 * n_streams).from_address(out_rows_ptr))
        for s in range(n_streams):
            n = int(rows[s])
            if n < 0:                       # det_rows = -1: the stream is not stepped in this call
                out_rows[s] = 0
                continue
            d = np.ctypeslib.as_array((ctypes.c_float * (n * dc)).from_address(det_ptrs[s])).reshape(n, dc).copy() if n else np.empty((0, dc), np.float32)
            e = None
            if emb_ptrs is not None and n and emb_ptrs[s]:
                e = np.ctypeslib.as_array((ctypes.c_float * (n * emb_cols)).from_address(emb_ptrs[s])).reshape(n, emb_cols).copy()
            got = rec["streams"][s].update(d, e)
            m = len(got)
            assert m <= out_cap
            o = np.ctypeslib.as_array((ctypes.c_float * (out_cap * 9)).from_address(out_ptrs[s])).reshape(out_cap, 9)
            o[:m, 8] = 0
            o[:m, :oc] = got
            out_rows[s] = m
        return 1

    def boxmot_hip_botsort_reset(self, h):
        self._remake(self._handles[h])
        return 1

    def _remake(self, rec):
        for e in rec["streams"]:
            e.close()
        rec["streams"] = [EmuBotSort(rec["cfg"], cap=rec["cap"], nd=rec["nd"], dim=rec["dim"], threads=self._threads, obb=rec["obb"], n_lists=rec["nl"])
                          for _ in rec["streams"]]
        rec["emu"] = rec["streams"][0]

    def boxmot_hip_botsort_set_warp(self, h, stream, ptr):
        rec = self._handles[h]
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
        assert out_cols == 9 and 0 <= class_list < rec["nl"]
        d = np.ctypeslib.as_array((ctypes.c_float * (n * dc)).from_address(dets)).reshape(n, dc).copy() if n else np.empty((0, dc), np.float32)
        e = None
        if embs and emb_rows:
            e = np.ctypeslib.as_array((ctypes.c_float * (emb_rows * emb_cols)).from_address(embs)).reshape(emb_rows, emb_cols).copy()
        try:
            got = rec["emu"].update(d, e, warp=rec["warp"], class_list=class_list, frame_count=frame_count_set)
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

    def boxmot_hip_botsort_reserve(self, h, max_tracks, max_dets):
        """(emulated ABI: only before the first step -- the tables are simply made again at the larger size)"""
        rec = self._handles[h]
        rec["cap"], rec["nd"] = max(rec["cap"], max_tracks), max(rec["nd"], max_dets)
        self._remake(rec)
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

    # ---- OC-SORT / DeepOCSORT without appearance (boxmot_hip_deepocsort_*): one stream, embedding_off ----
    def boxmot_hip_deepocsort_default_config(self, ref):
        """the constructor defaults of DeepOcSort, as boxmot_hip_deepocsort_default_config fills them (deepocsort.py:263-281)"""
        c = ref._obj
        c.det_thresh, c.max_age, c.max_obs, c.min_hits, c.iou_threshold = 0.3, 30, 50, 3, 0.3
        c.delta_t, c.inertia, c.w_association_emb, c.alpha_fixed_emb, c.aw_param = 3, 0.2, 0.5, 0.95, 0.5
        c.embedding_off, c.cmc_off, c.aw_off, c.Q_xy_scaling, c.Q_s_scaling, c.reid_model_path = 0, 0, 0, 0.01, 0.0001, None
        c.n_streams, c.max_tracks, c.max_dets, c.emb_dim, c.use_byte, c.min_conf = 1, 1024, 256, 512, 0, 0.1
        c.asso_func, c.frame_w, c.frame_h, c.is_obb = 0, 0, 0, 0

    def boxmot_hip_bytetrack_default_config(self, ref):
        """boxmot_hip_bytetrack_default_config: ByteTrack's constructor defaults in the shared configuration (bytetrack.py:225-233)"""
        self.boxmot_hip_botsort_default_config(ref)
        c = ref._obj
        c.tracker_kind, c.track_low_thresh, c.track_high_thresh, c.new_track_thresh = 1, 0.1, 0.45, 0.45
        c.match_thresh, c.track_buffer, c.frame_rate, c.second_match_thresh, c.unconfirmed_match_thresh = 0.8, 25, 30, 0.5, 0.7
        c.with_reid, c.fuse_first_associate, c.emb_dim = 0, 1, 1

    # one emulated step per stream; stream 0 is rec["emu"], the others are made when a batch call first touches them
    def _stream_emu(self, rec, s):
        if s == 0:
            return self._docs_emu(rec) if rec["kind"] == "docs" else rec["emu"]
        extra = rec.setdefault("extra", {})
        if s not in extra:
            if rec["kind"] == "docs":
                from emu_util import EmuDeepOcSort
                extra[s] = EmuDeepOcSort(rec["cfg"], cap=rec["cap"], nd=rec["nd"], dim=rec["dim"], threads=self._threads, obb=rec["obb"])
            else:
                from emu_util import EmuStrongSort
                extra[s] = EmuStrongSort(rec["cfg"], cap=rec["cap"], nd=rec["nd"], dim=rec["dim"], threads=self._threads)
        return extra[s]

    def _update_batch(self, h, n_streams, det_ptrs, rows_ptr, emb_ptrs, emb_cols, out_ptrs, out_cap, out_rows_ptr, cols_in, cols_out):
        rec = self._handles[h]
        rows = np.ctypeslib.as_array((ctypes.c_int32 * n_streams).from_address(rows_ptr))
        out_rows = np.ctypeslib.as_array((ctypes.c_int32 * n_streams).from_address(out_rows_ptr))
        use_emb = rec["kind"] == "ss" or not rec["cfg"]["embedding_off"]
        for s in range(n_streams):
            n = int(rows[s])
            if n < 0:                       # det_rows = -1: the stream is not stepped in this call
                out_rows[s] = 0
                continue
            d = np.ctypeslib.as_array((ctypes.c_float * (n * cols_in)).from_address(det_ptrs[s])).reshape(n, cols_in).copy() if n else np.empty((0, cols_in), np.float32)
            e = np.zeros((n, rec["dim"]), np.float32) if rec["kind"] == "ss" else None
            if use_emb and emb_ptrs is not None and n and emb_ptrs[s]:
                e = np.ctypeslib.as_array((ctypes.c_float * (n * emb_cols)).from_address(emb_ptrs[s])).reshape(n, emb_cols).copy()
            got = self._stream_emu(rec, s).update(d, e)
            m = len(got)
            assert m <= out_cap
            o = np.ctypeslib.as_array((ctypes.c_float * (out_cap * 9)).from_address(out_ptrs[s])).reshape(out_cap, 9)
            o[:m, 8] = 0
            o[:m, :cols_out] = got
            out_rows[s] = m
        return 1

    def boxmot_hip_deepocsort_update_batch(self, h, n_streams, det_ptrs, rows_ptr, emb_ptrs, emb_cols, img_ptrs, ir, ic, ch, out_ptrs, out_cap, out_rows_ptr):
        rec = self._handles[h]
        return self._update_batch(h, n_streams, det_ptrs, rows_ptr, emb_ptrs, emb_cols, out_ptrs, out_cap, out_rows_ptr,
                                  7 if rec["obb"] else 6, 9 if rec["obb"] else 8)

    def boxmot_hip_strongsort_update_batch(self, h, n_streams, det_ptrs, rows_ptr, emb_ptrs, emb_cols, img_ptrs, ir, ic, ch, out_ptrs, out_cap, out_rows_ptr):
        return self._update_batch(h, n_streams, det_ptrs, rows_ptr, emb_ptrs, emb_cols, out_ptrs, out_cap, out_rows_ptr, 6, 8)

    def boxmot_hip_deepocsort_create(self, ref):
        from emu_util import ASSO_MODES, EmuDeepOcSort
        c = ref._obj
        if c.is_obb and not c.embedding_off:
            self._err = b"boxmot_hip: oriented detections run on OC-SORT (embedding_off = 1); DeepOCSORT takes axis-aligned boxes only"
            return None
        if c.is_obb and c.asso_func not in (0, 5):
            self._err = b"boxmot_hip: the oriented step has the rotated IoU and the centroid distance (asso_func must be BOXMOT_HIP_ASSO_IOU or _CENTROID)"
            return None
        names = {v: k for k, v in ASSO_MODES.items()}
        cfg = dict(det_thresh=c.det_thresh, iou_threshold=c.iou_threshold, inertia=c.inertia, w_association_emb=c.w_association_emb,
                   alpha_fixed_emb=c.alpha_fixed_emb, aw_param=c.aw_param, Q_xy_scaling=c.Q_xy_scaling, Q_s_scaling=c.Q_s_scaling,
                   min_conf=c.min_conf, max_age=c.max_age, min_hits=c.min_hits, delta_t=c.delta_t, embedding_off=int(c.embedding_off), aw_off=c.aw_off,
                   use_byte=c.use_byte, asso_func=names[c.asso_func], frame_wh=(c.frame_w, c.frame_h))
        h = self._next
        self._next += 1
        # (the step is made at the first update: the library takes the frame size of `centroid` from the first image it sees,
        # docs_need_frame_size in boxmot_hip.hip)
        self._handles[h] = dict(cfg=cfg, cap=c.max_tracks, nd=c.max_dets, dim=1 if c.embedding_off else int(c.emb_dim), obb=bool(c.is_obb), kind="docs",
                                emu=None)
        return h

    def _docs_emu(self, rec, rows=0, cols=0):
        from emu_util import EmuDeepOcSort
        if rec["emu"] is None:
            cfg = dict(rec["cfg"])
            if tuple(cfg["frame_wh"]) == (0, 0):
                cfg["frame_wh"] = (cols, rows)
            rec["emu"] = EmuDeepOcSort(cfg, cap=rec["cap"], nd=rec["nd"], dim=rec["dim"], threads=self._threads, obb=rec["obb"])
        return rec["emu"]

    def boxmot_hip_deepocsort_destroy(self, h):
        rec = self._handles.pop(h, None)
        if rec and rec["emu"]:
            rec["emu"].close()
        for e in (rec or {}).get("extra", {}).values():
            e.close()

    def boxmot_hip_deepocsort_reset(self, h):
        rec = self._handles[h]
        if rec["emu"]:
            rec["emu"].close()
        rec["emu"] = None
        return 1

    def boxmot_hip_deepocsort_update_stream(self, h, stream, frame_count_set, id_count_ref, dets, n, det_cols, embs, emb_rows, emb_cols,
                                            img, rows, cols, ch, out, out_cap, out_cols, out_rows_ref, out_is_obb_ref):
        rec = self._handles[h]
        dc, oc = (7, 9) if rec["obb"] else (6, 8)
        if n and det_cols != dc:
            self._err = (b"boxmot_hip: this handle was created for oriented detections (7 columns)" if rec["obb"] else
                         b"boxmot_hip: oriented detections (7 columns) need a handle created with is_obb = 1 (BoT-SORT, ByteTrack, OC-SORT)")
            return 0
        assert out_cols == 9 and stream == 0 and frame_count_set < 0 and id_count_ref is None
        d = np.ctypeslib.as_array((ctypes.c_float * (n * dc)).from_address(dets)).reshape(n, dc).copy() if n else np.empty((0, dc), np.float32)
        e = None
        if embs and emb_rows and not rec["cfg"]["embedding_off"]:
            e = np.ctypeslib.as_array((ctypes.c_float * (emb_rows * emb_cols)).from_address(embs)).reshape(emb_rows, emb_cols).copy()
        try:
            got = self._docs_emu(rec, rows, cols).update(d, e)
        except RuntimeError as exc:
            self._err = str(exc).encode()
            return 0
        m = len(got)
        if m > out_cap:
            self._err = b"boxmot_hip: output buffer is too small for the current frame."
            return 0
        o = np.ctypeslib.as_array((ctypes.c_float * (out_cap * 9)).from_address(out)).reshape(out_cap, 9)
        o[:m, 8] = 0
        o[:m, :oc] = got
        out_rows_ref._obj.value = m
        out_is_obb_ref._obj.value = int(rec["obb"])
        return 1

    def boxmot_hip_deepocsort_capacity(self, h, a, b, c):
        rec = self._handles[h]
        a._obj.value, b._obj.value, c._obj.value = rec["cap"], rec["nd"], 0
        return 1

    def boxmot_hip_deepocsort_state_dump(self, h, stream, ints, kf, emb, rows, fc, ic):
        d = self._docs_emu(self._handles[h]).dump()
        for ptr, arr, ct in ((ints, d["ints"], ctypes.c_int32), (kf, d["kf"], ctypes.c_double)):
            if ptr and arr.size:
                np.ctypeslib.as_array((ct * arr.size).from_address(ptr))[:] = arr.reshape(-1)
        rows._obj.value = d["n"]
        fc._obj.value, ic._obj.value = int(d["counters"][0]), int(d["counters"][1])
        return 1

    # ---- StrongSORT (boxmot_hip_strongsort_*): one stream, embeddings supplied ----
    def boxmot_hip_strongsort_default_config(self, ref):
        """the constructor defaults of StrongSort, as boxmot_hip_strongsort_default_config fills them (strongsort.py:41-66)"""
        c = ref._obj
        c.max_age, c.min_conf, c.max_cos_dist, c.max_iou_dist, c.n_init, c.nn_budget = 30, 0.1, 0.2, 0.7, 3, 100
        c.mc_lambda, c.ema_alpha, c.reid_model_path, c.n_streams, c.max_tracks, c.max_dets, c.emb_dim = 0.98, 0.9, None, 1, 1024, 256, 512

    def boxmot_hip_strongsort_create(self, ref):
        from emu_util import EmuStrongSort
        c = ref._obj
        if c.reid_model_path:
            raise NotImplementedError("emulated ABI: embeddings from the caller")
        cfg = dict(min_conf=c.min_conf, max_cos_dist=c.max_cos_dist, max_iou_dist=c.max_iou_dist, mc_lambda=c.mc_lambda, ema_alpha=c.ema_alpha,
                   max_age=c.max_age, n_init=c.n_init, nn_budget=c.nn_budget)
        h = self._next
        self._next += 1
        rec = dict(cfg=cfg, cap=c.max_tracks, nd=c.max_dets, dim=int(c.emb_dim), kind="ss", warp=None)
        rec["emu"] = EmuStrongSort(cfg, cap=rec["cap"], nd=rec["nd"], dim=rec["dim"], threads=self._threads)
        self._handles[h] = rec
        return h

    def boxmot_hip_strongsort_destroy(self, h):
        rec = self._handles.pop(h, None)
        if rec:
            rec["emu"].close()
            for e in rec.get("extra", {}).values():
                e.close()

    def boxmot_hip_strongsort_reset(self, h):
        from emu_util import EmuStrongSort
        rec = self._handles[h]
        rec["emu"].close()
        rec["emu"] = EmuStrongSort(rec["cfg"], cap=rec["cap"], nd=rec["nd"], dim=rec["dim"], threads=self._threads)
        return 1

    def boxmot_hip_strongsort_set_warp(self, h, stream, ptr):
        rec = self._handles[h]
        rec["warp"] = None if not ptr else np.ctypeslib.as_array((ctypes.c_double * 6).from_address(ptr)).copy().reshape(2, 3)
        return 1

    def boxmot_hip_strongsort_update(self, h, dets, n, det_cols, embs, emb_rows, emb_cols, img, rows, cols, ch, out, out_cap, out_cols,
                                     out_rows_ref, out_is_obb_ref):
        rec = self._handles[h]
        if n and det_cols != 6:
            self._err = b"boxmot_hip live tracking supports AABB detections with 6 columns."
            return 0
        assert out_cols == 9
        d = np.ctypeslib.as_array((ctypes.c_float * (n * 6)).from_address(dets)).reshape(n, 6).copy() if n else np.empty((0, 6), np.float32)
        e = np.zeros((n, rec["dim"]), np.float32)
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
            self._err = b"boxmot_hip: output buffer is too small for the current frame."
            return 0
        o = np.ctypeslib.as_array((ctypes.c_float * (out_cap * 9)).from_address(out)).reshape(out_cap, 9)
        o[:m, 8] = 0
        o[:m, :8] = got
        out_rows_ref._obj.value = m
        out_is_obb_ref._obj.value = 0
        return 1

    def boxmot_hip_strongsort_track_count(self, h, stream, ref):
        ref._obj.value = int(self._handles[h]["emu"].dump()["n"])
        return 1

    def boxmot_hip_last_error(self):
        return self._err
