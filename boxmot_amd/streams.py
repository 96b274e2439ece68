"""Many independent camera streams on one GPU, and their sharding over a node.

The reference parallelises only across sequences (one tracker object per
sequence in a process pool, boxmot/engine/eval/replay.py:489-515).  Here S
streams share one handle and every kernel launch advances all of them by one
frame (``update_batch`` for host buffers, ``step_device`` for device-resident
inputs).  Across GPUs the path shards by stream with no data-path collective:
rank r owns streams ``r, r + world, ...``; the only communication is the gather
of the tiny per-frame result rows to rank 0 (``gather_results``, RCCL over xGMI
when the backend is "nccl", gloo on CPU tests).
"""
from __future__ import annotations

import ctypes

import numpy as np

from boxmot_amd import _lib
from boxmot_amd.reid_weights import load_weights
from boxmot_amd.track_results import TrackResults

BOTSORT_KEYS = (
    "track_high_thresh", "track_low_thresh", "new_track_thresh", "track_buffer", "match_thresh",
    "proximity_thresh", "appearance_thresh", "frame_rate", "fuse_first_associate", "with_reid",
    "second_match_thresh", "unconfirmed_match_thresh", "unconfirmed_emb_scale", "removed_stracks_buffer",
)


def shard_streams(n_streams_total: int, rank: int, world: int) -> list[int]:
    """Stream ids owned by ``rank`` (round-robin)."""
    return list(range(rank, n_streams_total, world))


class MultiStreamBotSort:
    def __init__(self, n_streams: int, max_tracks: int = 1024, max_dets: int = 256, emb_dim: int = 512,
                 reid_weights=None, use_cmc: bool = False, cmc_method: str | None = None, is_obb: bool = False, **botsort_kwargs):
        # camera-motion compensation: the warp of a stream is supplied per frame with set_warp() (estimating it from the
        # images is the caller's, as for BotSort(cmc=...)); use_cmc only documents the intent.  cmc_method = "sof" / "ecc"
        # makes the handle estimate it itself from the frames it is given (host updates and device-resident steps alike)
        self.use_cmc = bool(use_cmc) or cmc_method is not None
        unknown = set(botsort_kwargs) - set(BOTSORT_KEYS)
        if unknown:
            raise TypeError(f"unknown BoT-SORT options: {sorted(unknown)}")
        self._lib = _lib.load()
        cfg = _lib.BotSortConfig()
        self._lib.boxmot_hip_botsort_default_config(ctypes.byref(cfg))
        for k, v in botsort_kwargs.items():
            setattr(cfg, k, int(v) if isinstance(v, bool) else v)
        cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim, cfg.n_class_lists = n_streams, max_tracks, max_dets, emb_dim, 1
        # oriented detections (7 columns in, 9 out; include/boxmot_hip.h is_obb): every stream of the handle has the same layout
        self.is_obb = bool(is_obb)
        cfg.is_obb = int(self.is_obb)
        self._det_cols, self._out_cols = (7, 9) if self.is_obb else (6, 8)
        if cmc_method is not None:
            self._cmc_method = cmc_method.encode()          # (kept alive: the struct holds a char pointer)
            cfg.cmc_method = self._cmc_method
        self.n_streams, self.max_tracks, self.max_dets, self.emb_dim = n_streams, max_tracks, max_dets, emb_dim
        self.with_reid = bool(cfg.with_reid)
        self._handle = self._lib.boxmot_hip_botsort_create(ctypes.byref(cfg))
        if not self._handle:
            raise RuntimeError(_lib.last_error())
        self._blob = None
        if reid_weights is not None:
            self._blob = load_weights(reid_weights)
            _lib.check(self._lib.boxmot_hip_botsort_set_reid_blob(self._handle, self._blob.ctypes.data, int(self._blob.size)))

    # ---- host buffers: list of (n_s, 6) dets, optional list of embs / frames ----
    def update_batch(self, dets_list, imgs=None, embs_list=None, ring=None, slot: int = 0):
        """One frame of every stream.  ``imgs``: host frames (uploaded by the call), or ``ring`` + ``slot``: frames already
        submitted to a ``boxmot_amd.ingest.FrameRing`` slot -- the call makes the device wait for that upload, tracks, and
        releases the slot; the host never waits for a frame copy."""
        S = len(dets_list)
        dets = [np.ascontiguousarray(d, dtype=np.float32).reshape(-1, self._det_cols) for d in dets_list]
        rows = np.array([len(d) for d in dets], dtype=np.int32)
        det_ptrs = (ctypes.c_void_p * S)(*[d.ctypes.data if len(d) else None for d in dets])
        emb_ptrs = None
        embs = None
        if embs_list is not None:
            embs = [np.ascontiguousarray(e, dtype=np.float32).reshape(len(d), self.emb_dim) for e, d in zip(embs_list, dets)]
            emb_ptrs = (ctypes.c_void_p * S)(*[e.ctypes.data if len(e) else None for e in embs])
        img_ptrs, ir, ic = None, 1, 1
        keep = []
        if imgs is not None:
            keep = [None if im is None else np.ascontiguousarray(im) for im in imgs]
            first = next((im for im in keep if im is not None), None)
            if first is not None:       # every entry None: all streams keep their previously uploaded frame
                ir, ic = first.shape[0], first.shape[1]
                img_ptrs = (ctypes.c_void_p * S)(*[None if im is None else im.ctypes.data for im in keep])
        cap = max(int(rows.max()) if S else 0, 1)
        outs = [np.empty((cap, 9), dtype=np.float32) for _ in range(S)]
        out_ptrs = (ctypes.c_void_p * S)(*[o.ctypes.data for o in outs])
        out_rows = np.zeros(S, dtype=np.int32)
        if ring is not None:
            stream = self._lib.boxmot_hip_botsort_stream(self._handle)
            ring.wait(slot, stream)
            ok = self._lib.boxmot_hip_botsort_update_batch_frames(
                self._handle, S, det_ptrs, rows.ctypes.data, emb_ptrs, self.emb_dim if embs is not None else 0,
                ctypes.c_void_p(ring.device_frames(slot)), ring.rows, ring.cols, out_ptrs, cap, out_rows.ctypes.data)
            ring.release(slot, stream)
            _lib.check(ok)
            return [TrackResults(o[:n, :self._out_cols].copy()) for o, n in zip(outs, out_rows)]
        _lib.check(self._lib.boxmot_hip_botsort_update_batch(
            self._handle, S, det_ptrs, rows.ctypes.data, emb_ptrs, self.emb_dim if embs is not None else 0,
            img_ptrs, ir, ic, 3, out_ptrs, cap, out_rows.ctypes.data))
        return [TrackResults(o[:n, :self._out_cols].copy()) for o, n in zip(outs, out_rows)]

    # ---- device-resident step: arguments are raw device addresses (ints) ----
    def step_device(self, d_dets: int, d_det_rows: int, d_embs: int | None, d_frames: int | None, rows: int, cols: int,
                    d_out: int, d_out_rows: int) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_step_device(
            self._handle, d_dets, d_det_rows, d_embs, d_frames, rows, cols, d_out, d_out_rows))

    def set_warp(self, stream: int, warp_2x3) -> None:
        """2x3 camera-motion warp (what the reference's ``cmc.apply(img, dets)`` returns) for the NEXT update_batch /
        step_device of ``stream`` (STrack.multi_gmc, botsort_track.py:117-132); ``None`` clears a pending warp."""
        if warp_2x3 is None:
            _lib.check(self._lib.boxmot_hip_botsort_set_warp(self._handle, int(stream), None))
            return
        w = np.ascontiguousarray(np.asarray(warp_2x3, dtype=np.float64)[:2, :3])
        if w.shape != (2, 3):
            raise ValueError(f"warp shape {w.shape}, expected (2, 3)")
        _lib.check(self._lib.boxmot_hip_botsort_set_warp(self._handle, int(stream), w.ctypes.data))

    def synchronize(self) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_synchronize(self._handle))

    def timer_start(self) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_timer_start(self._handle))

    def timer_stop_ms(self) -> float:
        v = ctypes.c_double(0.0)
        _lib.check(self._lib.boxmot_hip_botsort_timer_stop_ms(self._handle, ctypes.byref(v)))
        return float(v.value)

    def reid_kernel_ms(self):
        v, n = ctypes.c_double(0.0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_botsort_reid_kernel_ms(self._handle, ctypes.byref(v), ctypes.byref(n)))
        return float(v.value), int(n.value)

    def set_reid_mode(self, mode: int) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_set_reid_mode(self._handle, int(mode)))

    def status(self) -> np.ndarray:
        st = np.zeros(self.n_streams, dtype=np.int32)
        _lib.check(self._lib.boxmot_hip_botsort_status(self._handle, st.ctypes.data, self.n_streams))
        return st

    def reset(self) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_reset(self._handle))

    def capacity(self) -> tuple[int, int, int]:
        """(max_tracks, max_dets, times the tables grew): the tables grow on demand in the host-API updates; ``reserve`` sizes
        them ahead of a device-resident burst (``step_device`` cannot grow them)."""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_botsort_capacity(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        self.max_tracks, self.max_dets = a.value, b.value
        return a.value, b.value, c.value

    def reserve(self, max_tracks: int = 0, max_dets: int = 0) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_reserve(self._handle, int(max_tracks), int(max_dets)))
        self.capacity()

    def state_dump(self, stream: int, which: int = 0) -> dict:
        cap, dim = self.capacity()[0], self.emb_dim
        ints = np.zeros((cap, 6), dtype=np.int32)
        kf = np.zeros((cap, 110 if self.is_obb else 72), dtype=np.float64)
        smooth = np.zeros((cap, dim), dtype=np.float32)
        misc = np.zeros((cap, 3), dtype=np.float32)
        rows, fc, ic = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_botsort_state_dump(
            self._handle, stream, which, 0, ints.ctypes.data, kf.ctypes.data, smooth.ctypes.data, misc.ctypes.data,
            ctypes.byref(rows), ctypes.byref(fc), ctypes.byref(ic)))
        n = rows.value
        return dict(n=n, ints=ints[:n], kf=kf[:n], smooth=smooth[:n], misc=misc[:n], frame_count=fc.value,
                    id_count=ic.value)

    def debug_costs_enable(self, on: bool = True) -> None:
        """Keep copies of the association cost matrices of every following update (parity tests; off by default)."""
        _lib.check(self._lib.boxmot_hip_botsort_debug_costs_enable(self._handle, int(bool(on))))

    def debug_costs(self, stage: int, plane: int = 0, stream: int = 0) -> np.ndarray:
        """(tracks, detections) fp64 cost matrix of the last update: ``stage`` 0 first / 1 second / 2 unconfirmed association;
        ``plane`` 0 the solver's matrix, 1 ``iou_distance``, 2 ``embedding_distance`` where evaluated (NaN elsewhere) --
        include/boxmot_hip.h, boxmot_hip_botsort_debug_costs."""
        cap, nd = self.capacity()[:2]
        buf = np.zeros(cap * nd, dtype=np.float64)
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_botsort_debug_costs(self._handle, int(stream), int(stage), int(plane), buf.ctypes.data, buf.size,
                                                            ctypes.byref(r), ctypes.byref(c)))
        return buf[: r.value * c.value].reshape(r.value, c.value).copy()

    def close(self) -> None:
        h = getattr(self, "_handle", None)
        if h:
            self._lib.boxmot_hip_botsort_destroy(h)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_results(results, max_dets: int) -> tuple[np.ndarray, np.ndarray]:
    """list of (M_s, 8) rows -> fixed-capacity (S, max_dets, 8) fp32 + (S,) int32 counts."""
    S = len(results)
    buf = np.zeros((S, max_dets, 8), dtype=np.float32)
    cnt = np.zeros(S, dtype=np.int32)
    for s, r in enumerate(results):
        r = np.asarray(r, dtype=np.float32).reshape(-1, 8)
        buf[s, : len(r)] = r
        cnt[s] = len(r)
    return buf, cnt


def gather_results(rows, counts, dst: int = 0):
    """Gather per-rank result buffers to rank ``dst`` with torch.distributed.

    ``rows``: tensor (S_local, T, max_dets, 8) fp32, ``counts``: (S_local, T) int32, same shapes on
    every rank, on the device of the process group's backend (GPU for "nccl" = RCCL, CPU for gloo).
    Returns (list_of_rows, list_of_counts) ordered by rank on ``dst`` and (None, None) elsewhere.
    This is the only collective of the path: <= 2 KB per stream-frame.
    """
    import torch.distributed as dist

    world = dist.get_world_size()
    rank = dist.get_rank()
    if dist.get_backend() == "gloo":
        rows_l = [rows.new_empty(rows.shape) for _ in range(world)] if rank == dst else None
        cnt_l = [counts.new_empty(counts.shape) for _ in range(world)] if rank == dst else None
        dist.gather(rows, rows_l, dst=dst)
        dist.gather(counts, cnt_l, dst=dst)
        return (rows_l, cnt_l) if rank == dst else (None, None)
    # RCCL: all_gather into one flat buffer (gather is emulated with send/recv pairs there; the
    # payload is a few KB, so the single fused all_gather is the cheaper call)
    rows_l = [rows.new_empty(rows.shape) for _ in range(world)]
    cnt_l = [counts.new_empty(counts.shape) for _ in range(world)]
    dist.all_gather(rows_l, rows)
    dist.all_gather(cnt_l, counts)
    return (rows_l, cnt_l) if rank == dst else (None, None)
