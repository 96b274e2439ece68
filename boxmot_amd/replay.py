"""Replay of cached detections + embeddings through the HIP trackers: the caller side of the hot path.

Mirrors the reference's evaluation replay (boxmot/engine/eval/replay.py:216-370 ``process_sequence``): per sequence,
frames come from the ``dets_n_embs`` cache (boxmot/data/dataset.py:307-440 -- ``dets/<seq>.npy`` rows
``[frame, x1, y1, x2, y2, conf, cls]`` sorted by frame, ``embs/<reid>/<preprocess>/<seq>.npy`` aligned rows), frames
without detections are NOT passed to the tracker, an optional confidence threshold drops rows first, and the tracker
output is written in MOT-challenge text format (boxmot/engine/tracking/mot.py:239-271, :318-344).

The reference replays sequences in a process pool, one tracker object per sequence (replay.py:489-515).  Here every
sequence is a stream of ONE device handle and a frame index advances all of them in a single launch set
(``update_batch``; streams whose sequence has no detections at that index, or has ended, are skipped with
``det_rows = -1``), so a whole benchmark split replays on one GPU at once.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from boxmot_amd import _lib

TRACKERS = ("botsort", "bytetrack", "deepocsort", "ocsort", "strongsort")


@dataclass
class CachedSequence:
    name: str
    frame_ids: np.ndarray          # (F,) int, ascending -- every frame of the sequence, with or without detections
    dets: np.ndarray               # (R, 7) fp32 [frame, x1, y1, x2, y2, conf, cls], sorted by frame
    embs: np.ndarray | None        # (R, D) fp32 aligned with dets, or None

    def frame(self, fid: int):
        lo = np.searchsorted(self.dets[:, 0], fid, side="left")
        hi = np.searchsorted(self.dets[:, 0], fid, side="right")
        d = np.asarray(self.dets[lo:hi, 1:], dtype=np.float32)
        e = None if self.embs is None else np.asarray(self.embs[lo:hi], dtype=np.float32)
        return d, e


def load_cached_sequence(name: str, det_path, emb_path=None, frame_ids=None) -> CachedSequence:
    """Read one sequence of a ``dets_n_embs`` cache (dataset.py:307-318).  ``frame_ids`` defaults to 1..max frame."""
    dets = np.load(det_path, mmap_mode="r")
    if dets.ndim != 2 or dets.shape[1] != 7:
        raise ValueError(f"{det_path}: expected (rows, 7) [frame, x1, y1, x2, y2, conf, cls], got {dets.shape}")
    embs = None
    if emb_path is not None:
        embs = np.load(emb_path, mmap_mode="r")
        if embs.ndim == 1:
            embs = embs.reshape(len(dets), -1)
        if embs.shape[0] != dets.shape[0]:
            raise ValueError(f"Row mismatch in {name}")            # dataset.py:318
    if len(dets) and np.any(np.diff(np.asarray(dets[:, 0])) < 0):
        raise ValueError(f"{det_path}: rows are not sorted by frame id")
    if frame_ids is None:
        last = int(dets[-1, 0]) if len(dets) else 0
        frame_ids = np.arange(1, last + 1)
    return CachedSequence(name, np.asarray(frame_ids, dtype=int), dets, embs)


def format_for_mot(tracks: np.ndarray, frame_idx: int) -> np.ndarray:
    """(M, 8) tracker rows -> (M, 9) MOT rows [frame, id, l, t, w, h, conf, cls + 1, det_ind] (mot.py:256-271)."""
    t = np.asarray(tracks, dtype=np.float32)
    if t.size == 0:
        return np.empty((0, 9), dtype=np.float32)
    t = t.reshape(-1, 8)
    tlwh = np.array(t[:, :4], copy=True)
    tlwh[:, 2] -= tlwh[:, 0]
    tlwh[:, 3] -= tlwh[:, 1]
    return np.column_stack((
        np.full((len(t), 1), frame_idx, dtype=np.int32),
        t[:, 4].astype(int).reshape(-1, 1).astype(np.int32),
        tlwh.round().astype(np.int32),
        t[:, 5].reshape(-1, 1),
        (t[:, 6].astype(int) + 1).reshape(-1, 1).astype(np.int32),
        t[:, 7].astype(int).reshape(-1, 1).astype(np.int32),
    ))


def xywha_to_corners(boxes: np.ndarray) -> np.ndarray:
    """(cx, cy, w, h, angle) -> the four corners, ordered top-left, top-right, bottom-right, bottom-left by the reference's rule
    (smallest / largest x + y, smallest / largest y - x), fp32 like the reference (mot.py:27-66)."""
    arr = np.asarray(boxes, dtype=np.float32).reshape(-1, 5)
    corners = np.empty((arr.shape[0], 4, 2), dtype=np.float32)
    for i, (cx, cy, w, h, angle) in enumerate(arr):
        c, sn = float(np.cos(angle)), float(np.sin(angle))
        rot = np.array([[c, -sn], [sn, c]], dtype=np.float32)
        rect = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]], dtype=np.float32)
        corners[i] = rect @ rot.T + np.array([cx, cy], dtype=np.float32)
    ordered = np.empty_like(corners)
    rows = np.arange(len(corners))
    sums, diffs = corners.sum(axis=2), np.diff(corners, axis=2).reshape(len(corners), 4)
    ordered[:, 0] = corners[rows, np.argmin(sums, axis=1)]
    ordered[:, 2] = corners[rows, np.argmax(sums, axis=1)]
    ordered[:, 1] = corners[rows, np.argmin(diffs, axis=1)]
    ordered[:, 3] = corners[rows, np.argmax(diffs, axis=1)]
    return ordered.reshape(len(corners), 8)


def format_for_mmot_obb(tracks: np.ndarray, frame_idx: int) -> np.ndarray:
    """(M, 9) oriented tracker rows -> (M, 13) MMOT rows [frame, id, x1, y1, ..., x4, y4, conf, cls, det_ind]
    (convert_to_mmot_obb_format, mot.py:297-315; TrackerRuntime.format_for_mot picks it for 9-column rows, runtime.py:81-88)."""
    t = np.asarray(tracks, dtype=np.float32)
    if t.size == 0:
        return np.empty((0, 13), dtype=np.float32)
    t = t.reshape(-1, t.shape[-1])
    if t.shape[1] < 9:
        raise ValueError(f"Expected OBB tracking results with at least 9 columns, got {t.shape[1]}")
    col = lambda v: np.asarray(v, dtype=np.float32).reshape(-1, 1)
    return np.concatenate((np.full((len(t), 1), frame_idx, dtype=np.float32), col(t[:, 5].astype(int)), xywha_to_corners(t[:, :5]),
                           col(t[:, 6]), col(t[:, 7].astype(int)), col(t[:, 8].astype(int))), axis=1)


def write_mot_results(txt_path, mot_rows: np.ndarray) -> None:
    """mot.py:318-344 (the file is created even when there is nothing to write; 9-column MOT rows in the fixed format, the 13-column
    MMOT rows of oriented trackers with ``%g``)."""
    txt_path = Path(txt_path)
    txt_path.parent.mkdir(parents=True, exist_ok=True)
    txt_path.touch(exist_ok=True)
    if mot_rows is not None and mot_rows.size:
        if mot_rows.ndim == 1:
            mot_rows = mot_rows.reshape(1, -1)
        with open(txt_path, "a") as fh:
            if mot_rows.shape[1] == 9:
                np.savetxt(fh, mot_rows, fmt="%d,%d,%d,%d,%d,%d,%.6f,%d,%d")
            else:
                np.savetxt(fh, mot_rows, fmt="%g", delimiter=",")


class MultiStreamTracker:
    """S independent trackers of one type in one device handle (host-buffer API, ``update_batch``)."""

    def __init__(self, tracker_type: str, n_streams: int, emb_dim: int, max_tracks: int = 1024, max_dets: int = 256, **kw):
        if tracker_type not in TRACKERS:
            raise NotImplementedError(f"tracker {tracker_type!r} is not implemented on the HIP backend (have: {TRACKERS})")
        self._lib = _lib.load()
        self.kind, self.n_streams, self.emb_dim, self.max_tracks = tracker_type, n_streams, emb_dim, max_tracks
        native = {"ocsort": "deepocsort", "bytetrack": "botsort"}.get(tracker_type, tracker_type)   # OC-SORT = the DeepOCSORT step without its
        prefix = f"boxmot_hip_{native}_"                                          # appearance / camera terms (deepocsort.py OcSort)
        cfg = {"botsort": _lib.BotSortConfig, "deepocsort": _lib.DeepOcSortConfig, "strongsort": _lib.StrongSortConfig}[native]()
        getattr(self._lib, f"boxmot_hip_{'bytetrack' if tracker_type == 'bytetrack' else native}_default_config")(ctypes.byref(cfg))
        fields = {f[0] for f in cfg._fields_}
        if tracker_type == "botsort":
            kw = {k: v for k, v in kw.items() if k not in ("use_cmc", "cmc_method")}
            cfg.n_class_lists = 1
        if tracker_type == "bytetrack":           # constructor names -> the shared step configuration (bytetrack.py:225-250)
            cfg.n_class_lists = 1
            if "min_conf" in kw:
                cfg.track_low_thresh = kw.pop("min_conf")
            if "track_thresh" in kw:
                cfg.track_high_thresh = cfg.new_track_thresh = kw.pop("track_thresh")
        if tracker_type == "ocsort":
            if "asso_func" in kw:                   # name -> BOXMOT_HIP_ASSO_* (use_byte / min_conf are fields of the shared configuration)
                kw["asso_func"] = _lib.ASSO_FUNCS[kw["asso_func"]]
            cfg.embedding_off, cfg.cmc_off = 1, 1
        if tracker_type == "deepocsort":
            if not kw.pop("cmc_off", True):
                raise NotImplementedError("DeepOCSORT camera-motion compensation is not implemented; pass cmc_off=True")
            cfg.cmc_off = 1
            kw.pop("iou_thresh", None)              # YAML key the reference swallows
            if "asso_func" in kw:
                kw["asso_func"] = _lib.ASSO_FUNCS[kw["asso_func"]]
        unknown = set(kw) - fields
        if unknown:
            raise TypeError(f"unknown {tracker_type} options: {sorted(unknown)}")
        for k, v in kw.items():
            setattr(cfg, k, int(v) if isinstance(v, bool) else v)
        cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim = n_streams, max_tracks, max_dets, emb_dim
        self._update = getattr(self._lib, prefix + "update_batch")
        self._destroy = getattr(self._lib, prefix + "destroy")
        self._handle = getattr(self._lib, prefix + "create")(ctypes.byref(cfg))
        if not self._handle:
            raise RuntimeError(_lib.last_error())

    def update_batch(self, dets_list, embs_list):
        """dets_list[s]: (n_s, 6) array, or None to leave stream s untouched in this call.  Returns a list of (M_s, 8) arrays."""
        S = len(dets_list)
        dets = [None if d is None else np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 6) for d in dets_list]
        rows = np.array([-1 if d is None else len(d) for d in dets], dtype=np.int32)
        det_ptrs = (ctypes.c_void_p * S)(*[None if d is None or not len(d) else d.ctypes.data for d in dets])
        if self.kind in ("ocsort", "bytetrack"):
            embs_list = [None] * S                  # appearance is not an input of OC-SORT / ByteTrack
        embs = [None if (d is None or e is None) else np.ascontiguousarray(e, dtype=np.float32).reshape(len(d), self.emb_dim)
                for d, e in zip(dets, embs_list)]
        emb_ptrs = (ctypes.c_void_p * S)(*[None if e is None or not len(e) else e.ctypes.data for e in embs])
        cap = max(int(rows.max()) if S else 0, 1)
        outs = [np.empty((cap, 9), dtype=np.float32) for _ in range(S)]
        out_ptrs = (ctypes.c_void_p * S)(*[o.ctypes.data for o in outs])
        out_rows = np.zeros(S, dtype=np.int32)
        _lib.check(self._update(self._handle, S, det_ptrs, rows.ctypes.data, emb_ptrs, self.emb_dim, None, 1, 1, 3, out_ptrs, cap,
                                out_rows.ctypes.data))
        return [o[:n, :8].copy() for o, n in zip(outs, out_rows)]

    def close(self):
        if getattr(self, "_handle", None):
            self._destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def replay(sequences, tracker_type: str = "botsort", conf_threshold: float = 0.0, max_tracks: int = 1024,
           max_dets: int | None = None, **tracker_kwargs) -> dict:
    """Run ``tracker_type`` over every cached sequence at once; returns {name: (rows, 9) MOT array}.

    Per frame exactly what ``process_sequence`` does (replay.py:306-341): optional ``conf >= conf_threshold`` filter,
    frames without detections are skipped, ``format_for_mot(tracks, frame_id)`` of non-empty outputs."""
    seqs = list(sequences)
    if not seqs:
        return {}
    if any(s.embs is None for s in seqs) and tracker_type not in ("botsort", "ocsort", "bytetrack"):
        raise ValueError("cached embeddings are required (live ReID over cached frames is not part of the replay path)")
    dim = 1 if tracker_type in ("ocsort", "bytetrack") else next((s.embs.shape[1] for s in seqs if s.embs is not None), 1)
    if max_dets is None:
        max_dets = 4
        for s in seqs:
            if len(s.dets):
                _, counts = np.unique(np.asarray(s.dets[:, 0]).astype(int), return_counts=True)
                max_dets = max(max_dets, int(counts.max()))
    if tracker_type == "botsort" and all(s.embs is None for s in seqs):
        tracker_kwargs.setdefault("with_reid", False)
    trk = MultiStreamTracker(tracker_type, len(seqs), dim, max_tracks=max_tracks, max_dets=max_dets, **tracker_kwargs)
    out = {s.name: [] for s in seqs}
    try:
        for t in range(max(len(s.frame_ids) for s in seqs)):
            dets_l, embs_l, fids = [], [], []
            for s in seqs:
                if t >= len(s.frame_ids):
                    dets_l.append(None); embs_l.append(None); fids.append(None)
                    continue
                fid = int(s.frame_ids[t])
                d, e = s.frame(fid)
                if d.size and conf_threshold > 0:
                    keep = d[:, 4] >= conf_threshold
                    d = d[keep]
                    e = e[keep] if e is not None else None
                if not d.size:
                    d = None                         # the tracker does not see this frame
                dets_l.append(d); embs_l.append(e); fids.append(fid)
            if all(d is None for d in dets_l):
                continue
            for s, fid, rows in zip(seqs, fids, trk.update_batch(dets_l, embs_l)):
                if fid is not None and rows.size:
                    out[s.name].append(format_for_mot(rows, fid))
    finally:
        trk.close()
    return {k: (np.vstack(v) if v else np.empty((0, 9), dtype=np.float32)) for k, v in out.items()}


def replay_to_dir(sequences, exp_folder, **kw) -> dict:
    """``replay`` + one ``<seq>.txt`` per sequence under ``exp_folder`` (what TrackEval reads, replay.py:361-362)."""
    res = replay(sequences, **kw)
    for name, rows in res.items():
        p = Path(exp_folder) / f"{name}.txt"
        if p.exists():
            p.unlink()
        write_mot_results(p, rows)
    return res
