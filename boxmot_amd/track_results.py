"""Result view returned by ``update`` -- same contract as the reference's
``TrackResults`` (boxmot/trackers/track_results.py:12-31, column layout from
boxmot/trackers/common/detection_layout.py:61-84): a float32 ndarray subclass of
shape (M, 8) ``[x1, y1, x2, y2, id, conf, cls, det_ind]`` -- or (M, 9) ``[cx, cy, w, h, angle, id, conf, cls, det_ind]`` for
oriented boxes -- with named accessors.
"""
from __future__ import annotations

import csv
import io
import json
from pathlib import Path

import numpy as np

COLUMNS = ("x1", "y1", "x2", "y2", "id", "conf", "cls", "det_ind")


class TrackResults(np.ndarray):
    def __new__(cls, data, masks=None):
        arr = np.asarray(data, dtype=np.float32)
        if arr.ndim == 1 and arr.size:
            arr = arr.reshape(1, -1)
        elif arr.size == 0:
            arr = arr.reshape(0, arr.shape[1] if arr.ndim == 2 else 0)
        view = arr.view(cls)
        view._masks = masks
        return view

    def __array_finalize__(self, obj):
        self._masks = getattr(obj, "_masks", None)

    masks = property(lambda self: self._masks)
    is_obb = property(lambda self: bool(self.ndim == 2 and self.shape[1] >= 9))
    xyxy = property(lambda self: np.asarray(self[:, :4]))
    xywha = property(lambda self: np.asarray(self[:, :5]))
    _meta = property(lambda self: 5 if self.is_obb else 4)            # first column after the box
    id = property(lambda self: np.asarray(self[:, self._meta], dtype=int))
    conf = property(lambda self: np.asarray(self[:, self._meta + 1]))
    cls = property(lambda self: np.asarray(self[:, self._meta + 2], dtype=int))
    det_ind = property(lambda self: np.asarray(self[:, self._meta + 3], dtype=int))

    @property
    def xywh(self):
        b = np.asarray(self[:, :4])
        if b.size == 0:
            return np.empty((0, 4), dtype=np.float32)
        return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1)

    def summary(self):
        if self.is_obb:
            return [
                {"id": int(r[5]), "conf": float(r[6]), "cls": int(r[7]),
                 "box": {"cx": float(r[0]), "cy": float(r[1]), "w": float(r[2]), "h": float(r[3]), "angle": float(r[4])}}
                for r in np.asarray(self)
            ]
        return [
            {"id": int(r[4]), "conf": float(r[5]), "cls": int(r[6]),
             "box": {"x1": float(r[0]), "y1": float(r[1]), "x2": float(r[2]), "y2": float(r[3])}}
            for r in np.asarray(self)
        ]

    def to_mot_lines(self, frame_id: int):
        """MOT-challenge rows ``frame,id,left,top,w,h,conf,cls,-1`` (track_results.py save_mot)."""
        if self.is_obb:         # frame,id,cx,cy,w,h,angle,conf,cls,-1
            return [
                f"{frame_id},{int(r[5])},{r[0]:.2f},{r[1]:.2f},{r[2]:.2f},{r[3]:.2f},{r[4]:.4f},{r[6]:.6f},{int(r[7])},-1"
                for r in np.asarray(self)
            ]
        return [
            f"{frame_id},{int(r[4])},{r[0]:.2f},{r[1]:.2f},{r[2] - r[0]:.2f},{r[3] - r[1]:.2f},{r[5]:.6f},{int(r[6])},-1"
            for r in np.asarray(self)
        ]

    # ---- export methods (track_results.py:100-200) ----
    @property
    def _csv_fields(self):
        return ["cx", "cy", "w", "h", "angle", "id", "conf", "cls", "det_ind"] if self.is_obb else list(COLUMNS)

    def _row(self, i: int):
        box = [float(v) for v in (self.xywha[i] if self.is_obb else self.xyxy[i])]
        return box + [int(self.id[i]), float(self.conf[i]), int(self.cls[i]), int(self.det_ind[i])]

    def to_json(self, indent=None) -> str:
        return json.dumps(self.summary(), indent=indent)

    def to_csv(self, frame_id=None) -> str:
        buf = io.StringIO()
        writer = csv.writer(buf)
        for i in range(len(self)):
            writer.writerow([frame_id] + self._row(i) if frame_id is not None else self._row(i))
        return buf.getvalue()

    def save_csv(self, path, frame_id=None, header: bool = True) -> None:
        path = Path(path)
        write_header = header and not path.exists()
        path.parent.mkdir(parents=True, exist_ok=True)
        with open(path, "a", newline="") as f:
            if write_header:
                csv.writer(f).writerow((["frame"] + self._csv_fields) if frame_id is not None else self._csv_fields)
            f.write(self.to_csv(frame_id=frame_id))

    def save_mot(self, path, frame_id: int = 0) -> None:
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        with open(path, "a") as f:
            for line in self.to_mot_lines(frame_id):
                f.write(line + "\n")
