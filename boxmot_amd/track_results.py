"""Result view returned by ``update`` -- same contract as the reference's
``TrackResults`` (boxmot/trackers/track_results.py:12-31, column layout from
boxmot/trackers/common/detection_layout.py:61-84): a float32 ndarray subclass of
shape (M, 8) ``[x1, y1, x2, y2, id, conf, cls, det_ind]`` with named accessors.
"""
from __future__ import annotations

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
    id = property(lambda self: np.asarray(self[:, 4], dtype=int))
    conf = property(lambda self: np.asarray(self[:, 5]))
    cls = property(lambda self: np.asarray(self[:, 6], dtype=int))
    det_ind = property(lambda self: np.asarray(self[:, 7], dtype=int))

    @property
    def xywh(self):
        b = np.asarray(self[:, :4])
        if b.size == 0:
            return np.empty((0, 4), dtype=np.float32)
        return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1)

    def summary(self):
        return [
            {"id": int(r[4]), "conf": float(r[5]), "cls": int(r[6]),
             "box": {"x1": float(r[0]), "y1": float(r[1]), "x2": float(r[2]), "y2": float(r[3])}}
            for r in np.asarray(self)
        ]

    def to_mot_lines(self, frame_id: int):
        """MOT-challenge rows ``frame,id,left,top,w,h,conf,cls,-1`` (track_results.py save_mot)."""
        return [
            f"{frame_id},{int(r[4])},{r[0]:.2f},{r[1]:.2f},{r[2] - r[0]:.2f},{r[3] - r[1]:.2f},{r[5]:.6f},{int(r[6])},-1"
            for r in np.asarray(self)
        ]
