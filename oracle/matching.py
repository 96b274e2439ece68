"""Oracle association costs -- TEST INFRASTRUCTURE ONLY.

Restates (array-in / array-out, no track objects):
  * AssociationFunction.iou_batch        boxmot/trackers/association/iou.py:133-150
  * iou_distance                         boxmot/trackers/association/matching.py:46-80
  * embedding_distance (SciPy cdist)     matching.py:85-107
  * fuse_score                           matching.py:139-147
  * linear_assignment (lap.lapjv)        matching.py:28-43
dtype rules kept: track boxes fp64 (from the Kalman mean), detection boxes fp32
(so the detection area is rounded to fp32 before promotion), costs fp64.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.distance import cdist

from oracle.lap import lapjv


def xyxy2xywh32(xyxy32: np.ndarray) -> np.ndarray:
    """geometry.py:10-24 on a float32 (…,4) array (all arithmetic in fp32)."""
    x = np.asarray(xyxy32, dtype=np.float32)
    y = np.copy(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def xywh2xyxy(xywh: np.ndarray) -> np.ndarray:
    """geometry.py:27-42; keeps the input dtype (fp32 for dets, fp64 for tracks)."""
    x = np.asarray(xywh)
    y = np.copy(x)
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def iou_batch(b1: np.ndarray, b2: np.ndarray) -> np.ndarray:
    b2 = np.expand_dims(b2, 0)
    b1 = np.expand_dims(b1, 1)
    xx1 = np.maximum(b1[..., 0], b2[..., 0])
    yy1 = np.maximum(b1[..., 1], b2[..., 1])
    xx2 = np.minimum(b1[..., 2], b2[..., 2])
    yy2 = np.minimum(b1[..., 3], b2[..., 3])
    w = np.maximum(0.0, xx2 - xx1)
    h = np.maximum(0.0, yy2 - yy1)
    wh = w * h
    return wh / (
        (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
        + (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
        - wh
    )


def iou_distance(a_xyxy: np.ndarray, b_xyxy: np.ndarray) -> np.ndarray:
    """1 - IoU; zero-sized inputs give the reference's fp32 empty matrix."""
    na, nb = len(a_xyxy), len(b_xyxy)
    if na == 0 or nb == 0:
        return np.zeros((na, nb), dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        return 1 - iou_batch(a_xyxy, b_xyxy)


def embedding_distance(track_feats: np.ndarray, det_feats: np.ndarray) -> np.ndarray:
    nt, nd = len(track_feats), len(det_feats)
    if nt == 0 or nd == 0:
        return np.zeros((nt, nd), dtype=np.float32)
    tf = np.asarray(track_feats, dtype=np.float32)
    df = np.asarray(det_feats, dtype=np.float32)
    return np.maximum(0.0, cdist(tf, df, "cosine"))


def fuse_score(cost: np.ndarray, det_confs32: np.ndarray) -> np.ndarray:
    if cost.size == 0:
        return cost
    iou_sim = 1 - cost
    confs = np.asarray(det_confs32)
    confs = np.expand_dims(confs, axis=0).repeat(cost.shape[0], axis=0)
    return 1 - iou_sim * confs


def linear_assignment(cost: np.ndarray, thresh: float):
    """Returns (matches (K,2) int, unmatched_rows, unmatched_cols), rows ascending."""
    if cost.size == 0:
        return (
            np.empty((0, 2), dtype=int),
            np.arange(cost.shape[0]),
            np.arange(cost.shape[1]),
        )
    _, x, y = lapjv(cost, extend_cost=True, cost_limit=thresh)
    rows = np.nonzero(x >= 0)[0]
    matches = np.stack([rows, x[rows]], axis=1).astype(int) if len(rows) else np.empty((0, 2), dtype=int)
    return matches, np.where(x < 0)[0], np.where(y < 0)[0]
