"""Caller-side adapters: the two places the reference's engine calls a tracker from.

* ``TrackerRuntime`` -- boxmot/engine/tracking/runtime.py:15-128: wraps one tracker, ``create(tracker_name, reid_weights, ...)``
  through the factory, ``update(dets, img, embs=None, masks=None) -> (tracks (M, 8) float32, elapsed_ms)`` that only forwards
  the keyword arguments the tracker's ``update`` accepts, ``format_for_mot(tracks, frame_idx)``.
* ``run_tracker`` -- ``Results._run_tracker``, boxmot/engine/tracking/results.py:467-496: the call the live pipeline makes per
  frame (embeddings / masks as keywords, positional fallbacks for trackers that do not take them, result wrapped as
  ``TrackResults``).
Both drive any object with the ``BaseTracker.update`` surface; with ``boxmot_amd`` trackers the tracker time reported is the
device time of the frame step where the handle exposes it.
"""
from __future__ import annotations

import inspect
import time
from typing import Any

import numpy as np

from boxmot_amd.replay import format_for_mot as _format_for_mot
from boxmot_amd.track_results import TrackResults
from boxmot_amd.tracker_zoo import SUPPORTED, create_tracker


class TrackerRuntime:
    """Wrap one tracker instance with timing and formatting helpers (runtime.py:15-128)."""

    def __init__(self, tracker: Any) -> None:
        self.tracker = tracker
        self._accepts_embs = True
        self._accepts_masks = True
        try:
            params = inspect.signature(self.tracker.update).parameters
        except (ValueError, TypeError):
            return
        var_kw = any(p.kind == inspect.Parameter.VAR_KEYWORD for p in params.values())
        self._accepts_embs = "embs" in params or var_kw
        self._accepts_masks = "masks" in params or var_kw

    @classmethod
    def create(cls, tracker_name: str, reid_weights=None, device=None, half: bool = False, per_class: bool = False,
               evolve_param_dict: dict | None = None, target_id: int | None = None, reid_preprocess: str | None = None,
               **overrides) -> "TrackerRuntime":
        name = str(tracker_name).lower()
        if name not in SUPPORTED:
            raise ValueError(f"'{tracker_name}' is not supported. Supported ones are {', '.join(sorted(SUPPORTED))}")
        tracker = create_tracker(tracker_type=name, tracker_config=None, reid_weights=reid_weights, device=device, half=half,
                                 per_class=per_class, evolve_param_dict=evolve_param_dict, reid_preprocess=reid_preprocess,
                                 **overrides)
        if target_id is not None:
            tracker.target_id = target_id
        return cls(tracker)

    @staticmethod
    def _ensure_2d_tracks(tracks) -> np.ndarray:
        arr = np.asarray(tracks, dtype=np.float32)
        if arr.size == 0:
            return arr if arr.ndim == 2 else np.empty((0, 0), dtype=np.float32)
        return arr.reshape(1, -1) if arr.ndim == 1 else arr

    @staticmethod
    def format_for_mot(tracks, frame_idx: int) -> np.ndarray:
        arr = TrackerRuntime._ensure_2d_tracks(tracks)
        if arr.size == 0:
            return np.empty((0, 0), dtype=np.float32)
        return _format_for_mot(arr, frame_idx)

    @property
    def names(self):
        return getattr(self.tracker, "names", None)

    @names.setter
    def names(self, value) -> None:
        setattr(self.tracker, "names", value)

    def update(self, dets, img, embs=None, masks=None):
        t0 = time.perf_counter()
        kwargs = {}
        if embs is not None and self._accepts_embs:
            kwargs["embs"] = embs
        if masks is not None and self._accepts_masks:
            kwargs["masks"] = masks
        tracks = self.tracker.update(dets, img, **kwargs) if kwargs else self.tracker.update(dets, img)
        return self._ensure_2d_tracks(tracks), (time.perf_counter() - t0) * 1000.0


def run_tracker(tracker, dets, frame, features=None, masks=None) -> TrackResults:
    """``Results._run_tracker`` (results.py:467-496)."""
    kwargs: dict[str, Any] = {}
    if features is not None:
        kwargs["embs"] = features
    if masks is not None:
        kwargs["masks"] = masks
    if kwargs:
        try:
            result = tracker.update(dets, frame, **kwargs)
        except TypeError:
            if features is not None:
                try:
                    result = tracker.update(dets, frame, features)
                except TypeError:
                    result = tracker.update(dets, frame)
            else:
                result = tracker.update(dets, frame)
    else:
        result = tracker.update(dets, frame)
    return result if isinstance(result, TrackResults) else TrackResults(result)
