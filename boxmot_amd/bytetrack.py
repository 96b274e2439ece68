"""ByteTrack on MI355X behind the reference plugin surface.

``ByteTrack(...)`` takes the reference constructor's keyword arguments (boxmot/trackers/bbox/bytetrack/bytetrack.py:225-233
plus the BaseTracker ones) and ``update(dets, img, embs=None)`` returns the reference's rows (bytetrack.py:258-408).
BoT-SORT grew out of this tracker and the frame step is the same sequence of stages, so it runs on the BoT-SORT step
kernel in its ByteTrack mode (include/boxmot_hip.h ``tracker_kind = 1``): (x, y, aspect, height) Kalman state with the
XYAH noise model, only the height velocity zeroed for non-tracked tracks, score fusion in the first and the unconfirmed
association, fixed 0.5 / 0.7 thresholds for the second / unconfirmed association, no appearance, no class vote, an
unbounded removed list.  Pinned against the reference class through ``oracle/bytetrack.py``.

Deviation: the id counter is per tracker (the reference's ``BaseTrack._count`` is process-global and is NOT rewound by the
constructor, bytetrack/basetrack.py:16,37-40).  ``per_class=True`` runs on the kernel's per-class active lists (shared lost list,
removed flags and id counter, basetracker.py:213-271).  Oriented detections (7 columns, bytetrack.py:266, :286, :303) run on
the oriented twin of the step (boxmot_amd.botsort).
"""
from __future__ import annotations

from typing import Any

from boxmot_amd.botsort import BotSort


class ByteTrack(BotSort):
    supports_obb = True

    def __init__(self, min_conf: float = 0.1, track_thresh: float = 0.45, match_thresh: float = 0.8, track_buffer: int = 25,
                 frame_rate: int = 30, max_tracks: int = 1024, max_dets: int = 256, **kwargs: Any):
        for k in ("reid_model", "with_reid", "use_cmc", "cmc", "emb_dim"):
            if k in kwargs:
                raise TypeError(f"ByteTrack() got an unexpected keyword argument {k!r}")
        super().__init__(reid_model=None, track_high_thresh=track_thresh, track_low_thresh=min_conf, new_track_thresh=track_thresh,
                         track_buffer=track_buffer, match_thresh=match_thresh, use_cmc=False, frame_rate=frame_rate,
                         fuse_first_associate=True, with_reid=False, second_match_thresh=0.5, unconfirmed_match_thresh=0.7,
                         max_tracks=max_tracks, max_dets=max_dets, emb_dim=1, _tracker_kind=1, _tracker_name="ByteTrack", **kwargs)
        self.min_conf, self.track_thresh, self.det_thresh = min_conf, track_thresh, track_thresh
        self.track_buffer = track_buffer

    def _update_impl(self, dets, img, embs=None, masks=None, class_list: int = 0):
        return super()._update_impl(dets, img, None, masks, class_list)        # appearance is not an input of ByteTrack
