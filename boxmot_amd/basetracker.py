"""Host-side mirror of the reference plugin surface ``BaseTracker.update``.

Same names, argument meaning and error behaviour as
boxmot/trackers/basetracker.py:120-372 (input unwrapping :153-183, mask handling
:185-211, empty input and per-class fan-out :213-271, ``check_inputs`` :356-372,
class split :306-335); the per-frame math itself lives behind the C ABI
(``_update_impl`` of the subclasses).  The detection layout -- axis-aligned (6 columns) or
oriented (7 columns) -- is inferred from the first detection table like the reference does
(basetracker.py:163-173 over common/detection_layout.py:87-108); a tracker class without
``supports_obb`` raises the reference's ``AssertionError`` for oriented input.
"""
from __future__ import annotations

import logging

import numpy as np

from boxmot_amd.track_results import TrackResults

LOGGER = logging.getLogger("boxmot_amd")

AABB_COLS = 6   # x1,y1,x2,y2,conf,cls     detection_layout.py:61-71
OBB_COLS = 7    # cx,cy,w,h,angle,conf,cls  detection_layout.py:74-84
CONF_IDX, CLS_IDX = 4, 5
OUT_COLS = 8
OBB_OUT_COLS = 9
# the keys of AssociationFunction._get_asso_func's table (trackers/association/iou.py:408-417), in its order
ASSO_NAMES = ("iou", "iou_obb", "hmiou", "giou", "ciou", "diou", "centroid", "centroid_obb")


class BaseTracker:
    supports_obb = False
    supports_masks = False

    def __init__(self, det_thresh: float = 0.3, max_age: int = 30, max_obs: int = 50, min_hits: int = 3,
                 iou_threshold: float = 0.3, per_class: bool = False, nr_classes: int = 80, asso_func: str = "iou",
                 is_obb: bool = False, **kwargs):
        if is_obb and not self.supports_obb:
            raise AssertionError(f"{type(self).__name__} does not support OBB detections.")
        self.det_thresh = det_thresh
        self.max_age = max_age
        self.max_obs = max_obs
        self.min_hits = min_hits
        self.iou_threshold = iou_threshold
        self.per_class = per_class
        self.nr_classes = nr_classes
        self._asso_func_base_name = asso_func      # resolved on the first frame, like the reference (basetracker.py:175-180)
        self.is_obb = bool(is_obb)
        self.asso_func_name = f"{asso_func}_obb" if self.is_obb else asso_func     # detection_layout.py:25-26
        self.frame_count = 0
        self.last_emb_size = None
        self._first_frame_processed = False
        self._first_dets_processed = False
        self._masks_warning_issued = False
        if self.max_age >= self.max_obs:   # basetracker.py:93-97
            LOGGER.warning("Max age > max observations, increasing size of max observations...")
            self.max_obs = self.max_age + 5
        name = kwargs.pop("_tracker_name", None)
        if name:
            LOGGER.info("%s: %s", name, ", ".join(f"{k}={v}" for k, v in kwargs.items() if not k.startswith("_")))

    # ------------------------------------------------------------------ public
    def update(self, dets: np.ndarray, img: np.ndarray, embs: np.ndarray = None, masks: np.ndarray = None) -> TrackResults:
        dets, img = self._preprocess(dets, img)
        masks = self._preprocess_masks(dets, masks)
        return TrackResults(self._do_update(dets, img, embs, masks))

    def reset(self) -> None:
        self.frame_count = 0

    # ---------------------------------------------------------------- pipeline
    def _preprocess(self, dets, img):
        if hasattr(dets, "data"):                 # ultralytics-style wrappers; an ndarray yields its buffer
            dets = dets.data
        if isinstance(dets, memoryview):          # ... which lands here: a float32 copy (basetracker.py:155-161)
            dets = np.array(dets, dtype=np.float32)
        if (not self._first_dets_processed and isinstance(dets, np.ndarray) and dets.ndim == 2
                and dets.shape[1] in (AABB_COLS, OBB_COLS)):
            oriented = dets.shape[1] == OBB_COLS
            if oriented and not self.supports_obb:
                raise AssertionError(
                    f"{type(self).__name__} does not support OBB detections. "
                    "Use an OBB-capable tracker such as ByteTrack, BotSort, OCSort, or SFSORT."
                )
            self._set_detection_mode(oriented)
            self._first_dets_processed = True
        if not self._first_frame_processed and img is not None:
            self.h, self.w = img.shape[0:2]
            if self.asso_func_name not in ASSO_NAMES:      # AssociationFunction._get_asso_func (iou.py:419-422)
                raise ValueError(f"Invalid association mode: {self.asso_func_name}. Choose from {list(ASSO_NAMES)}")
            self._first_frame_processed = True
        return dets, img

    def _preprocess_masks(self, dets, masks):
        if masks is None:
            return None
        if not self._masks_warning_issued:
            LOGGER.warning("%s does not support masks. Masks will be ignored.", type(self).__name__)
            self._masks_warning_issued = True
        return None

    def _set_detection_mode(self, is_obb: bool) -> None:        # basetracker.py:337-348
        self.is_obb = bool(is_obb)
        self.asso_func_name = f"{self._asso_func_base_name}_obb" if self.is_obb else self._asso_func_base_name

    # the layout's numbers (detection_layout.py:61-84)
    det_cols = property(lambda self: OBB_COLS if self.is_obb else AABB_COLS)
    box_cols = property(lambda self: 5 if self.is_obb else 4)
    conf_idx = property(lambda self: 5 if self.is_obb else CONF_IDX)
    cls_idx = property(lambda self: 6 if self.is_obb else CLS_IDX)
    output_cols = property(lambda self: OBB_OUT_COLS if self.is_obb else OUT_COLS)

    def empty_detections(self, dtype=np.float32):
        return np.empty((0, self.det_cols), dtype=dtype)

    def empty_output(self, dtype=float):
        return np.empty((0, self.output_cols), dtype=dtype)

    def _do_update(self, dets, img, embs=None, masks=None):
        if dets is None or len(dets) == 0:
            dets = self.empty_detections()
        if not self.per_class:
            return self._update_impl(dets=dets, img=img, embs=embs, masks=None, class_list=0)
        rows = []
        frame_count = self.frame_count
        for cls_id in range(self.nr_classes):
            class_dets, class_embs = self.get_class_dets_n_embs(dets, embs, cls_id)
            self.frame_count = frame_count            # every class sees the same frame number
            tracks = self._update_impl(dets=class_dets, img=img, embs=class_embs, masks=None, class_list=cls_id)
            if tracks.size > 0:
                rows.append(tracks)
        self.frame_count = frame_count + 1
        return np.vstack(rows) if rows else self.empty_output()

    def get_class_dets_n_embs(self, dets, embs, cls_id):
        class_dets = self.empty_detections()
        class_embs = np.empty((0, self.last_emb_size)) if self.last_emb_size is not None else None
        if dets.size == 0:
            return class_dets, class_embs
        idx = np.where(dets[:, self.cls_idx] == cls_id)[0]
        class_dets = dets[idx]
        if embs is None:
            return class_dets, class_embs
        assert dets.shape[0] == embs.shape[0], (
            "Detections and embeddings must have the same number of elements when both are provided"
        )
        class_embs = None
        if embs.size > 0:
            class_embs = embs[idx]
            self.last_emb_size = class_embs.shape[1]
        return class_dets, class_embs

    def check_inputs(self, dets, img, embs=None):
        assert isinstance(dets, np.ndarray), f"Unsupported 'dets' input format '{type(dets)}', valid format is np.ndarray"
        assert isinstance(img, np.ndarray), f"Unsupported 'img_numpy' input format '{type(img)}', valid format is np.ndarray"
        assert len(dets.shape) == 2, "Unsupported 'dets' dimensions, valid number of dimensions is two"
        if embs is not None:
            assert dets.shape[0] == embs.shape[0], "Missmatch between detections and embeddings sizes"
        assert dets.shape[1] == self.det_cols, (
            f"Unsupported 'dets' 2nd dimension length, valid length is {self.det_cols} "
            + ("(cx,cy,w,h,angle,conf,cls)" if self.is_obb else "(x1,y1,x2,y2,conf,cls)")
        )

    def _update_impl(self, dets, img, embs=None, masks=None, class_list: int = 0) -> np.ndarray:
        raise NotImplementedError("The _update_impl method needs to be implemented by the subclass.")
