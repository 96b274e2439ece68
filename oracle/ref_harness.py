"""Import the REAL reference (/root/reference) under documented stand-ins.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  The reference SOURCE tree exists in the build
container only: ``tests/golden/make_golden.py`` uses it to produce the committed
fixtures; ``tests/test_oracle_vs_reference.py`` uses it (skipped when the reference
is absent) to pin the oracle bit-for-bit.  On the GPU box there is no /root/reference;
what travels there is ``oracle/_ref/`` -- the same modules byte-compiled from
/root/reference by ``oracle/make_ref.py`` (sourceless ``.pyc`` build outputs, git-ignored
like ``liboracle.so``; no source text is copied) -- and ``bench.py``'s ``cpu_baseline``
leg imports the reference classes from there so that the reference itself is timed on
the GPU box's host cores (``cpu_baseline.kind = "reference"``).  ``reference_available()``
still means "the source tree is mounted" (the pinning tests read other reference files
by path); ``reference_runnable()`` is true for either form.

Stand-ins injected through ``sys.modules``:
  * ``lap``  -> ``oracle.lap`` (lapx 0.9.4 is not installed; PARITY UNPINNED);
  * ``cv2``  -> constants touched at import time + ``resize`` / ``cvtColor``
               from ``oracle.crops`` (opencv is not installed; PARITY UNPINNED);
  * ``gdown``, ``filterpy``, ``yacs``, ``ftfy`` -> empty modules.
The reference tracker/Kalman/association classes themselves run unmodified.
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

SOURCE_ROOT = Path("/root/reference")
COMPILED_ROOT = Path(__file__).resolve().parent / "_ref"


def _pick_root() -> Path:
    """The source tree when it is mounted; otherwise the byte-compiled copy (oracle/make_ref.py).  BOXMOT_ORACLE_REF=compiled forces
    the compiled copy (the test of the fallback in the build container)."""
    import os

    have_src = (SOURCE_ROOT / "boxmot" / "trackers" / "bbox" / "botsort" / "botsort.py").exists()
    if have_src and os.environ.get("BOXMOT_ORACLE_REF", "") != "compiled":
        return SOURCE_ROOT
    if _compiled_usable():
        return COMPILED_ROOT
    return SOURCE_ROOT


def _compiled_usable() -> bool:
    """oracle/_ref/ exists AND was byte-compiled for THIS interpreter: a .pyc carries the bytecode magic of the Python that wrote it,
    and a different minor version refuses every import with "bad magic number" -- then the copy is as good as absent and the callers
    fall back to the port (bench.py: cpu_baseline.kind = "port")."""
    import importlib.util
    import json

    man = COMPILED_ROOT / "MANIFEST.json"
    if not man.exists():
        return False
    try:
        return json.loads(man.read_text()).get("magic") == importlib.util.MAGIC_NUMBER.hex()
    except (OSError, ValueError):
        return False


REFERENCE_ROOT = _pick_root()


def reference_kind():
    """"source" (/root/reference mounted), "compiled" (oracle/_ref/) or None."""
    if REFERENCE_ROOT == SOURCE_ROOT:
        return "source" if (SOURCE_ROOT / "boxmot" / "trackers" / "bbox" / "botsort" / "botsort.py").exists() else None
    return "compiled" if _compiled_usable() else None


def reference_available() -> bool:
    """The reference SOURCE tree is mounted (build container)."""
    return reference_kind() == "source"


def reference_runnable() -> bool:
    """The reference classes can be imported: from the source tree or from the byte-compiled copy."""
    return reference_kind() is not None


def _load_by_path(rel: str, name: str):
    """Load one reference module by its path relative to the reference root (source ``.py`` or compiled ``.pyc``)."""
    import importlib.machinery

    if reference_kind() == "compiled":
        path = (REFERENCE_ROOT / rel).with_suffix(".pyc")
        loader = importlib.machinery.SourcelessFileLoader(name, str(path))
        spec = importlib.util.spec_from_loader(name, loader)
    else:
        spec = importlib.util.spec_from_file_location(name, REFERENCE_ROOT / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _cv2_stub():
    from oracle import crops

    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.COLOR_BGR2RGB = 4
    cv2.COLOR_BGR2GRAY = 6
    cv2.MOTION_TRANSLATION = 0
    cv2.MOTION_EUCLIDEAN = 1
    cv2.MOTION_AFFINE = 2
    cv2.MOTION_HOMOGRAPHY = 3
    cv2.TERM_CRITERIA_EPS = 2
    cv2.TERM_CRITERIA_COUNT = 1
    cv2.BORDER_CONSTANT = 0

    def resize(src, dsize, interpolation=1, **_):
        assert interpolation == cv2.INTER_LINEAR
        return crops.cv2_resize_linear_u8(src, dsize)

    def cvtColor(src, code):
        assert code == cv2.COLOR_BGR2RGB
        return np.ascontiguousarray(src[:, :, ::-1])

    def copyMakeBorder(src, top, bottom, left, right, borderType, value=0):
        assert borderType == cv2.BORDER_CONSTANT
        h, w = src.shape[:2]
        out = np.empty((h + top + bottom, w + left + right, src.shape[2]), dtype=src.dtype)
        out[:] = np.asarray(value, dtype=src.dtype)
        out[top:top + h, left:left + w] = src
        return out

    # oriented boxes (iou.py:5-115, bytetrack.py:147-189): the intersection polygon is represented by its area alone -- what the
    # reference does with it is cv2.contourArea -- and that area is oracle/obb.py's fp64 clipping, NOT OpenCV's fp32 edge enumeration
    # (parity unpinned for this one quantity, see oracle/obb.py)
    class _Polygon:
        def __init__(self, area):
            self.area = area

    def rotatedRectangleIntersection(r1, r2):
        from oracle import obb
        area = obb.rotated_intersection_area(r1, r2)
        return (1 if area > 0.0 else 0), (_Polygon(area) if area > 0.0 else None)

    def contourArea(poly):
        return poly.area

    def boxPoints(rect):
        from oracle import obb
        return obb.box_points(rect[0][0], rect[0][1], rect[1][0], rect[1][1], rect[2])

    # camera-motion compensation of oriented tracks (botsort_track.py:134-195): oracle/obb.py's restatements, unpinned
    def transform(pts, m):
        from oracle import obb
        return obb.transform_points(pts, m).reshape(-1, 1, 2)

    def minAreaRect(pts):
        from oracle import obb
        return obb.min_area_rect(pts)

    cv2.transform = transform
    cv2.minAreaRect = minAreaRect
    cv2.resize = resize
    cv2.cvtColor = cvtColor
    cv2.copyMakeBorder = copyMakeBorder
    cv2.rotatedRectangleIntersection = rotatedRectangleIntersection
    cv2.contourArea = contourArea
    cv2.boxPoints = boxPoints
    return cv2


def install_standins():
    if "lap" not in sys.modules or not hasattr(sys.modules["lap"], "_oracle_standin"):
        from oracle import lap as oracle_lap

        lap = types.ModuleType("lap")
        lap.lapjv = oracle_lap.lapjv
        lap._oracle_standin = True
        sys.modules["lap"] = lap
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = _cv2_stub()
    for name in ("gdown", "filterpy", "yacs", "ftfy"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if str(REFERENCE_ROOT) not in sys.path:
        sys.path.insert(0, str(REFERENCE_ROOT))


def load_sof():
    """Return the reference ``SOF`` class (boxmot/motion/cmc/sof.py) with ``cv2``'s numerics answered by oracle/sof.py's
    restatements (``cvtColor(BGR2GRAY)``, ``resize(fx, fy)``, ``goodFeaturesToTrack``, ``cornerSubPix``, ``calcOpticalFlowPyrLK``,
    ``estimateAffinePartial2D``): what runs is the REFERENCE's control flow -- initialisation, re-detection, the inlier test,
    the mask, the scaling of the translation -- which is what ``SofOracle`` is pinned against (tests/test_sof.py)."""
    install_standins()
    from oracle import crops, ecc, sof
    cv2 = sys.modules["cv2"]
    cv2.RANSAC = 8
    base_cvt, base_resize = cv2.cvtColor, cv2.resize

    def cvtColor(src, code):
        return ecc.bgr2gray_u8(src) if code == cv2.COLOR_BGR2GRAY else base_cvt(src, code)

    def resize(src, dsize, fx=0.0, fy=0.0, interpolation=1, **kw):
        if tuple(dsize) == (0, 0):
            assert interpolation == cv2.INTER_LINEAR and src.ndim == 2
            h, w = src.shape
            return crops.cv2_resize_linear_u8(src, (int(np.rint(w * fx)), int(np.rint(h * fy))), (1.0 / fx, 1.0 / fy))
        return base_resize(src, dsize, interpolation=interpolation, **kw)

    def goodFeaturesToTrack(img, mask=None, maxCorners=1000, qualityLevel=0.01, minDistance=1, blockSize=3, useHarrisDetector=False, k=0.04):
        assert (maxCorners, qualityLevel, minDistance, blockSize, useHarrisDetector) == (1000, 0.01, 1, 3, False)
        pts = sof.good_features(img, mask)
        return None if pts is None else pts.reshape(-1, 1, 2)

    def cornerSubPix(img, corners, winSize, zeroZone, criteria):
        assert tuple(winSize) == (5, 5) and tuple(zeroZone) == (-1, -1) and tuple(criteria[1:]) == (30, 0.01)
        corners[:] = sof.corner_subpix(img, corners.reshape(-1, 2)).reshape(corners.shape)      # in place, like OpenCV
        return corners

    def calcOpticalFlowPyrLK(prev, nxt, pts, _next, winSize=(21, 21), maxLevel=3, criteria=None):
        assert tuple(winSize) == (21, 21) and maxLevel == 3 and tuple(criteria[1:]) == (30, 0.01)
        p = np.asarray(pts, np.float32).reshape(-1, 2)
        out, status = sof.lk_track(sof.build_pyramid(prev, maxLevel), sof.build_pyramid(nxt, maxLevel), p)
        return out.reshape(-1, 1, 2), status.reshape(-1, 1), np.zeros((len(p), 1), np.float32)

    def estimateAffinePartial2D(frm, to, method=None, ransacReprojThreshold=3.0):
        assert method == cv2.RANSAC
        H, inl = sof.estimate_affine_partial_2d(np.asarray(frm, np.float32).reshape(-1, 2), np.asarray(to, np.float32).reshape(-1, 2),
                                                ransacReprojThreshold)
        return H, inl.reshape(-1, 1)

    cv2.cvtColor, cv2.resize = cvtColor, resize
    cv2.goodFeaturesToTrack, cv2.cornerSubPix = goodFeaturesToTrack, cornerSubPix
    cv2.calcOpticalFlowPyrLK, cv2.estimateAffinePartial2D = calcOpticalFlowPyrLK, estimateAffinePartial2D
    from boxmot.motion.cmc.sof import SOF

    return SOF


def load_botsort():
    """Return the reference BotSort class (imported from /root/reference)."""
    install_standins()
    from boxmot.trackers.bbox.botsort.botsort import BotSort

    return BotSort


def load_deepocsort():
    """Return the reference DeepOcSort class (imported from /root/reference)."""
    install_standins()
    from boxmot.trackers.bbox.deepocsort.deepocsort import DeepOcSort

    return DeepOcSort


def load_ocsort():
    """Return the reference OcSort class (imported from /root/reference)."""
    install_standins()
    from boxmot.trackers.bbox.ocsort.ocsort import OcSort

    return OcSort


def load_bytetrack():
    """Return the reference ByteTrack class with its process-global id counter rewound (bytetrack/basetrack.py:16)."""
    install_standins()
    from boxmot.trackers.bbox.bytetrack.basetrack import BaseTrack
    from boxmot.trackers.bbox.bytetrack.bytetrack import ByteTrack

    BaseTrack._count = 0
    return ByteTrack


def load_strongsort():
    """Return the reference StrongSort class.  Tracks start Confirmed when GITHUB_ACTIONS == "true"
    (sort/track.py:91-98), so that variable is cleared first."""
    import os

    os.environ.pop("GITHUB_ACTIONS", None)
    install_standins()
    from boxmot.trackers.bbox.strongsort.strongsort import StrongSort

    return StrongSort


def load_engine_callers():
    """The reference's two per-frame tracker call sites, imported unmodified: ``TrackerRuntime``
    (boxmot/engine/tracking/runtime.py:15-128) and ``Results`` (its ``_run_tracker``, engine/tracking/results.py:467-496).
    Their import chain reaches ``boxmot.reid.core`` -> the CLIP backbone's ``torchvision.transforms`` names (clip.py:11), which are
    only referenced inside functions the callers never run: an empty stand-in module with those names is injected."""
    install_standins()
    if "torchvision" not in sys.modules:
        tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
        for name in ("CenterCrop", "Compose", "Normalize", "Resize", "ToTensor", "InterpolationMode"):
            setattr(tvt, name, type(name, (), {"BICUBIC": 3}))
        tv.transforms = tvt
        sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    from boxmot.engine.tracking.results import Results
    from boxmot.engine.tracking.runtime import TrackerRuntime

    return TrackerRuntime, Results


class IdentityCMC:
    """Stands where StrongSort's unconditional ECC object stands (strongsort.py:67): no camera motion."""

    def apply(self, img, dets):
        return np.eye(2, 3)


def load_osnet_module():
    """Load boxmot/reid/backbones/osnet.py by path (``boxmot.reid`` itself cannot import)."""
    install_standins()
    return _load_by_path("boxmot/reid/backbones/osnet.py", "_ref_osnet")


class RefReID:
    """Reference ``BaseModelBackend.get_crops/get_features`` bound to a reference OSNet.

    ``boxmot.reid.backends.base_backend`` cannot be imported offline (it pulls the
    whole ReID registry incl. CLIP/torchvision), so its two methods are executed
    from source text against this minimal object: the crop loop and the feature
    normalisation are the reference's own statements.
    """

    def __init__(self, model, input_shape=(256, 128), preprocess=None):
        import torch

        install_standins()
        import cv2

        if reference_kind() == "compiled":
            import marshal

            code = marshal.loads((REFERENCE_ROOT / "base_backend_subset.marshal").read_bytes())
        else:
            from oracle.make_ref import BACKEND, backend_subset_code

            code = backend_subset_code((REFERENCE_ROOT / BACKEND).read_text())
        ns = {"np": np, "torch": torch, "cv2": cv2}
        exec(code, ns)
        Backend = ns["BaseModelBackend"]
        pp = _load_by_path("boxmot/reid/core/preprocessing.py", "_ref_reid_preprocessing")
        get_preprocess_fn = pp.get_preprocess_fn

        class _Bound(Backend):
            def forward(self_inner, im_batch):
                return model(im_batch)

        b = _Bound.__new__(_Bound)
        b.device = torch.device("cpu")
        b.half = False
        b.nhwc = False
        b.input_shape = input_shape
        b.preprocess_fn = get_preprocess_fn(preprocess)
        b.mean_array = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        b.std_array = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        self._b = b
        self.model = model

    def get_features(self, xyxys, img):
        return self._b.get_features(xyxys, img)

    def get_crops(self, xyxys, img):
        return self._b.get_crops(xyxys, img)
