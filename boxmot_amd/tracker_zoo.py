"""Factory slot: ``create_tracker(..., tracker_backend="hip")``.

Same call shape as the reference factory (boxmot/trackers/tracker_zoo.py:33-147):
the tracker's YAML defaults ``{param: {default: ...}}`` are flattened to keyword
arguments, ``reid_weights`` / ``reid_model`` select the ReID backend, and the
backend name picks the implementation -- here the closed set is {"hip"}
(the reference's is {"python", "cpp"}, boxmot/trackers/specs.py:6).
"""
from __future__ import annotations

from pathlib import Path

import yaml

from boxmot_amd.botsort import BotSort

# boxmot/configs/trackers/botsort.yaml defaults (what create_tracker applies when no config is given)
BOTSORT_YAML_DEFAULTS = dict(
    track_high_thresh=0.6296854875023994, track_low_thresh=0.1014392537025336,
    new_track_thresh=0.6246494191492591, track_buffer=40, match_thresh=0.7722224024589055,
    use_cmc=True, cmc_method="sof", frame_rate=30, fuse_first_associate=True, with_reid=True,
    proximity_thresh=0.6084297894561342, appearance_thresh=0.6188818853936099,
    unconfirmed_emb_scale=2.5445206391993294, second_match_thresh=0.28795081514328974,
    unconfirmed_match_thresh=0.41148010638233784, removed_stracks_buffer=329,
)
# boxmot/configs/trackers/deepocsort.yaml defaults.  The YAML key `iou_thresh` is not a constructor argument
# (`iou_threshold` is): the reference swallows it in BaseTracker(**kwargs) and the 0.3 default applies
#  -- reproduced by passing it through unchanged.
DEEPOCSORT_YAML_DEFAULTS = dict(
    det_thresh=0.5, max_age=30, min_hits=3, iou_thresh=0.3, delta_t=3, asso_func="iou", inertia=0.2,
    w_association_emb=0.75, alpha_fixed_emb=0.95, aw_param=0.5, embedding_off=False, cmc_off=False, aw_off=False,
    Q_xy_scaling=0.01, Q_s_scaling=0.0001,
)
# boxmot/configs/trackers/strongsort.yaml defaults
STRONGSORT_YAML_DEFAULTS = dict(min_conf=0.6, max_cos_dist=0.4, max_iou_dist=0.7, max_age=30, n_init=3, nn_budget=100,
                                mc_lambda=0.98, ema_alpha=0.9)
# boxmot/configs/trackers/ocsort.yaml defaults
OCSORT_YAML_DEFAULTS = dict(min_conf=0.1, det_thresh=0.6, max_age=30, min_hits=3, delta_t=3, asso_func="iou", use_byte=False,
                            inertia=0.1, Q_xy_scaling=0.01, Q_s_scaling=0.0001)
# boxmot/configs/trackers/bytetrack.yaml defaults
BYTETRACK_YAML_DEFAULTS = dict(min_conf=0.1, track_thresh=0.6, match_thresh=0.9, track_buffer=30, frame_rate=30)
SUPPORTED = ("botsort", "bytetrack", "deepocsort", "ocsort", "strongsort")
# the names of the reference's TRACKER_MAPPING (tracker_zoo.py:14-25): the ones not in SUPPORTED are known but not built here
REFERENCE_TRACKERS = SUPPORTED + ("sfsort", "hybridsort", "boosttrack", "occluboost", "sam2mot")


def flatten_yaml_config(cfg: dict) -> dict:
    """{param: {default, activates: {...}}} -> {param: default} (tracker_zoo.py:112-119)."""
    out = {}
    for key, spec in cfg.items():
        if isinstance(spec, dict) and "default" in spec:
            out[key] = spec["default"]
            for sub, sub_spec in (spec.get("activates") or {}).items():
                out[sub] = sub_spec["default"] if isinstance(sub_spec, dict) else sub_spec
        else:
            out[key] = spec
    return out


def create_tracker(tracker_type: str = "botsort", tracker_config=None, reid_weights=None, device=None, half=None,
                   per_class: bool = False, evolve_param_dict: dict | None = None, reid_preprocess=None,
                   reid_model=None, tracker_backend: str = "hip", **overrides):
    """``overrides`` are constructor arguments laid over the YAML defaults.  Camera-motion compensation runs on the device with the
    estimator the reference's defaults name: BoT-SORT ``cmc_method="sof"`` (botsort.yaml; ``"ecc"`` selects boxmot_amd.cmc.HipECC),
    DeepOCSORT the built-in sparse-optical-flow estimator (boxmot_amd.cmc.HipSOF; ``cmc_off=True`` opts out), StrongSORT ECC
    (``cmc=None`` opts out).  ``cmc=<object with apply(img, dets) -> 2x3 warp>`` overrides any of them."""
    if tracker_backend != "hip":
        raise ValueError(f"tracker_backend={tracker_backend!r}: boxmot_amd provides the 'hip' backend only")
    if tracker_type not in SUPPORTED:
        if tracker_type not in REFERENCE_TRACKERS:      # tracker_zoo.py:103-105: a name the reference does not know either
            raise ValueError(f"Unknown tracker type: '{tracker_type}'. Available trackers are: {', '.join(SUPPORTED)}")
        raise NotImplementedError(f"tracker {tracker_type!r} is not implemented on the HIP backend (have: {SUPPORTED})")
    if reid_preprocess not in (None, "resize", "resize_pad"):
        raise ValueError(f"Unknown preprocess '{reid_preprocess}'. Available: ['resize', 'resize_pad']")   # preprocessing.py:60-64
    if evolve_param_dict is not None:
        kwargs = dict(evolve_param_dict)
    elif tracker_config is None:
        kwargs = dict({"botsort": BOTSORT_YAML_DEFAULTS, "deepocsort": DEEPOCSORT_YAML_DEFAULTS,
                       "ocsort": OCSORT_YAML_DEFAULTS, "bytetrack": BYTETRACK_YAML_DEFAULTS, "strongsort": STRONGSORT_YAML_DEFAULTS}[tracker_type])
    elif isinstance(tracker_config, dict):
        kwargs = flatten_yaml_config(tracker_config)
    else:
        kwargs = flatten_yaml_config(yaml.safe_load(Path(tracker_config).read_text()))
    kwargs.update(overrides)
    kwargs["per_class"] = per_class
    if tracker_type == "strongsort":
        from boxmot_amd.strongsort import StrongSort

        if reid_model is None and reid_weights is not None:
            from boxmot_amd.reid import HipReID

            reid_model = HipReID(reid_weights, preprocess=reid_preprocess)
        kwargs.setdefault("cmc", "ecc")       # strongsort.py:67: the reference's StrongSort always runs ECC; cmc=None opts out (identity)
        return StrongSort(reid_model=reid_model, **kwargs)
    if tracker_type == "bytetrack":
        from boxmot_amd.bytetrack import ByteTrack

        return ByteTrack(**kwargs)
    if tracker_type == "ocsort":
        from boxmot_amd.deepocsort import OcSort

        return OcSort(**kwargs)
    if tracker_type == "deepocsort":
        from boxmot_amd.deepocsort import DeepOcSort

        if not kwargs.get("embedding_off", False) and reid_model is None and reid_weights is not None:
            from boxmot_amd.reid import HipReID

            reid_model = HipReID(reid_weights, preprocess=reid_preprocess)
        return DeepOcSort(reid_model=reid_model, **kwargs)
    if kwargs.get("with_reid", True) and reid_model is None and reid_weights is not None:
        from boxmot_amd.reid import HipReID

        reid_model = HipReID(reid_weights, preprocess=reid_preprocess)
    return BotSort(reid_model=reid_model, **kwargs)
