"""ctypes binding of ``libboxmot_hip.so`` (include/boxmot_hip.h).

Mirrors the reference's Python<->C shim for native trackers
(boxmot/native/trackers/botsort.py:94-146 for the config struct and signatures,
boxmot/native/trackers/_common.py:158-221 for the call convention).  The
library is built in-tree by ``__graft_entry__.build()``; importing it when it is
missing, or calling into it on a machine without a HIP device, fails loudly --
there is no Python/CPU fallback for the tracker math.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libboxmot_hip.so"

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_double_p = ctypes.POINTER(ctypes.c_double)


class BotSortConfig(ctypes.Structure):
    """``BoxMOTHipBotSortConfig`` (include/boxmot_hip.h)."""

    _fields_ = [
        ("track_high_thresh", ctypes.c_double),
        ("track_low_thresh", ctypes.c_double),
        ("new_track_thresh", ctypes.c_double),
        ("track_buffer", ctypes.c_int),
        ("match_thresh", ctypes.c_double),
        ("proximity_thresh", ctypes.c_double),
        ("appearance_thresh", ctypes.c_double),
        ("cmc_method", ctypes.c_char_p),
        ("frame_rate", ctypes.c_int),
        ("fuse_first_associate", ctypes.c_int),
        ("with_reid", ctypes.c_int),
        ("max_obs", ctypes.c_int),
        ("reid_model_path", ctypes.c_char_p),
        ("reid_preprocess", ctypes.c_char_p),
        ("second_match_thresh", ctypes.c_double),
        ("unconfirmed_match_thresh", ctypes.c_double),
        ("unconfirmed_emb_scale", ctypes.c_double),
        ("removed_stracks_buffer", ctypes.c_int),
        ("n_streams", ctypes.c_int),
        ("max_tracks", ctypes.c_int),
        ("max_dets", ctypes.c_int),
        ("emb_dim", ctypes.c_int),
        ("n_class_lists", ctypes.c_int),
        ("tracker_kind", ctypes.c_int),
        ("is_obb", ctypes.c_int),
    ]


class DeepOcSortConfig(ctypes.Structure):
    """``BoxMOTHipDeepOcSortConfig`` (include/boxmot_hip.h)."""

    _fields_ = [
        ("det_thresh", ctypes.c_double),
        ("max_age", ctypes.c_int),
        ("max_obs", ctypes.c_int),
        ("min_hits", ctypes.c_int),
        ("iou_threshold", ctypes.c_double),
        ("delta_t", ctypes.c_int),
        ("inertia", ctypes.c_double),
        ("w_association_emb", ctypes.c_double),
        ("alpha_fixed_emb", ctypes.c_double),
        ("aw_param", ctypes.c_double),
        ("embedding_off", ctypes.c_int),
        ("cmc_off", ctypes.c_int),
        ("aw_off", ctypes.c_int),
        ("Q_xy_scaling", ctypes.c_double),
        ("Q_s_scaling", ctypes.c_double),
        ("reid_model_path", ctypes.c_char_p),
        ("n_streams", ctypes.c_int),
        ("max_tracks", ctypes.c_int),
        ("max_dets", ctypes.c_int),
        ("emb_dim", ctypes.c_int),
        ("use_byte", ctypes.c_int),
        ("min_conf", ctypes.c_double),
        ("asso_func", ctypes.c_int),
        ("frame_w", ctypes.c_int),
        ("frame_h", ctypes.c_int),
        ("is_obb", ctypes.c_int),
    ]


# BOXMOT_HIP_ASSO_* (include/boxmot_hip.h): the axis-aligned entries of AssociationFunction._get_asso_func (iou.py:408-417)
ASSO_FUNCS = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3, "hmiou": 4, "centroid": 5}


class StrongSortConfig(ctypes.Structure):
    """``BoxMOTHipStrongSortConfig`` (include/boxmot_hip.h)."""

    _fields_ = [
        ("max_age", ctypes.c_int),
        ("min_conf", ctypes.c_double),
        ("max_cos_dist", ctypes.c_double),
        ("max_iou_dist", ctypes.c_double),
        ("n_init", ctypes.c_int),
        ("nn_budget", ctypes.c_int),
        ("mc_lambda", ctypes.c_double),
        ("ema_alpha", ctypes.c_double),
        ("reid_model_path", ctypes.c_char_p),
        ("n_streams", ctypes.c_int),
        ("max_tracks", ctypes.c_int),
        ("max_dets", ctypes.c_int),
        ("emb_dim", ctypes.c_int),
    ]


# every symbol include/boxmot_hip.h declares: (name, restype, argtypes)
_VP = ctypes.c_void_p
_I = ctypes.c_int
SIGNATURES = {
    "boxmot_hip_botsort_default_config": (None, [ctypes.POINTER(BotSortConfig)]),
    "boxmot_hip_bytetrack_default_config": (None, [ctypes.POINTER(BotSortConfig)]),
    "boxmot_hip_botsort_create": (_VP, [ctypes.POINTER(BotSortConfig)]),
    "boxmot_hip_botsort_destroy": (None, [_VP]),
    "boxmot_hip_botsort_reset": (_I, [_VP]),
    "boxmot_hip_botsort_reserve": (_I, [_VP, _I, _I]),
    "boxmot_hip_botsort_capacity": (_I, [_VP, c_int_p, c_int_p, c_int_p]),
    "boxmot_hip_botsort_update": (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _I, c_int_p, c_int_p]),
    "boxmot_hip_botsort_update_stream": (_I, [_VP, _I, _I, _I, _VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _I,
                                              c_int_p, c_int_p]),
    "boxmot_hip_botsort_update_batch": (_I, [_VP, _I, _VP, _VP, _VP, _I, _VP, _I, _I, _I, _VP, _I, _VP]),
    "boxmot_hip_botsort_step_device": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "boxmot_hip_botsort_set_warp": (_I, [_VP, _I, _VP]),
    "boxmot_hip_botsort_synchronize": (_I, [_VP]),
    "boxmot_hip_botsort_stream": (_VP, [_VP]),
    "boxmot_hip_botsort_timer_start": (_I, [_VP]),
    "boxmot_hip_botsort_timer_stop_ms": (_I, [_VP, c_double_p]),
    "boxmot_hip_botsort_reid_kernel_ms": (_I, [_VP, c_double_p, c_int_p]),
    "boxmot_hip_botsort_phase_clocks": (_I, [_VP, _VP]),
    "boxmot_hip_botsort_status": (_I, [_VP, _VP, _I]),
    "boxmot_hip_botsort_set_reid_blob": (_I, [_VP, _VP, ctypes.c_long]),
    "boxmot_hip_botsort_set_reid_mode": (_I, [_VP, _I]),
    "boxmot_hip_botsort_last_reid_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_hip_botsort_last_reid_preprocess_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_hip_botsort_last_reid_process_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_hip_botsort_last_reid_postprocess_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_hip_botsort_last_track_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_hip_botsort_state_dump": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, c_int_p, c_int_p, c_int_p]),
    "boxmot_hip_botsort_debug_costs_enable": (_I, [_VP, _I]),
    "boxmot_hip_botsort_debug_costs": (_I, [_VP, _I, _I, _I, _VP, ctypes.c_long, c_int_p, c_int_p]),
    "boxmot_hip_deepocsort_default_config": (None, [ctypes.POINTER(DeepOcSortConfig)]),
    "boxmot_hip_deepocsort_create": (_VP, [ctypes.POINTER(DeepOcSortConfig)]),
    "boxmot_hip_deepocsort_destroy": (None, [_VP]),
    "boxmot_hip_deepocsort_reset": (_I, [_VP]),
    "boxmot_hip_deepocsort_reserve": (_I, [_VP, _I, _I]),
    "boxmot_hip_deepocsort_capacity": (_I, [_VP, c_int_p, c_int_p, c_int_p]),
    "boxmot_hip_deepocsort_set_warp": (_I, [_VP, _I, _VP]),
    "boxmot_hip_deepocsort_update": (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _I, c_int_p, c_int_p]),
    "boxmot_hip_deepocsort_update_stream": (_I, [_VP, _I, _I, c_int_p, _VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _I,
                                                 c_int_p, c_int_p]),
    "boxmot_hip_deepocsort_update_batch": (_I, [_VP, _I, _VP, _VP, _VP, _I, _VP, _I, _I, _I, _VP, _I, _VP]),
    "boxmot_hip_deepocsort_step_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "boxmot_hip_deepocsort_step_device_frames": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "boxmot_hip_deepocsort_reid_kernel_ms": (_I, [_VP, ctypes.POINTER(ctypes.c_double), c_int_p]),
    "boxmot_hip_deepocsort_set_reid_mode": (_I, [_VP, _I]),
    "boxmot_hip_deepocsort_synchronize": (_I, [_VP]),
    "boxmot_hip_deepocsort_set_crop_bound": (_I, [_VP, _I]),
    "boxmot_hip_deepocsort_state_dump": (_I, [_VP, _I, _VP, _VP, _VP, c_int_p, c_int_p, c_int_p]),
    "boxmot_hip_deepocsort_debug_costs_enable": (_I, [_VP, _I]),
    "boxmot_hip_deepocsort_debug_costs": (_I, [_VP, _I, _I, _VP, ctypes.c_long, c_int_p, c_int_p, c_int_p]),
    "boxmot_hip_strongsort_default_config": (None, [ctypes.POINTER(StrongSortConfig)]),
    "boxmot_hip_strongsort_create": (_VP, [ctypes.POINTER(StrongSortConfig)]),
    "boxmot_hip_strongsort_destroy": (None, [_VP]),
    "boxmot_hip_strongsort_reset": (_I, [_VP]),
    "boxmot_hip_strongsort_reserve": (_I, [_VP, _I, _I]),
    "boxmot_hip_strongsort_capacity": (_I, [_VP, c_int_p, c_int_p, c_int_p]),
    "boxmot_hip_strongsort_set_warp": (_I, [_VP, _I, _VP]),
    "boxmot_hip_strongsort_update": (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _I, c_int_p, c_int_p]),
    "boxmot_hip_strongsort_update_batch": (_I, [_VP, _I, _VP, _VP, _VP, _I, _VP, _I, _I, _I, _VP, _I, _VP]),
    "boxmot_hip_strongsort_step_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "boxmot_hip_strongsort_step_device_frames": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "boxmot_hip_strongsort_reid_kernel_ms": (_I, [_VP, ctypes.POINTER(ctypes.c_double), c_int_p]),
    "boxmot_hip_strongsort_set_reid_mode": (_I, [_VP, _I]),
    "boxmot_hip_strongsort_synchronize": (_I, [_VP]),
    "boxmot_hip_strongsort_set_crop_bound": (_I, [_VP, _I]),
    "boxmot_hip_strongsort_track_count": (_I, [_VP, _I, c_int_p]),
    "boxmot_hip_strongsort_state_dump": (_I, [_VP, _I, _VP, _VP, _VP, c_int_p, c_int_p, c_int_p]),
    "boxmot_hip_strongsort_debug_costs_enable": (_I, [_VP, _I]),
    "boxmot_hip_strongsort_debug_costs": (_I, [_VP, _I, _I, _I, _VP, ctypes.c_long, c_int_p, c_int_p]),
    "boxmot_hip_reid_create": (_VP, [ctypes.c_char_p, _VP, ctypes.c_long, _I]),
    "boxmot_hip_reid_destroy": (None, [_VP]),
    "boxmot_hip_reid_feature_dim": (_I, [_VP]),
    "boxmot_hip_reid_set_mode": (_I, [_VP, _I]),
    "boxmot_hip_reid_set_preprocess": (_I, [_VP, ctypes.c_char_p]),
    "boxmot_hip_reid_compute_features": (_I, [_VP, _VP, _I, _I, _I, _VP, _I, _I, _VP, _I]),
    "boxmot_hip_reid_preprocess": (_I, [_VP, _VP, _I, _I, _I, _VP, _I, _I, _VP]),
    "boxmot_hip_reid_last_time_ms": (_I, [_VP, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "boxmot_hip_botsort_update_batch_frames": (_I, [_VP, _I, _VP, _VP, _VP, _I, _VP, _I, _I, _VP, _I, _VP]),
    "boxmot_hip_deepocsort_stream": (_VP, [_VP]),
    "boxmot_hip_strongsort_stream": (_VP, [_VP]),
    "boxmot_hip_ecc_create": (_VP, [_I, _I, _I, ctypes.c_double, ctypes.c_double, _I]),
    "boxmot_hip_ecc_destroy": (None, [_VP]),
    "boxmot_hip_ecc_reset": (_I, [_VP, _I]),
    "boxmot_hip_ecc_apply": (_I, [_VP, _I, _VP, _I, _I, _I, _VP, c_int_p]),
    "boxmot_hip_ecc_apply_device": (_I, [_VP, _I, _VP, _VP, c_int_p]),
    "boxmot_hip_sof_create": (_VP, [_I, _I, _I, ctypes.c_double, _I, ctypes.c_double, ctypes.c_double]),
    "boxmot_hip_sof_destroy": (None, [_VP]),
    "boxmot_hip_sof_reset": (_I, [_VP, _I]),
    "boxmot_hip_sof_apply": (_I, [_VP, _I, _VP, _I, _I, _I, _VP, _I, _I, _VP, c_int_p]),
    "boxmot_hip_sof_apply_device": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "boxmot_hip_sof_keypoints": (_I, [_VP, _I, _VP, _I, c_int_p]),
    "boxmot_hip_sof_debug_map": (_I, [_VP, _I, _I, _VP, _I, c_int_p, c_int_p]),
    "boxmot_hip_ingest_create": (_VP, [_I, _I, _I, _I]),
    "boxmot_hip_ingest_destroy": (None, [_VP]),
    "boxmot_hip_ingest_host_ptr": (_VP, [_VP, _I, _I]),
    "boxmot_hip_ingest_device_frames": (_VP, [_VP, _I]),
    "boxmot_hip_ingest_submit": (_I, [_VP, _I, _I]),
    "boxmot_hip_ingest_wait": (_I, [_VP, _I, _VP]),
    "boxmot_hip_ingest_release": (_I, [_VP, _I, _VP]),
    "boxmot_hip_ingest_host_done": (_I, [_VP, _I]),
    "boxmot_hip_last_error": (ctypes.c_char_p, []),
    "boxmot_hip_device_count": (_I, []),
    "boxmot_hip_botsort_device": (_I, [_VP]),
}


class RefBotSortConfig(ctypes.Structure):
    """``BoxMOTBotSortConfig`` exactly as the reference declares it (c_api.hpp:17-32; its ctypes twin is
    ``_BotSortCConfig``, boxmot/native/trackers/botsort.py:94-110)."""

    _fields_ = [
        ("track_high_thresh", ctypes.c_float), ("track_low_thresh", ctypes.c_float), ("new_track_thresh", ctypes.c_float),
        ("track_buffer", ctypes.c_int),
        ("match_thresh", ctypes.c_float), ("proximity_thresh", ctypes.c_float), ("appearance_thresh", ctypes.c_float),
        ("cmc_method", ctypes.c_char_p),
        ("frame_rate", ctypes.c_int), ("fuse_first_associate", ctypes.c_int), ("with_reid", ctypes.c_int), ("max_obs", ctypes.c_int),
        ("reid_model_path", ctypes.c_char_p), ("reid_preprocess", ctypes.c_char_p),
    ]


class RefByteTrackConfig(ctypes.Structure):
    """``BoxMOTByteTrackConfig`` (bytetrack/c_api.hpp:16-23)."""

    _fields_ = [("min_conf", ctypes.c_float), ("track_thresh", ctypes.c_float), ("match_thresh", ctypes.c_float),
                ("track_buffer", ctypes.c_int), ("frame_rate", ctypes.c_int), ("max_obs", ctypes.c_int)]


class RefOcSortConfig(ctypes.Structure):
    """``BoxMOTOCSORTConfig`` (ocsort/c_api.hpp:16-28)."""

    _fields_ = [("min_conf", ctypes.c_float), ("det_thresh", ctypes.c_float), ("iou_threshold", ctypes.c_float),
                ("max_age", ctypes.c_int), ("min_hits", ctypes.c_int), ("delta_t", ctypes.c_int), ("use_byte", ctypes.c_int),
                ("inertia", ctypes.c_float), ("q_xy_scaling", ctypes.c_float), ("q_s_scaling", ctypes.c_float),
                ("max_obs", ctypes.c_int)]


# every symbol include/boxmot_compat.h declares (the reference's own FFI names)
_UPD_EMBS = [_VP, _VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _I, c_int_p, c_int_p]
_UPD = [_VP, _VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _I, c_int_p, c_int_p]
COMPAT_SIGNATURES = {
    "boxmot_botsort_create": (_VP, [ctypes.POINTER(RefBotSortConfig)]),
    "boxmot_botsort_destroy": (None, [_VP]),
    "boxmot_botsort_reset": (_I, [_VP]),
    "boxmot_botsort_update": (_I, _UPD_EMBS),
    "boxmot_botsort_last_reid_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_botsort_last_reid_preprocess_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_botsort_last_reid_process_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_botsort_last_reid_postprocess_time_ms": (_I, [_VP, c_double_p]),
    "boxmot_botsort_last_error": (ctypes.c_char_p, []),
    "boxmot_bytetrack_create": (_VP, [ctypes.POINTER(RefByteTrackConfig)]),
    "boxmot_bytetrack_destroy": (None, [_VP]),
    "boxmot_bytetrack_reset": (_I, [_VP]),
    "boxmot_bytetrack_update": (_I, _UPD),
    "boxmot_bytetrack_last_error": (ctypes.c_char_p, []),
    "boxmot_ocsort_create": (_VP, [ctypes.POINTER(RefOcSortConfig)]),
    "boxmot_ocsort_destroy": (None, [_VP]),
    "boxmot_ocsort_reset": (_I, [_VP]),
    "boxmot_ocsort_update": (_I, _UPD),
    "boxmot_ocsort_last_error": (ctypes.c_char_p, []),
    "boxmot_reid_capi_create": (_I, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(_VP)]),
    "boxmot_reid_capi_destroy": (None, [_VP]),
    "boxmot_reid_capi_feature_dim": (_I, [_VP, c_int_p]),
    "boxmot_reid_capi_compute_features": (_I, [_VP, _VP, _I, _VP, _I, _I, _I, _VP, _I]),
    "boxmot_reid_capi_preprocess": (_I, [_VP, _VP, _I, _VP, _I, _I, _I]),
    "boxmot_reid_capi_process": (_I, [_VP]),
    "boxmot_reid_capi_postprocess": (_I, [_VP, _VP, _I]),
    "boxmot_reid_capi_last_error": (ctypes.c_char_p, []),
}

_lib = None


def load():
    """Load the shared library and attach signatures; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("BOXMOT_HIP_LIB") or LIB_PATH)       # a differently built library (tools/ab_variants.py)
    if not path.exists():
        raise ImportError(
            f"{path} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). boxmot_amd has no CPU fallback."
        )
    # PyTorch-ROCm wheels bundle their own HIP runtime.  When this library (linked against the system libamdhip64) is loaded
    # BEFORE torch in a process, torch's runtime later reports "No HIP GPUs are available" (seen on the MI355X box, ROCm 7.2
    # + torch 2.10-rocm7.0); loaded after torch, both coexist.  Everything in this package that touches torch device memory
    # needs that order, so torch -- when installed -- is imported first.  Pure ctypes users without torch are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(str(path))
    for name, (restype, argtypes) in list(SIGNATURES.items()) + list(COMPAT_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().boxmot_hip_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(ok: int) -> None:
    if ok == 0:
        raise RuntimeError(last_error())


_STATUS_MSG = __import__("re").compile(r"boxmot_hip: [\w-]+ stream \d+: ")


def step_ran(ok: int) -> bool:
    """True when the frame step of an update call ran on the device: the call succeeded, or it failed with a per-stream
    status report (capacity overflow, solver stall).  Those are raised after the step -- the device's frame counter has
    advanced and the rows of the frame were returned -- so the caller's frame counter must advance too."""
    return ok != 0 or bool(_STATUS_MSG.match(last_error()))
