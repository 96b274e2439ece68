"""TEST-ONLY ctypes driver for tests/host_emu (device kernel source run on CPU threads)."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent / "host_emu"
LIB = HERE / "libemu_botsort.so"

CFG_D = ("track_high_thresh", "track_low_thresh", "new_track_thresh", "match_thresh", "proximity_thresh",
         "appearance_thresh", "second_match_thresh", "unconfirmed_match_thresh", "unconfirmed_emb_scale")
CFG_I = ("fuse_first_associate", "with_reid", "frame_rate", "track_buffer", "removed_stracks_buffer", "kind")


def build(sanitize: bool = False, dense: bool = False, threads: int = 64, obb: bool = False) -> Path:
    src = HERE / "emu_botsort.cpp"
    deps = [src, HERE / "hip_shim.hpp"] + list((HERE.parent.parent / "boxmot_amd" / "csrc").glob("botsort_*.hpp")) \
        + [HERE.parent.parent / "boxmot_amd" / "csrc" / "block_prims.hpp",
           HERE.parent.parent / "boxmot_amd" / "csrc" / "kernel_macros.hpp",
           HERE.parent.parent / "boxmot_amd" / "csrc" / "obb_geometry.hpp"]
    out = HERE / ("libemu_botsort_asan.so" if sanitize else ("libemu_botsort_dense.so" if dense else
                                                              ("libemu_botsort.so" if threads == 64 else f"libemu_botsort_t{threads}.so")))
    if obb:         # the oriented-box copy of the step (bm::obb)
        out = out.with_name(out.name.replace("libemu_botsort", "libemu_botsort_obb"))
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread"]
        if sanitize:
            flags += ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
        if dense:
            flags += ["-DBM_SPARSE_MAX=0"]      # force the dense LDS-tiled cosine path
        subprocess.check_call(["g++", *flags, f"-DEMU_NTHR={threads}", f"-DEMU_OBB={int(obb)}", "-o", str(out), str(src)])
    return out


class EmuBotSort:
    def __init__(self, cfg: dict, cap=256, nd=64, dim=32, sanitize=False, dense=False, threads=64, obb=False, n_lists=1):
        self.lib = ctypes.CDLL(str(build(sanitize, dense, threads=threads, obb=obb)))
        self.det_cols, self.out_cols, self.kf_stride = (7, 9, 110) if obb else (6, 8, 72)
        self.lib.emu_create_lists.restype = ctypes.c_void_p
        self.lib.emu_create_lists.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self.lib.emu_update_list.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self.lib.emu_dump.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
        self.lib.emu_destroy.argtypes = [ctypes.c_void_p]
        cd = np.array([cfg[k] for k in CFG_D], dtype=np.float64)
        ci = np.array([int(cfg.get(k, 0)) for k in CFG_I], dtype=np.int32)
        self.cap, self.nd, self.dim = cap, nd, dim
        self.h = self.lib.emu_create_lists(cd.ctypes.data, ci.ctypes.data, cap, nd, dim, n_lists)

    def update(self, dets, embs=None, warp=None, class_list=0, frame_count=-1):
        if warp is not None:
            w = np.ascontiguousarray(warp, dtype=np.float64).reshape(6)
            self.lib.emu_set_warp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            self.lib.emu_set_warp(self.h, w.ctypes.data)
        dets = np.ascontiguousarray(dets, dtype=np.float32).reshape(-1, self.det_cols)
        n = len(dets)
        e = None if embs is None else np.ascontiguousarray(embs, dtype=np.float32)
        out = np.zeros((self.nd, self.out_cols), dtype=np.float32)
        out_n = ctypes.c_int(0)
        status = self.lib.emu_update_list(self.h, dets.ctypes.data, n, None if e is None else e.ctypes.data,
                                          out.ctypes.data, ctypes.byref(out_n), int(class_list), int(frame_count))
        if status != 0:
            raise RuntimeError(f"emulated kernel status {status}")
        return out[: out_n.value].copy()

    def debug_costs_enable(self):
        self.lib.emu_debug_costs_enable.argtypes = [ctypes.c_void_p]
        self.lib.emu_debug_costs_enable(self.h)

    def debug_costs(self, stage, plane=0):
        """(tracks, detections) fp64 cost matrix of the last update (boxmot_hip_botsort_debug_costs' planes)."""
        self.lib.emu_debug_costs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        buf = np.zeros(self.cap * self.nd, dtype=np.float64)
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        n = self.lib.emu_debug_costs(self.h, stage, plane, buf.ctypes.data, ctypes.byref(r), ctypes.byref(c))
        if n < 0:
            raise RuntimeError("debug costs are not enabled")
        return buf[:n].reshape(r.value, c.value).copy()

    def dump(self, which):
        ints = np.zeros((self.cap, 6), dtype=np.int32)
        kf = np.zeros((self.cap, self.kf_stride), dtype=np.float64)
        sm = np.zeros((self.cap, self.dim), dtype=np.float32)
        misc = np.zeros((self.cap, 3), dtype=np.float32)
        cnt = np.zeros(3, dtype=np.int32)
        n = self.lib.emu_dump(self.h, which, ints.ctypes.data, kf.ctypes.data, sm.ctypes.data, misc.ctypes.data,
                              cnt.ctypes.data)
        return dict(n=n, ints=ints[:n], kf=kf[:n], smooth=sm[:n], misc=misc[:n], counters=cnt)

    def close(self):
        if self.h:
            self.lib.emu_destroy(self.h)
            self.h = None


DOCS_D = ("det_thresh", "iou_threshold", "inertia", "w_association_emb", "alpha_fixed_emb", "aw_param", "Q_xy_scaling", "Q_s_scaling",
          "min_conf", "asso_diag")
DOCS_I = ("max_age", "min_hits", "delta_t", "embedding_off", "aw_off", "use_byte", "asso_mode")
ASSO_MODES = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3, "hmiou": 4, "centroid": 5}


def build_docs(sanitize: bool = False, threads: int = 64, obb: bool = False) -> Path:
    src = HERE / "emu_docs.cpp"
    csrc = HERE.parent.parent / "boxmot_amd" / "csrc"
    deps = [src, HERE / "hip_shim.hpp", csrc / "deepocsort_step.hpp", csrc / "deepocsort_step_body.hpp", csrc / "obb_geometry.hpp", csrc / "lap_jv.hpp",
            csrc / "block_prims.hpp", csrc / "kernel_macros.hpp", csrc / "botsort_types.hpp"]
    out = HERE / ("libemu_docs_asan.so" if sanitize else ("libemu_docs.so" if threads == 64 else f"libemu_docs_t{threads}.so"))
    if obb:         # the oriented copy of the step (bm::obb)
        out = out.with_name(out.name.replace("libemu_docs", "libemu_docs_obb"))
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread"]
        if sanitize:
            flags += ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
        subprocess.check_call(["g++", *flags, f"-DEMU_NTHR={threads}", f"-DEMU_OBB={int(obb)}", "-o", str(out), str(src)])
    return out


class EmuDeepOcSort:
    """The DeepOCSORT device step (deepocsort_step.hpp) executed on CPU threads."""

    def __init__(self, cfg: dict, cap=256, nd=64, dim=32, sanitize=False, threads=64, obb=False):
        self.lib = ctypes.CDLL(str(build_docs(sanitize, threads=threads, obb=obb)))
        self.det_cols, self.out_cols, self.kf_stride = (7, 9, 90) if obb else (6, 8, 72)
        self.lib.emu_docs_create.restype = ctypes.c_void_p
        self.lib.emu_docs_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self.lib.emu_docs_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p]
        self.lib.emu_docs_dump.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
        self.lib.emu_docs_destroy.argtypes = [ctypes.c_void_p]
        cfg = {"min_conf": 0.1, "use_byte": 0, "asso_func": "iou", "frame_wh": (0, 0), **cfg}
        cfg["asso_mode"] = ASSO_MODES[cfg["asso_func"]]
        fw, fh = cfg["frame_wh"] or (0, 0)
        cfg["asso_diag"] = float(np.sqrt(fw ** 2 + fh ** 2))
        cd = np.array([cfg[k] for k in DOCS_D], dtype=np.float64)
        ci = np.array([int(cfg[k]) for k in DOCS_I], dtype=np.int32)
        self.cap, self.nd, self.dim = cap, nd, dim
        self.h = self.lib.emu_docs_create(cd.ctypes.data, ci.ctypes.data, cap, nd, dim)

    def update(self, dets, embs=None, warp=None):
        dets = np.ascontiguousarray(dets, dtype=np.float32).reshape(-1, self.det_cols)
        n = len(dets)
        e = None if embs is None else np.ascontiguousarray(embs, dtype=np.float32)
        w = None if warp is None else np.ascontiguousarray(warp, dtype=np.float64).reshape(6)
        out = np.zeros((self.cap, self.out_cols), dtype=np.float32)
        out_n = ctypes.c_int(0)
        status = self.lib.emu_docs_update(self.h, dets.ctypes.data, n, None if e is None else e.ctypes.data,
                                          None if w is None else w.ctypes.data, out.ctypes.data, ctypes.byref(out_n))
        if status != 0:
            raise RuntimeError(f"emulated kernel status {status}")
        return out[: out_n.value].copy()

    def debug_costs_enable(self):
        self.lib.emu_docs_debug_costs_enable.argtypes = [ctypes.c_void_p]
        self.lib.emu_docs_debug_costs_enable(self.h)

    def debug_costs(self, plane=0):
        """((detections, tracks) fp64 matrix, branch) of the last update's ``associate`` (boxmot_hip_deepocsort_debug_costs' planes)."""
        self.lib.emu_docs_debug_costs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        buf = np.zeros(self.cap * self.nd, dtype=np.float64)
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        branch = self.lib.emu_docs_debug_costs(self.h, plane, buf.ctypes.data, ctypes.byref(r), ctypes.byref(c))
        if branch < 0:
            raise RuntimeError("debug costs are not enabled")
        return buf[: r.value * c.value].reshape(r.value, c.value).copy(), branch

    def dump(self):
        ints = np.zeros((self.cap, 5), dtype=np.int32)
        kf = np.zeros((self.cap, self.kf_stride), dtype=np.float64)
        emb = np.zeros((self.cap, self.dim), dtype=np.float64)
        cnt = np.zeros(2, dtype=np.int32)
        n = self.lib.emu_docs_dump(self.h, ints.ctypes.data, kf.ctypes.data, emb.ctypes.data, cnt.ctypes.data)
        return dict(n=n, ints=ints[:n], kf=kf[:n], emb=emb[:n], counters=cnt)

    def close(self):
        if self.h:
            self.lib.emu_docs_destroy(self.h)
            self.h = None


SS_D = ("min_conf", "max_cos_dist", "max_iou_dist", "mc_lambda", "ema_alpha")
SS_I = ("max_age", "n_init", "nn_budget")


def build_ss(sanitize: bool = False, threads: int = 64) -> Path:
    """threads: workgroup size of the emulated kernels (64 = one wavefront; 256 exercises the cross-wavefront paths)."""
    src = HERE / "emu_ssort.cpp"
    csrc = HERE.parent.parent / "boxmot_amd" / "csrc"
    deps = [src, HERE / "hip_shim.hpp", csrc / "strongsort_step.hpp", csrc / "block_prims.hpp", csrc / "kernel_macros.hpp",
            csrc / "botsort_types.hpp"]
    out = HERE / ("libemu_ssort_asan.so" if sanitize else ("libemu_ssort.so" if threads == 64 else f"libemu_ssort_t{threads}.so"))
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread"]
        if sanitize:
            flags += ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
        subprocess.check_call(["g++", *flags, f"-DEMU_NTHR={threads}", *(["-DBM_LSA_SCAN_THREADS=128"] if threads > 128 else []), "-o", str(out), str(src)])
    return out


class EmuStrongSort:
    """The StrongSORT device kernels (strongsort_step.hpp) executed on CPU threads."""

    def __init__(self, cfg: dict, cap=256, nd=64, dim=32, sanitize=False, threads=64):
        self.lib = ctypes.CDLL(str(build_ss(sanitize, threads=threads)))
        self.lib.emu_ss_create.restype = ctypes.c_void_p
        self.lib.emu_ss_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self.lib.emu_ss_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]
        self.lib.emu_ss_dump.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
        self.lib.emu_ss_destroy.argtypes = [ctypes.c_void_p]
        cd = np.array([cfg[k] for k in SS_D], dtype=np.float64)
        ci = np.array([int(cfg[k]) for k in SS_I], dtype=np.int32)
        self.cap, self.nd, self.dim = cap, nd, dim
        self.h = self.lib.emu_ss_create(cd.ctypes.data, ci.ctypes.data, cap, nd, dim)

    def update(self, dets, embs, warp=None):
        dets = np.ascontiguousarray(dets, dtype=np.float32).reshape(-1, 6)
        n = len(dets)
        e = np.ascontiguousarray(embs, dtype=np.float32)
        w = None if warp is None else np.ascontiguousarray(warp, dtype=np.float64).reshape(6)
        out = np.zeros((self.cap, 8), dtype=np.float32)
        out_n = ctypes.c_int(0)
        status = self.lib.emu_ss_update(self.h, dets.ctypes.data, n, e.ctypes.data, None if w is None else w.ctypes.data,
                                        out.ctypes.data, ctypes.byref(out_n))
        if status != 0:
            raise RuntimeError(f"emulated kernel status {status}")
        return out[: out_n.value].copy()

    def app(self, rows: int, cols: int) -> np.ndarray:
        """Appearance distances of the last step: [list position before the step][detection index], fp32."""
        out = np.zeros((max(rows, 1), max(cols, 1)), dtype=np.float32)
        self.lib.emu_ss_app.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self.lib.emu_ss_app(self.h, out.ctypes.data, out.shape[0], out.shape[1])
        return out[:rows, :cols]

    def debug_costs_enable(self):
        self.lib.emu_ss_debug_costs_enable.argtypes = [ctypes.c_void_p]
        self.lib.emu_ss_debug_costs_enable(self.h)

    def debug_costs(self, stage, plane=0):
        """(tracks, detections) fp64 matrix of the last update's min_cost_matching call ``stage`` (boxmot_hip_strongsort_debug_costs' planes)."""
        self.lib.emu_ss_debug_costs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        big = max(self.cap, self.nd)
        buf = np.zeros(big * big, dtype=np.float64)
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        n = self.lib.emu_ss_debug_costs(self.h, stage, plane, buf.ctypes.data, ctypes.byref(r), ctypes.byref(c))
        if n < 0:
            raise RuntimeError("debug costs are not enabled")
        return buf[:n].reshape(r.value, c.value).copy()

    def dump(self):
        ints = np.zeros((self.cap, 6), dtype=np.int32)
        kf = np.zeros((self.cap, 72), dtype=np.float64)
        feat = np.zeros((self.cap, self.dim), dtype=np.float32)
        cnt = np.zeros(2, dtype=np.int32)
        n = self.lib.emu_ss_dump(self.h, ints.ctypes.data, kf.ctypes.data, feat.ctypes.data, cnt.ctypes.data)
        return dict(n=n, ints=ints[:n], kf=kf[:n], feat=feat[:n], counters=cnt)

    def close(self):
        if self.h:
            self.lib.emu_ss_destroy(self.h)
            self.h = None
