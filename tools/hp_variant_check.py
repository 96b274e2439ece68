"""Builds the CPU-thread emulation of the fp32-grade x0.25 kernels (tests/host_emu/emu_reid.cpp: the device source, unchanged) with extra
-D switches and compares embeddings and every stored stage with the default build, bit for bit -- how a scheduling / synchronisation
variant of reid_hp.hpp is checked before it costs GPU time.   python tools/hp_variant_check.py -DBM_HP_NBR_SYNC=1 [-D...]"""
import ctypes
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
HERE = ROOT / "tests" / "host_emu"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(defs, out):
    subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-DEMU_DEFER_GLDS=1", *defs,
                           "-o", str(out), str(HERE / "emu_reid.cpp")])
    return ctypes.CDLL(str(out))


def forward(lib, blob, img, boxes):
    lib.emu_reid_forward_hp.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    n = len(boxes)
    feats = np.zeros((n, 512), np.float32)
    shapes = [(2048, 16), (2048, 64), (2048, 64), (512, 64), (512, 96), (512, 96), (128, 96), (128, 128), (128, 128)]
    bufs = [np.zeros((n,) + s, np.float32) for s in shapes]
    ptrs = (ctypes.c_void_p * 9)(*[b.ctypes.data for b in bufs])
    rc = lib.emu_reid_forward_hp(blob.ctypes.data, blob.size, img.ctypes.data, img.shape[1], img.shape[0], boxes.ctypes.data, n, feats.ctypes.data, ptrs, 1)
    assert rc == 0
    return feats, bufs


def main():
    from boxmot_amd.reid_weights import pack_osnet, random_osnet_state_dict
    import os
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    persist = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("PERSIST=")), None)      # PERSIST=G: the variant runs the
    os.environ.pop("EMU_HP_PERSIST_GRID", None)                                                          # persistent launch form on G workgroups
    blob = pack_osnet(random_osnet_state_dict("osnet_x0_25", seed=0))
    img = np.random.default_rng(5).integers(0, 255, (480, 641, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [-12.0, 300.0, 120.5, 500.0], [100, 100, 228, 356]], dtype=np.float32)
    with tempfile.TemporaryDirectory() as td:
        base = forward(build([], Path(td) / "base.so"), blob, img, boxes)
        if persist:
            os.environ["EMU_HP_PERSIST_GRID"] = persist
            defs = defs + [f"(persistent launch on {persist} workgroups)"]
        var = forward(build([d for d in defs if d.startswith("-D")], Path(td) / "var.so"), blob, img, boxes)
    same = np.array_equal(base[0], var[0]) and all(np.array_equal(a, b) for a, b in zip(base[1], var[1]))
    print(f"{' '.join(defs) or '(no switches)'}: embeddings max|diff| {np.abs(base[0] - var[0]).max():.3e}, "
          f"{'bit-identical to the default build' if same else 'DIFFERENT from the default build'}; |emb| {np.linalg.norm(var[0], axis=1)}")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
