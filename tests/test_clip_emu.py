"""Runs the CLIP-ReID kernels (boxmot_amd/csrc/clip_kernels.hpp, device source unchanged) on CPU threads with the emulated
MFMA of tests/host_emu and compares the embeddings with the torch fp32 oracle (oracle/clipreid.py) on reduced geometries.
Validates the CLP1 blob layout, the GEMM tiling / epilogues, the attention and the necks without a GPU.  Not a product path."""
import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent / "host_emu"
CLANG = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")


def _build():
    out = HERE / "libemu_clip.so"
    csrc = HERE.parent.parent / "boxmot_amd" / "csrc"
    deps = [HERE / "emu_clip.cpp", HERE / "hip_shim.hpp", csrc / "clip_kernels.hpp", csrc / "reid_pack.hpp", csrc / "kernel_macros.hpp"]
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread",
                               "-ffp-contract=off", "-DEMU_DEFER_GLDS=1",       # global -> LDS copies land at the issuing thread's BM_WAIT_VM0 (hip_shim.hpp)
                               "-o", str(out), str(HERE / "emu_clip.cpp")])
    return out


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
@pytest.mark.parametrize("width,layers,out_dim,hw,n", [(128, 2, 128, (32, 32), 3), (256, 1, 128, (48, 16), 2), (128, 1, 128, (128, 64), 2), (768, 1, 128, (32, 16), 2)])
def test_clip_kernels_emulated_vs_oracle(width, layers, out_dim, hw, n):
    import torch

    from boxmot_amd.clip_weights import pack_clipreid, random_clipreid_state_dict
    from oracle.clipreid import clipreid_forward

    lib = ctypes.CDLL(str(_build()))
    lib.emu_clip_forward.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    sd = random_clipreid_state_dict(7, width=width, layers=layers, out_dim=out_dim, input_hw=hw)
    blob = pack_clipreid(sd, hw)
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (n, hw[0], hw[1], 3), dtype=np.uint8)
    nhwc = ((u8.astype(np.float32) / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)
    feats = np.zeros((n + 1, width + out_dim), np.float32)
    rows = np.array([2, 0, 3][:n] if n == 3 else [1, 0], dtype=np.int32)         # scattered output rows, like the tracker's crop list
    assert lib.emu_clip_forward(blob.ctypes.data, blob.size, nhwc.ctypes.data, n, rows.ctypes.data, feats.ctypes.data) == 0
    want = clipreid_forward(sd, torch.from_numpy(np.ascontiguousarray(nhwc.transpose(0, 3, 1, 2)))).numpy()
    want = want / np.linalg.norm(want, axis=1, keepdims=True)
    got = feats[rows]
    err = np.abs(got - want).max()
    print(f"CLIP-ReID emulated (width {width}, {layers} layers): max|diff| = {err:.2e}")
    assert err < 1e-3
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
def test_compile_time_attention_kernel_equals_the_run_time_one_and_softmax():
    """k_clip_attention_t<T> (three query tiles per wave, V^T fragments by the emulated ds_read_b64_tr_b16) against k_clip_attention on the
    same q | k | v rows: identical halves, and both equal softmax(q k^T / 8) v in float64 (random, non-symmetric inputs: a transposed
    fragment or a permuted key order shows)."""
    lib = ctypes.CDLL(str(_build()))
    lib.emu_clip_attention_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    n, heads, D, T = 2, 2, 128, 33
    rng = np.random.default_rng(11)
    qkv = (rng.standard_normal((n * T, 3 * D)) * 1.5).astype(np.float16)
    a = np.full((n * T, D), np.nan, np.float16)
    b = np.full((n * T, D), np.nan, np.float16)
    assert lib.emu_clip_attention_pair(qkv.ctypes.data, a.ctypes.data, b.ctypes.data, n, heads, D) == 0
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
    x = qkv.astype(np.float64).reshape(n, T, 3, heads, 64)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]                           # [n][T][heads][64]
    s = np.einsum("nqhd,nkhd->nhqk", q, k) * 0.125
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    want = np.einsum("nhqk,nkhd->nqhd", p, v).reshape(n * T, D)
    err = np.abs(b.astype(np.float64) - want).max()
    print(f"attention (T = {T}) emulated: max|diff vs float64| = {err:.2e}")
    assert err < 4e-3
