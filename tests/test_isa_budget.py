"""Build-time guard of the occupancy the kernels are designed for (no GPU needed: hipcc cross-compiles, tools/isa_stats.py reads
the code object).  Two workgroups of eight waves per CU need <= 128 VGPRs; a 16-wave workgroup needs <= 128 as well."""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_hot_kernels_keep_their_register_budget():
    import isa_stats
    st = isa_stats.collect()
    find = lambda sub: {k: v for k, v in st.items() if sub in k}
    stem = find("k_stem_resize_fusedE")             # the fp16-operand family's stem
    assert len(stem) == 1
    for k, v in stem.items():
        assert v["vgpr"] <= 128 and v["scratch"] == 0, (k, v)               # two crops per CU, no spills
        assert v["mfma"] == 63, (k, v)                                      # strip-wise conv phase: 63 MFMAs per wave and band (84 before)
    blocks = {k: v for k, v in find("k_osblockI").items() if "ILi0E" in k or "ILi1E" in k}        # the fp16-operand family
    assert len(blocks) >= 4
    for k, v in blocks.items():
        assert v["vgpr"] <= 128 and v["scratch"] <= 64, (k, v)              # stages 0 and 1: two crops per CU
    stem_hp = find("k_stem_resize_fused_hp")
    assert len(stem_hp) == 1
    for k, v in stem_hp.items():
        assert v["vgpr"] <= 128 and v["scratch"] <= 32 and v["mfma"] == 126, (k, v)    # (hi, lo) weights: two MFMAs per k-step; the lo
                                                                                      # fragments live in LDS so two crops share a CU
    hp = find("k_osblock_hp")                                                # the fp32-grade family: one 8-wave workgroup per CU in
    assert len(hp) == 6                                                      # stages 0 / 1 (<= 256 registers), two in stage 2 (<= 128)
    for k, v in hp.items():
        assert v["vgpr"] <= (128 if "ILi2E" in k else 256) and v["scratch"] == 0, (k, v)
    for k, v in find("strongsort_step_kernelILi1024").items():
        assert v["vgpr"] <= 128, (k, v)                                     # 16 waves = 4 per SIMD
    for k, v in find("k_gemm_f16_glds").items():
        assert v["scratch"] == 0, (k, v)
