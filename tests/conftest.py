import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fast: the short GPU tier (-m 'gpu and fast', about a minute): one golden per tracker, one ReID gate per "
                            "family, the compat ABI, the cost values, one long-parity case -- for intermediate GPU sessions")


# The short GPU tier: `pytest -m "gpu and fast"` -- one golden per tracker, one ReID gate per kernel family, the
# reference-named ABI, the cost values (marked in tests/test_gpu_cost_values.py), one long-parity case per configuration.  The full
# `-m gpu` suite runs once per round on the shipped library (tools/gpu_session.sh: `fast` vs `tests`).
FAST_GPU = (
    "test_gpu_botsort.py::test_hip_matches_reference_golden_and_oracle[c2_yaml]",
    "test_gpu_deepocsort.py::test_hip_deepocsort_matches_reference_golden_and_oracle[docs_c2]",
    "test_gpu_strongsort.py::test_hip_strongsort_matches_reference_golden_and_oracle[ss_c2]",
    "test_gpu_bytetrack.py::test_config0_bytetrack_32_dets_640x640",
    "test_gpu_reid.py::test_crops_bit_exact_vs_oracle_and_reference",
    "test_gpu_reid.py::test_fused_families_on_calibrated_weights[0]",
    "test_gpu_reid.py::test_botsort_with_reid_in_the_loop_matches_oracle_ids[2]",
    "test_gpu_long_parity.py::test_osnet_x1_0_fp32_grade_family_on_calibrated_weights[0]",
    "test_gpu_long_parity.py::test_long_config2_reid_inside_update_240_frames_vs_reference_rows[calib-2]",
    "test_gpu_long_parity.py::test_config2_operating_point_256_streams_sampled_streams_vs_reference_rows",
    "test_gpu_long_parity.py::test_config3_reid_inside_update_full_size_vs_reference_rows[2-calib-160]",
    "test_gpu_clipreid.py::test_vitb16_features_on_gain_randomised_weights[0]",
    "test_gpu_compat_abi.py::",
)


def pytest_collection_modifyitems(config, items):
    import pytest

    for it in items:
        nid = it.nodeid.split("tests/")[-1]
        if any(nid == f or (f.endswith("::") and nid.startswith(f)) for f in FAST_GPU):
            it.add_marker(pytest.mark.fast)
