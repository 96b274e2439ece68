"""The C-ABI library loads and exports every symbol include/boxmot_hip.h declares (no compute:
this runs without a GPU), and the ctypes table covers exactly that set."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols(header="boxmot_hip.h", prefix="boxmot_hip_"):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def compat_symbols():
    return declared_symbols("boxmot_compat.h", "boxmot_(?:botsort|bytetrack|ocsort|reid_capi)_")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return ROOT / "boxmot_amd" / "libboxmot_hip.so"


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(str(built))
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/boxmot_hip.h but not exported"
    compat = compat_symbols()
    assert len(compat) == 27
    for n in compat:
        assert hasattr(lib, n), f"{n} declared in include/boxmot_compat.h but not exported"


def test_ctypes_table_matches_header(built):
    from boxmot_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert sorted(_lib.COMPAT_SIGNATURES) == compat_symbols()
    _lib.load()


def test_compat_structs_have_the_reference_layout():
    """boxmot_compat.h restates the reference's config structs: same field names, order and C types as the ctypes twins
    the reference itself binds with (float thresholds, native/trackers/botsort.py:94-110)."""
    from boxmot_amd import _lib
    names = [f[0] for f in _lib.RefBotSortConfig._fields_]
    assert names == ["track_high_thresh", "track_low_thresh", "new_track_thresh", "track_buffer", "match_thresh",
                     "proximity_thresh", "appearance_thresh", "cmc_method", "frame_rate", "fuse_first_associate", "with_reid",
                     "max_obs", "reid_model_path", "reid_preprocess"]
    assert ctypes.sizeof(_lib.RefBotSortConfig) == 72 and _lib.RefBotSortConfig.cmc_method.offset == 32
    assert ctypes.sizeof(_lib.RefByteTrackConfig) == 24 and ctypes.sizeof(_lib.RefOcSortConfig) == 44
    text = (ROOT / "include" / "boxmot_compat.h").read_text()
    body = text[text.index("struct BoxMOTBotSortConfig {"):text.index("};", text.index("struct BoxMOTBotSortConfig {"))]
    assert re.findall(r"(\w+);", body) == names


def test_compat_create_fails_loudly_without_a_device(built):
    from boxmot_amd import _lib
    lib = _lib.load()
    if lib.boxmot_hip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    cfg = _lib.RefBotSortConfig(0.5, 0.1, 0.6, 30, 0.8, 0.5, 0.25, b"none", 30, 0, 0, 50, None, None)
    assert not lib.boxmot_botsort_create(ctypes.byref(cfg))
    assert b"no HIP device" in lib.boxmot_botsort_last_error()
    h = ctypes.c_void_p()
    assert lib.boxmot_reid_capi_create(b"/nonexistent.osn1", None, ctypes.byref(h)) == 0 and not h.value
    assert lib.boxmot_reid_capi_last_error()


def test_no_device_fails_loudly(built):
    """Without a HIP device create() must fail with a message, never fall back to CPU."""
    from boxmot_amd import _lib
    lib = _lib.load()
    if lib.boxmot_hip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    cfg = _lib.BotSortConfig()
    lib.boxmot_hip_botsort_default_config(ctypes.byref(cfg))
    assert cfg.track_high_thresh == 0.5 and cfg.removed_stracks_buffer == 100 and cfg.n_streams == 1
    h = lib.boxmot_hip_botsort_create(ctypes.byref(cfg))
    assert not h
    assert "no HIP device" in _lib.last_error()
    with pytest.raises(RuntimeError, match="no HIP device"):
        from boxmot_amd.botsort import BotSort
        BotSort(use_cmc=False)


def test_product_does_not_import_oracle():
    """The product package may never route through the oracle."""
    for py in (ROOT / "boxmot_amd").rglob("*.py"):
        src = py.read_text()
        assert "import oracle" not in src and "from oracle" not in src, py
    for c in (ROOT / "boxmot_amd" / "csrc").iterdir():
        if c.suffix not in (".hpp", ".hip", ".h"):
            continue
        assert "oracle" not in c.read_text().replace("the oracle's", "").replace("oracle's", ""), c
