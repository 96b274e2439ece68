"""The C-ABI library loads and exports every symbol include/boxmot_hip.h declares (no compute:
this runs without a GPU), and the ctypes table covers exactly that set."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "boxmot_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(boxmot_hip_[a-z0-9_]+)\s*\(", text)))


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


def test_ctypes_table_matches_header(built):
    from boxmot_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


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
