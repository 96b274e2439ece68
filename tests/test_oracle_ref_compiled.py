"""oracle/_ref/ -- the reference's own modules byte-compiled by oracle/make_ref.py so that bench.py's cpu_baseline leg can time the
reference itself on the GPU box (where /root/reference does not exist): the compiled copy imports, runs, and returns the rows the
oracle returns; the directory holds build outputs only (no source text) and is git-ignored."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF = ROOT / "oracle" / "_ref"


def _ensure_built():
    if not (REF / "MANIFEST.json").exists():
        if not Path("/root/reference/boxmot").exists():
            pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/ is present")
        subprocess.check_call([sys.executable, str(ROOT / "oracle" / "make_ref.py")], cwd=str(ROOT))


def test_compiled_reference_holds_no_source_text_and_is_git_ignored():
    _ensure_built()
    files = [p for p in REF.rglob("*") if p.is_file()]
    assert files and all(p.suffix in (".pyc", ".marshal", ".json") for p in files), [p.name for p in files if p.suffix not in (".pyc", ".marshal", ".json")]
    assert "oracle/_ref/" in (ROOT / ".gitignore").read_text().split()
    ignore = ROOT / ".gpurunignore"
    assert not ignore.exists() or "oracle/_ref" not in ignore.read_text()         # it has to travel to the GPU box
    man = json.loads((REF / "MANIFEST.json").read_text())
    assert "boxmot/trackers/bbox/botsort/botsort.py" in man["modules"] and "boxmot/reid/backbones/osnet.py" in man["modules"]


def test_compiled_reference_runs_and_agrees_with_the_oracle():
    _ensure_built()
    code = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
from oracle import ref_harness as rh
assert rh.reference_kind() == "compiled" and not rh.reference_available() and rh.reference_runnable()
from boxmot_amd.scenario import Scenario
from oracle.botsort import BotSortOracle
from oracle.bytetrack import ByteTrackOracle
ref = rh.load_botsort()(reid_model=None, use_cmc=False)
assert sys.modules[type(ref).__module__].__file__.endswith(".pyc")
orc = BotSortOracle()
sc = Scenario(16, 40, width=640, height=480, emb_dim=32, random_image=False)
for t in range(10):
    d, e = sc.frame(t)
    a, b = np.asarray(ref.update(d, sc.image, e.copy())), orc.update(d, sc.image, e.copy())
    assert a.shape == b.shape and np.array_equal(a, b), t
bt, bo = rh.load_bytetrack()(), ByteTrackOracle()
sc = Scenario(32, 32, width=640, height=640, emb_dim=8, random_image=False)
for t in range(10):
    d, _ = sc.frame(t)
    a, b = np.asarray(bt.update(d, sc.image)), bo.update(d, sc.image)
    assert a.shape == b.shape and np.array_equal(a, b), t
osn = rh.load_osnet_module().osnet_x0_25(num_classes=1, pretrained=False).eval()
reid = rh.RefReID(osn)
f = reid.get_features(np.array([[10., 20., 90., 200.]], dtype=np.float32), np.zeros((480, 640, 3), np.uint8))
assert f.shape == (1, 512)
print("OK")
""" % str(ROOT)
    env = dict(os.environ, BOXMOT_ORACLE_REF="compiled")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stderr[-2000:]


def test_a_compiled_copy_of_another_interpreter_counts_as_absent(tmp_path, monkeypatch):
    """oracle/_ref/ byte-compiled by a different Python minor version cannot be imported ("bad magic number"): the harness must say
    "not runnable" so that bench.py falls back to the port instead of recording errors (the MANIFEST's magic is compared)."""
    import importlib.util
    import json

    from oracle import ref_harness

    monkeypatch.setattr(ref_harness, "COMPILED_ROOT", tmp_path)
    assert ref_harness._compiled_usable() is False                       # no manifest
    (tmp_path / "MANIFEST.json").write_text(json.dumps({"python": "3.99.0", "magic": "deadbeef"}))
    assert ref_harness._compiled_usable() is False
    (tmp_path / "MANIFEST.json").write_text(json.dumps({"python": "this", "magic": importlib.util.MAGIC_NUMBER.hex()}))
    assert ref_harness._compiled_usable() is True
    (tmp_path / "MANIFEST.json").write_text("not json")
    assert ref_harness._compiled_usable() is False
