"""Recipe for ``oracle/_ref/``: the REAL reference, compiled, so that it can travel to the GPU box like a prebuilt ``.so``.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (``bench.py``'s ``cpu_baseline`` leg and ``tests/``; never imported by ``boxmot_amd``).

``/root/reference`` exists in the build container only.  north_star asks for "the reference boxmot CPU tracker timed on the same
box's host cores", so the reference's own modules have to run on the GPU box.  For a compiled reference the task's rule is "build it
from the sources where they lie, outputs only into the git-ignored ``oracle/_ref/``"; for this Python reference the build step is
``compile()``: every module the harness (``oracle/ref_harness.py``) imports for the benchmarked callers is byte-compiled FROM
``/root/reference`` into a sourceless ``.pyc`` at the same relative path under ``oracle/_ref/`` -- no reference source text is
written anywhere in the repository (history or working tree), the ``.pyc`` files are build outputs like ``liboracle.so``, and
``oracle/_ref/`` is listed in ``.gitignore`` (not in ``.gpurunignore``: it ships with the gpurun snapshot).

    python oracle/make_ref.py          (``__graft_entry__.build()`` runs it whenever /root/reference is present)

What is compiled: the closure of ``load_botsort`` / ``load_bytetrack`` / ``load_osnet_module`` / ``RefReID`` (the reference's
``trackers/__init__`` imports every tracker, hence ~65 modules), found by importing them here and listing ``sys.modules``; plus
the two ReID files the harness loads by path (``reid/backbones/osnet.py``, ``reid/core/preprocessing.py``) and the method subset of
``reid/backends/base_backend.py`` that ``RefReID`` executes (marshalled code object of the SAME ast filter ref_harness applies).
A manifest (relative path, sha256 of the source it was compiled from) is written beside them.
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import marshal
import py_compile
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SRC = Path("/root/reference")
DST = ROOT / "oracle" / "_ref"

BY_PATH = ("boxmot/reid/backbones/osnet.py", "boxmot/reid/core/preprocessing.py")
BACKEND = "boxmot/reid/backends/base_backend.py"
BACKEND_KEEP = ("get_crops", "get_features", "_is_obb_box", "_boxes_to_xyxy", "inference_preprocess", "inference_postprocess", "to_numpy")


def backend_subset_code(src_text: str):
    """Code object of ``BaseModelBackend`` reduced to the crop / feature methods (the filter RefReID has always applied)."""
    import ast

    tree = ast.parse(src_text)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BaseModelBackend")
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in BACKEND_KEEP]
    cls.bases = []
    return compile(ast.Module(body=[cls], type_ignores=[]), "base_backend_subset", "exec")


def main() -> int:
    if not (SRC / "boxmot" / "trackers" / "bbox" / "botsort" / "botsort.py").exists():
        print("[make_ref] /root/reference is absent: nothing to compile (the GPU box uses the prebuilt oracle/_ref/)", file=sys.stderr)
        return 0
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    from oracle import ref_harness as rh

    if rh.REFERENCE_ROOT != SRC:
        print(f"[make_ref] harness root is {rh.REFERENCE_ROOT}, not {SRC}: refusing to compile from a compiled copy", file=sys.stderr)
        return 1
    rh.load_botsort()
    rh.load_bytetrack()
    mods = sorted({Path(m.__file__) for m in list(sys.modules.values())
                   if getattr(m, "__file__", None) and str(m.__file__).startswith(str(SRC) + "/") and str(m.__file__).endswith(".py")})
    rels = sorted({str(p.relative_to(SRC)) for p in mods} | set(BY_PATH))
    if DST.exists():
        shutil.rmtree(DST)
    manifest = {}
    for rel in rels:
        src = SRC / rel
        out = (DST / rel).with_suffix(".pyc")
        out.parent.mkdir(parents=True, exist_ok=True)
        # unchecked pyc: the import system never looks for the (absent) source; dfile keeps reference-relative names in tracebacks
        py_compile.compile(str(src), cfile=str(out), dfile=f"<reference>/{rel}", doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest[rel] = hashlib.sha256(src.read_bytes()).hexdigest()[:16]
    text = (SRC / BACKEND).read_text()
    out = DST / "base_backend_subset.marshal"
    out.write_bytes(marshal.dumps(backend_subset_code(text)))
    manifest[BACKEND + " (method subset " + ",".join(BACKEND_KEEP) + ")"] = hashlib.sha256(text.encode()).hexdigest()[:16]
    (DST / "MANIFEST.json").write_text(json.dumps({"python": sys.version.split()[0], "magic": importlib.util.MAGIC_NUMBER.hex(),
                                                   "compiled_from": str(SRC), "modules": manifest}, indent=1))
    print(f"[make_ref] {len(rels)} reference modules byte-compiled into {DST.relative_to(ROOT)}/ (sourceless .pyc; no source text copied)",
          file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
