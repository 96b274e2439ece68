"""Static statistics of the compiled gfx950 kernels (no GPU needed): registers, spills, LDS, and the instruction mix per kernel,
with the quarter-rate integer multiplies and the other slow VALU forms counted separately.

    python tools/isa_stats.py [--filter SUBSTR] [--defines -DX=1 ...]

Compiles boxmot_amd/csrc/boxmot_hip.hip with the flags of __graft_entry__.build() plus -save-temps into a scratch directory and
reads the code-object metadata and the assembly.  `vgpr` is the allocation the occupancy follows (512 / vgpr waves per SIMD; the
`vgpr` column of rocprofv3's kernel trace reads half of it for these kernels)."""
import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

SLOW = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_")


def collect(defines=()):
    """{kernel name: {vgpr, agpr, sgpr, spill, scratch, lds, wg, valu, slow, mfma, lds_op, vmem, salu}} of the library's code object."""
    import __graft_entry__ as g
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", *g.HIPCC_FLAGS, *defines, "-save-temps=obj", "-o", str(Path(td) / "lib.so"), str(ROOT / "boxmot_amd" / "csrc" / "boxmot_hip.hip")]
        subprocess.run(cmd, check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = next(Path(td).glob("*gfx950.s")).read_text()
    md = yaml.safe_load(re.search(r"\.amdgpu_metadata\n(.*?)\n\s*\.end_amdgpu_metadata", asm, re.S).group(1))
    bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(\S+):\s+; @\1\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M)}
    out = {}
    for k in md["amdhsa.kernels"]:
        name = k[".name"]
        ops = re.findall(r"^\s+([a-z][a-z0-9_]+)", bodies.get(name, ""), re.M)
        n = lambda pred: sum(1 for o in ops if pred(o))
        out[name] = dict(
            vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0), sgpr=k[".sgpr_count"], spill=k[".vgpr_spill_count"],
            scratch=k[".private_segment_fixed_size"], lds=k[".group_segment_fixed_size"], wg=k[".max_flat_workgroup_size"],
            valu=n(lambda o: o.startswith("v_") and not o.startswith("v_mfma")), slow=n(lambda o: any(o.startswith(s) for s in SLOW)),
            mfma=n(lambda o: o.startswith("v_mfma")), lds_op=n(lambda o: o.startswith("ds_")),
            vmem=n(lambda o: o.startswith(("global_", "buffer_", "flat_", "scratch_"))), salu=n(lambda o: o.startswith("s_")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--defines", nargs="*", default=[])
    a = ap.parse_args()
    stats = collect(a.defines)
    print(f"{'kernel':64s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'spill':>5s} {'scratch':>7s} {'lds':>6s} {'wg':>5s} | {'valu':>5s} {'slow':>4s} {'mfma':>4s} {'lds_op':>6s} {'vmem':>4s} {'salu':>5s}")
    for name, k in sorted(stats.items()):
        if a.filter not in name:
            continue
        short = re.sub(r"^_ZN\d*(?:_GLOBAL__N_1)?", "", name)[:64]
        print(f"{short:64s} {k['vgpr']:4d} {k['agpr']:4d} {k['sgpr']:4d} {k['spill']:5d} {k['scratch']:7d} {k['lds']:6d} {k['wg']:5d} | "
              f"{k['valu']:5d} {k['slow']:4d} {k['mfma']:4d} {k['lds_op']:6d} {k['vmem']:4d} {k['salu']:5d}")


if __name__ == "__main__":
    main()
