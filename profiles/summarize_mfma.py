"""Matrix-pipe utilisation per kernel from one rocprofv3 --pmc pass (rocpd sqlite):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d <out> -o p -- <cmd>
SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy, summed over the chip's SIMDs
(MI355X_MICROARCH.md: = 32 x N_mfma for a 32x32x16 MFMA; measured here: exactly 16 per v_mfma_f32_16x16x32_f16); GRBM_GUI_ACTIVE
is the kernel's busy time in shader cycles SUMMED OVER THE 8 XCDs (a 185 us dispatch reads 3.37 M = 8 x 421 k cycles).
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
Cross-check: the qkv GEMM of CLIP-ReID issues 7.13 M MFMAs = 116.9 GFLOP / 16384 FLOP (SQ_INSTS_MFMA agrees) in 185 us: 114 M busy
SIMD-cycles of 455 M available = 25 %, the same as its 632 TFLOP/s against the 2.5 PFLOP/s peak.
Values are averages per dispatch.  Usage: python profiles/summarize_mfma.py <pass.db> [min_us_filter]"""
import sqlite3
import sys
from collections import defaultdict

SIMDS = 256 * 4
XCDS = 8


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    per = defaultdict(dict)
    for k, c, n, s in rows:
        per[k][c] = (s / n, n)
    print("# kernel | dispatches | GRBM_GUI_ACTIVE (cycles) | SQ_INSTS_MFMA | SQ_VALU_MFMA_BUSY_CYCLES | busy cycles per MFMA | mfma_util")
    for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]):
        if k.startswith("__amd_rocclr") or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
            continue
        busy, n = d["SQ_VALU_MFMA_BUSY_CYCLES"]
        act = d.get("GRBM_GUI_ACTIVE", (0, 0))[0]
        insts = d.get("SQ_INSTS_MFMA", (0, 0))[0]
        if busy == 0:
            continue
        util = busy / (act / XCDS * SIMDS) if act else float("nan")
        print(f"{k[:72]:72s} {n:5d} {act:14.0f} {insts:14.0f} {busy:16.0f} {busy / insts if insts else 0:8.1f} {100 * util:7.2f}%")


if __name__ == "__main__":
    main()
