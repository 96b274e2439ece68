"""Generates a PROBE variant of boxmot_amd/csrc/reid_hp.hpp (not part of the library, not committed: written to the directory given)
in which the depthwise taps of stages 0 / 1 are read from LDS at their use instead of being held in 40 registers per lane -- the
missing registers of a 16-wave workgroup (BM_HP_NW0=16).  Build tools/hp_prof against it with the variant directory first on the
include path:

    python tools/hp_taps_lds_variant.py /tmp/hpv
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DBM_HP_NW0=16 -DBM_HP_EPI_TG0E=1 -I /tmp/hpv -I boxmot_amd/csrc \
          tools/hp_prof.hip -o tools/_build/hp_prof_r5_nw16_tapslds_tg1

Timing probe only (profiles/r5_hp_s0_ab.txt, last block): the 16-wave build changes the order of the gates' sums (fp32 round-off), so
its checksums differ from the 8-wave build's by construction."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def main():
    out = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/hpv")
    out.mkdir(parents=True, exist_ok=True)
    s = (ROOT / "boxmot_amd" / "csrc" / "reid_hp.hpp").read_text()
    a = s.index("#else\n#pragma unroll\n                    for (int sq = 0; sq < NSEQ; ++sq) {\n                        f4 acc[3];\n#pragma unroll\n"
                "                        for (int rr = 0; rr < L + 2; ++rr) {          // input row")
    b = s.index("#endif\n                }\n            }", a)
    blk = re.sub(r"wd\[(\d)\]", r"WDL(\1)", s[a:b]).replace("fma_f4(WDL(0), v0, bias)", "fma_f4(WDL(0), v0, BIASL())")
    blk = blk.replace(
        "                            const unsigned char* rp = cbase + sq * 256 + (rr - 1) * G::ROWP;",
        "                            const unsigned char* rp = cbase + sq * 256 + (rr - 1) * G::ROWP;\n"
        "                            unsigned tap_o = 0; BM_OPAQUE_U32(tap_o);      // (per row: the reads are not hoisted back into registers)\n"
        "#define WDL(t) (*reinterpret_cast<const f4*>(wdl + tap_o + ((ct * 4 + g) * 9 + (t)) * 16))\n"
        "#define BIASL() (*reinterpret_cast<const f4*>(wdl + tap_o + MIDP * 9 * 4 + (16 * ct + 4 * g) * 4))")
    s2 = s[:a] + blk + s[b:]
    s2 = s2.replace("#pragma unroll\n            for (int c = 0; c < NWD; ++c) load_dw(c, wdv[c], dbias[c]);",
                    "            if constexpr (STAGE == 2) {\n#pragma unroll\n            for (int c = 0; c < NWD; ++c) load_dw(c, wdv[c], dbias[c]);\n            }")
    (out / "reid_hp.hpp").write_text(s2)
    print(f"wrote {out / 'reid_hp.hpp'}")


if __name__ == "__main__":
    main()
