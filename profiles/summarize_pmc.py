"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (rocpd sqlite databases).
Usage: python profiles/summarize_pmc.py <fetch_pass.db> <write_pass.db> <crops_per_launch> > profiles/<round>_pmc_traffic.txt

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB per dispatch.  Per /opt/skills/guides/MI355X_MICROARCH.md
(HBM / rocprofv3 section) the gfx950 FETCH_SIZE expression counts 128-byte requests as 64 bytes, so FETCH_SIZE is
doubled here; WRITE_SIZE is used as reported.  Values are averaged per launch (dispatch) of each kernel."""
import sqlite3
import sys


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    fetch, write, crops = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    names = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, (1, 0))[1] / fetch.get(k, (1, 0))[0]
                                                             + write.get(k, (1, 0))[1] / write.get(k, (1, 0))[0]))
    print(f"# HBM traffic per launch, KB (FETCH_SIZE x2 gfx950 correction; WRITE_SIZE as reported); {crops} crops per launch")
    tf = tw = 0.0
    for k in names:
        if k.startswith("__amd_rocclr"):
            continue
        nf, sf = fetch.get(k, (1, 0.0))
        nw, sw = write.get(k, (1, 0.0))
        f, w = 2.0 * sf / nf, sw / nw
        tf += f
        tw += w
        print(f"{k[:70]:70s} fetch {f:12.0f} KB ({f / crops:8.1f}/crop)  write {w:12.0f} KB ({w / crops:8.1f}/crop)")
    print(f"TOTAL per {crops}-crop launch set: fetch {tf / 1e6:.2f} GB, write {tw / 1e6:.2f} GB -> {(tf + tw) / crops:.0f} KB per crop")


if __name__ == "__main__":
    main()
