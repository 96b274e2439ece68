"""Per-kernel SQ counters from a rocprofv3 --pmc pass (rocpd sqlite): averages per dispatch and the share of wave time
spent parked / issue-stalled / issuing (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).
Usage: python profiles/summarize_sq.py <pass.db>"""
import sqlite3
import sys
from collections import defaultdict


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    per = defaultdict(dict)
    for k, c, n, s in rows:
        per[k][c] = s / n
    for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if k.startswith("__amd_rocclr"):
            continue
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        print(k[:78])
        for c, v in sorted(d.items()):
            extra = f"  {100 * v / wc:5.1f}% of wave cycles" if c.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
            print(f"    {c:24s} {v:16.0f}{extra}")


if __name__ == "__main__":
    main()
