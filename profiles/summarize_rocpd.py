"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
Usage: python profiles/summarize_rocpd.py gpurun_out/<dir>/<name>_results.db > profiles/<round>_<tag>_kernel_stats.txt
(rocprofv3 --kernel-trace --stats in this image writes a rocpd database, not CSV)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute(
    "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
    "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
    "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'%':>6s} "
      f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s}")
for r in rows:
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:10.3f} {r[3]:10.1f} {r[4]:9.1f} {r[5]:10.1f} {100 * r[2] / tot:6.1f} "
          f"{r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:7d}")
print(f"total kernel time {tot:.3f} ms")
