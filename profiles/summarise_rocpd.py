"""Turns a rocprofv3 results database (rocpd sqlite; `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`)
into the per-kernel text summary committed under profiles/.  usage: summarise_rocpd.py results.db > out.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                 "max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# kernels: {len(rows)}  total GPU kernel time: {tot / 1e6:.3f} ms")
print("%-78s %6s %12s %11s %11s %11s %6s %5s %5s %7s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%", "vgpr", "sgpr", "lds", "scratch"))
for r in rows:
    print("%-78s %6d %12.3f %11.4f %11.4f %11.4f %6.2f %5d %5d %7d %8d" % (r[0][:78], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))
