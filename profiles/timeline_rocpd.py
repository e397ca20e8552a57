"""Timeline of the LAST encode step of a rocprofv3 kernel trace (rocpd sqlite): every kernel with its stream / queue, start and end relative to the
step's first kernel -- what overlaps what in the default three-stream run.  usage: timeline_rocpd.py results.db [first-kernel-substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_front"
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if first in r[0]]
if not starts:
    sys.exit("no kernel matches " + first)
b = starts[-1]
t0 = rows[b][1]
print("%-44s %6s %9s %9s %8s" % ("kernel", "queue", "start_ms", "end_ms", "ms"))
for name, s, e, qq in rows[b:]:
    print("%-44s %6s %9.3f %9.3f %8.3f" % (name[:44], str(qq)[-6:], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
