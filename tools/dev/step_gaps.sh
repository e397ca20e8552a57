# Developer tool (GPU box): the idle time between consecutive encode steps (end of a step's last kernel -> start of the next step's first), from a kernel trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gaps; mkdir -p $OUT; rm -rf $OUT/st
rocprofv3 --kernel-trace -d $OUT/st -o s -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep= > $OUT/log.txt 2>&1
python - $(ls $OUT/st/*.db | head -1) <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
prev_end = None; last = None
for i, (name, s, e) in enumerate(rows):
    if "k_front_image" in name and prev_end is not None:
        print(f"front starts {(s - prev_end) / 1e3:8.1f} us after the end of {last[:40]}; the kernels just before it:", [ (r[0][:22], round((r[2]-r[1])/1e3,1)) for r in rows[max(0,i-3):i] ])
    if prev_end is None or e > prev_end: prev_end = e; last = name
P
rm -rf $OUT/st
