#!/bin/bash
# Developer tool (GPU box): start and end of every kernel of one encode step in the default multi-stream run (profiles/timeline_rocpd.py).  usage: bash tools/dev/timeline.sh [quality]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q=${1:-20}
OUT=gpurun_out/r5tl; mkdir -p $OUT; rm -rf $OUT/st
rocprofv3 --kernel-trace --stats -d $OUT/st -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep= --quality $Q > $OUT/log.txt 2>&1
python profiles/timeline_rocpd.py $(ls $OUT/st/*.db | head -1) > $OUT/timeline_q$Q.txt 2>&1
rm -rf $OUT/st
cat $OUT/timeline_q$Q.txt
