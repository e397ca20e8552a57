#!/bin/bash
# Developer tool (GPU box): the same for one decode step.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5tl; mkdir -p $OUT; rm -rf $OUT/st
rocprofv3 --kernel-trace --stats -d $OUT/st -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-chroma-l1 --no-config4-shape --sweep= > $OUT/logd.txt 2>&1
python profiles/timeline_rocpd.py $(ls $OUT/st/*.db | head -1) k_dec_parse > $OUT/timeline_dec.txt 2>&1
rm -rf $OUT/st
head -30 $OUT/timeline_dec.txt
