#!/bin/bash
# Run on the GPU box: kernel times of the decode leg (one stream and default), summaries under gpurun_out/<tag>/.  usage: bash profiles/quick_dec.sh <tag>
set -u
TAG=${1:-qdec}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-config4-shape --sweep="
NHW_CHROMA_FORK=0 rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o s -- $CMD > $OUT/stats1.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1
python profiles/summarise_rocpd.py $(ls $OUT/stats1/*.db | head -1) > $OUT/kernel_stats_1stream.txt 2>&1
python profiles/summarise_rocpd.py $(ls $OUT/stats/*.db | head -1) > $OUT/kernel_stats.txt 2>&1
grep -E "k_dec|fillBuffer" $OUT/kernel_stats_1stream.txt | cut -c1-130
rm -rf $OUT/stats1 $OUT/stats
