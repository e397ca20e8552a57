#!/bin/bash
# Run on the GPU box: one-stream kernel times and HBM bytes per kernel of one encode bench command, as one table.
# usage: bash profiles/quick.sh <tag> [quality]
set -u
TAG=${1:-quick}; Q=${2:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep= --quality $Q"
NHW_CHROMA_FORK=0 NHW_LOW_PARTS=1 rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o s -- $CMD > $OUT/stats1.log 2>&1
NHW_CHROMA_FORK=0 NHW_LOW_PARTS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc/f -o p --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
NHW_CHROMA_FORK=0 NHW_LOW_PARTS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc/w -o p --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
python profiles/summarise_rocpd.py $(ls $OUT/stats1/*.db | head -1) > $OUT/kernel_stats_1stream.txt 2>&1
python profiles/pmc_summarise.py $OUT/pmc > $OUT/pmc.json 2>$OUT/pmc.err
python profiles/quick_table.py $OUT/kernel_stats_1stream.txt $OUT/pmc.json 4 | tee $OUT/table.txt
rm -rf $OUT/pmc $OUT/stats1
