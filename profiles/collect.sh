#!/bin/bash
# Run on the GPU box (gpurun): kernel stats (two streams and one stream) and the HBM counters of one bench command, summaries under
# gpurun_out/<tag>/.  usage: bash profiles/collect.sh <tag> <commit>      (counters in their own passes: FETCH_SIZE and WRITE_SIZE do not fit one)
set -u
TAG=${1:-r2}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep="
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1
NHW_CHROMA_FORK=0 NHW_LOW_PARTS=1 rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o s -- $CMD > $OUT/stats1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
python profiles/summarise_rocpd.py $(ls $OUT/stats/*.db | head -1) > $OUT/kernel_stats.txt 2>&1
python profiles/summarise_rocpd.py $(ls $OUT/stats1/*.db | head -1) > $OUT/kernel_stats_1stream.txt 2>&1
mkdir -p $OUT/pmc && cp -r $OUT/pmc_fetch $OUT/pmc/f && cp -r $OUT/pmc_write $OUT/pmc/w
python profiles/pmc_summarise.py $OUT/pmc > $OUT/pmc.json 2>$OUT/pmc.err
python profiles/make_front_pmc.py $OUT/pmc.json $COMMIT 20 4096 > $OUT/front_pmc.json 2>>$OUT/pmc.err
rm -rf $OUT/pmc $OUT/stats $OUT/stats1 $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
