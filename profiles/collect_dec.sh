#!/bin/bash
# Run on the GPU box: HBM counters of the decode leg (FETCH_SIZE and WRITE_SIZE in their own passes), per kernel, and profiles/dec_pmc.json
# (k_dec_final's traffic per file, tied to nhw_dec.hip by a hash that bench.py checks).  usage: bash profiles/collect_dec.sh <tag> <commit>
set -u
TAG=${1:-decpmc}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT/pmc
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-chroma-l1 --no-config4-shape --sweep="
NHW_CHROMA_FORK=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc/f -o p --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
NHW_CHROMA_FORK=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc/w -o p --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
python profiles/pmc_summarise.py $OUT/pmc > $OUT/pmc.json 2>$OUT/pmc.err
python profiles/make_dec_pmc.py $OUT/pmc.json $COMMIT 20 4096 > $OUT/dec_pmc.json 2>>$OUT/pmc.err
rm -rf $OUT/pmc
cat $OUT/dec_pmc.json | head -60
