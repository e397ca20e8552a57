#!/bin/bash
# Run on the GPU box: the LDS counters of the front kernel for two builds of the library (tools/dev/old.so and the tree's), one --pmc pass each.
# usage: bash profiles/collect_front_lds.sh <tag>      -> gpurun_out/<tag>/front_lds_{old,new}.json
set -u
TAG=${1:-flds}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python tools/dev/gpu_q_timing.py 20"
cp nhwcodec_amd/libnhwhip.so /tmp/new.so
for v in old new; do
	if [ $v = old ]; then cp tools/dev/old.so nhwcodec_amd/libnhwhip.so; else cp /tmp/new.so nhwcodec_amd/libnhwhip.so; fi
	rm -rf $OUT/pmc_$v
	rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_$v/g1 -o p --output-format csv -- $CMD > $OUT/pmc_$v.log 2>&1
	rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_$v/g2 -o p --output-format csv -- $CMD >> $OUT/pmc_$v.log 2>&1
	python profiles/pmc_summarise.py $OUT/pmc_$v | python -c "
import json,sys
d=json.load(sys.stdin)
print(json.dumps({k:{c:round(v['per_launch']) for c,v in d[k].items()} for k in d if 'k_front' in k or 'k_dwt_ana' in k}, indent=1))" > $OUT/front_lds_$v.json
	rm -rf $OUT/pmc_$v
done
cp /tmp/new.so nhwcodec_amd/libnhwhip.so
cat $OUT/front_lds_old.json $OUT/front_lds_new.json
