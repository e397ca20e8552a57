#!/bin/bash
# Run on the GPU box: instruction-fetch, scalar-unit and LDS-wait counters of the q <= 16 pre-filter kernels (k_low_pre, k_low_mapfix, k_low_chain, k_low_apply, k_low_markrows, k_low_marks), one
# --pmc pass per group over `python tools/dev/gpu_prefilter_time.py <q>`.  usage: bash profiles/collect_low.sh <tag> [q ...]
set -u
TAG=${1:-low}; shift; QS=${@:-10 8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for q in $QS; do
	OUT=gpurun_out/$TAG/q$q; mkdir -p $OUT/pmc
	i=0
	for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQC_DCACHE_REQ" "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES"; do
		i=$((i+1))
		rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc/g$i -o p --output-format csv -- python tools/dev/gpu_prefilter_time.py $q > $OUT/pmc_g$i.log 2>&1
	done
	python profiles/pmc_summarise.py $OUT/pmc > $OUT/pmc_low.json 2>$OUT/pmc.err
	rm -rf $OUT/pmc
done
