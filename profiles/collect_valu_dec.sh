#!/bin/bash
# Run on the GPU box: issue / stall counters of the decode kernels (one --pmc pass per group); summary in gpurun_out/decvalu/pmc_valu.json.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/decvalu; mkdir -p $OUT/pmc
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-chroma-l1 --no-config4-shape --sweep="
i=0
for grp in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
	i=$((i+1))
	NHW_CHROMA_FORK=0 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc/g$i -o p --output-format csv -- $CMD > $OUT/pmc_g$i.log 2>&1
done
python profiles/pmc_summarise.py $OUT/pmc > $OUT/pmc_valu.json 2>$OUT/pmc.err
rm -rf $OUT/pmc
