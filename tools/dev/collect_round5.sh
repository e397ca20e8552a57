#!/bin/bash
# Run on the GPU box: everything the round's documents cite, in one go.  usage: bash tools/dev/collect_round5.sh <commit>
C=${1:-unknown}
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh round5 $C > gpurun_out/collect.log 2>&1
cp gpurun_out/round5/front_pmc.json profiles/front_pmc.json
bash profiles/collect_valu.sh round5v >> gpurun_out/collect.log 2>&1
cp gpurun_out/round5v/pmc_valu.json profiles/round5_pmc_valu.json
for q in 1 8 10 23; do bash profiles/quick.sh round5_q$q $q >> gpurun_out/collect.log 2>&1; done
bash profiles/collect_dec.sh round5dec $C >> gpurun_out/collect.log 2>&1
cp gpurun_out/round5dec/dec_pmc.json profiles/dec_pmc.json
bash profiles/quick_dec.sh round5_dec >> gpurun_out/collect.log 2>&1
bash profiles/collect_valu_dec.sh >> gpurun_out/collect.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/round5_bench.json 2> gpurun_out/round5_bench.err
tail -c 600 gpurun_out/round5_bench.json
(timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/round5_pytest.log 2>&1); tail -3 gpurun_out/round5_pytest.log
