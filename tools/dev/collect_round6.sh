#!/bin/bash
# Run on the GPU box: everything the round's documents cite, in one go.  usage: bash tools/dev/collect_round6.sh <commit>
C=${1:-unknown}
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh round6 $C > gpurun_out/collect.log 2>&1
cp gpurun_out/round6/front_pmc.json profiles/front_pmc.json
bash profiles/collect_valu.sh round6v >> gpurun_out/collect.log 2>&1
cp gpurun_out/round6v/pmc_valu.json profiles/round6_pmc_valu.json
for q in 1 8 10 23; do bash profiles/quick.sh round6_q$q $q >> gpurun_out/collect.log 2>&1; done
bash profiles/collect_dec.sh round6dec $C >> gpurun_out/collect.log 2>&1
cp gpurun_out/round6dec/dec_pmc.json profiles/dec_pmc.json
bash profiles/quick_dec.sh round6_dec >> gpurun_out/collect.log 2>&1
bash profiles/collect_valu_dec.sh >> gpurun_out/collect.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/round6_bench.json 2> gpurun_out/round6_bench.err
tail -c 600 gpurun_out/round6_bench.json
(timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/round6_pytest.log 2>&1); tail -3 gpurun_out/round6_pytest.log
