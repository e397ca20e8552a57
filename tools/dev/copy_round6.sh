#!/bin/bash
# Developer tool (container): what tools/dev/collect_round6.sh merged into gpurun_out/ -> the files under profiles/ the documents cite; then the kernel table of DESIGN.md.
set -e
cd "$(dirname "$0")/../.."
cp gpurun_out/round6/kernel_stats.txt profiles/round6_kernel_stats.txt
cp gpurun_out/round6/kernel_stats_1stream.txt profiles/round6_kernel_stats_1stream.txt
cp gpurun_out/round6/pmc.json profiles/round6_pmc.json
cp gpurun_out/round6/front_pmc.json profiles/front_pmc.json
cp gpurun_out/round6v/pmc_valu.json profiles/round6_pmc_valu.json
for q in 1 8 10 23; do cp gpurun_out/round6_q$q/table.txt profiles/round6_q${q}_kernel_table.txt; done
cp gpurun_out/round6_dec/kernel_stats.txt profiles/round6_decode_kernel_stats.txt
cp gpurun_out/round6_dec/kernel_stats_1stream.txt profiles/round6_decode_kernel_stats_1stream.txt
cp gpurun_out/round6dec/pmc.json profiles/round6_decode_pmc.json
cp gpurun_out/round6dec/dec_pmc.json profiles/dec_pmc.json
cp gpurun_out/decvalu/pmc_valu.json profiles/round6_decode_pmc_valu.json
cp gpurun_out/round6_bench.json profiles/round6_bench.json
python profiles/make_design_table.py
python profiles/make_design_table.py --check
