cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
(timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_encoder or low_quality_stages or robustness or many_seeds or compat or golden" > gpurun_out/r5q/pytest.log 2>&1); tail -3 gpurun_out/r5q/pytest.log
bash tools/dev/ab.sh tools/dev/old.so 1 10 8 12 > gpurun_out/r5q/ab.log 2>&1; cat gpurun_out/r5q/ab.log
bash profiles/quick.sh r5ll2 1 > /dev/null 2>&1; grep "k_low_ll2\|total" gpurun_out/r5ll2/table.txt
