cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
(timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_encoder or stages_match or robustness or many_seeds" > gpurun_out/r5q/pytest.log 2>&1); tail -3 gpurun_out/r5q/pytest.log
bash tools/dev/ab.sh tools/dev/old.so 20 23 > gpurun_out/r5q/ab.log 2>&1; cat gpurun_out/r5q/ab.log
(timeout 300 bash profiles/quick.sh r5q_t 20 > gpurun_out/r5q/quick.log 2>&1); grep -i "k_phase<0>\|total" gpurun_out/r5q_t/table.txt
