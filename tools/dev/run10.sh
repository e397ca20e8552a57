cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
(timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_encoder or robustness or many_seeds or residual_rules" > gpurun_out/r5q/pytest.log 2>&1); tail -3 gpurun_out/r5q/pytest.log
bash tools/dev/ab.sh tools/dev/old.so 17 19 14 16 > gpurun_out/r5q/ab.log 2>&1; cat gpurun_out/r5q/ab.log
