cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_encoder or many_seeds or robustness or compat_mode or every_image" > gpurun_out/r5q/pytest.log 2>&1); tail -5 gpurun_out/r5q/pytest.log
