cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
(timeout 1200 python -m pytest tests/test_decode.py -x -q -m gpu > gpurun_out/r5f/pytest_dec.log 2>&1; echo rc=$? >> gpurun_out/r5f/pytest_dec.log)
tail -8 gpurun_out/r5f/pytest_dec.log
(timeout 600 bash tools/dev/ab_dec.sh tools/dev/old.so 20 > gpurun_out/r5f/ab_dec.log 2>&1); cat gpurun_out/r5f/ab_dec.log
(timeout 600 bash profiles/quick_dec.sh r5f_dec > gpurun_out/r5f/quick_dec.log 2>&1); tail -25 gpurun_out/r5f/quick_dec.log
