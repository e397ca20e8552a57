cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
(timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5i/pytest.log 2>&1; echo rc=$? >> gpurun_out/r5i/pytest.log)
tail -15 gpurun_out/r5i/pytest.log
(timeout 600 bash tools/dev/ab.sh tools/dev/old.so 20 23 > gpurun_out/r5i/ab.log 2>&1); cat gpurun_out/r5i/ab.log
