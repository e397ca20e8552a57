cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
(timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5k/pytest.log 2>&1; echo rc=$? >> gpurun_out/r5k/pytest.log)
tail -15 gpurun_out/r5k/pytest.log
