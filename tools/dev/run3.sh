cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or golden or batch or stale or 4096" > gpurun_out/r5c/pytest.log 2>&1; echo rc=$? >> gpurun_out/r5c/pytest.log)
tail -5 gpurun_out/r5c/pytest.log
(timeout 600 bash tools/dev/ab.sh tools/dev/old.so 20 23 1 > gpurun_out/r5c/ab.log 2>&1); cat gpurun_out/r5c/ab.log
(timeout 600 bash profiles/quick.sh r5c_q20 20 > gpurun_out/r5c/quick20.log 2>&1); cat gpurun_out/r5c_q20/table.txt
