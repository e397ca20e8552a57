cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filterbank or level1 or host_path or fused_front or two_devices or color or repeated or 4096" > gpurun_out/r5a/pytest_sel.log 2>&1; echo rc=$? >> gpurun_out/r5a/pytest_sel.log)
tail -5 gpurun_out/r5a/pytest_sel.log
(timeout 600 bash tools/dev/ab.sh tools/dev/old.so 20 17 > gpurun_out/r5a/ab.log 2>&1); cat gpurun_out/r5a/ab.log
(timeout 600 bash profiles/collect_front_lds.sh r5a > gpurun_out/r5a/lds.log 2>&1); tail -40 gpurun_out/r5a/lds.log
