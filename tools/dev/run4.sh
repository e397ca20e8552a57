cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or golden or batch or stale or 4096" > gpurun_out/r5j/pytest.log 2>&1; echo rc=$? >> gpurun_out/r5j/pytest.log)
tail -5 gpurun_out/r5j/pytest.log
(timeout 600 bash tools/dev/ab.sh tools/dev/old.so 20 23 > gpurun_out/r5j/ab.log 2>&1); cat gpurun_out/r5j/ab.log
(timeout 600 bash profiles/quick.sh r5j_q20 20 > gpurun_out/r5j/quick20.log 2>&1); head -16 gpurun_out/r5j_q20/table.txt
cp nhwcodec_amd/libnhwhip.so /tmp/keep.so; cp tools/dev/prof.so nhwcodec_amd/libnhwhip.so
(timeout 300 python tests/gpu_pass_profile.py 1024 20 > gpurun_out/r5j/passes.log 2>&1); cat gpurun_out/r5j/passes.log
cp /tmp/keep.so nhwcodec_amd/libnhwhip.so
