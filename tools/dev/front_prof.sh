set -e
cd $GRAFT_REPO_ROOT
cp nhwcodec_amd/libnhwhip.so /tmp/orig.so
cp tools/dev/dev.so nhwcodec_amd/libnhwhip.so
mkdir -p gpurun_out
NHW_FRONT_PROF=1 timeout 300 python tools/dev/gpu_q_timing.py 20 > gpurun_out/front_prof.txt 2>&1 || true
timeout 600 python tools/dev/gpu_band_ablate.py 20 > gpurun_out/front_ablate.txt 2>&1 || true
cp /tmp/orig.so nhwcodec_amd/libnhwhip.so
tail -40 gpurun_out/front_prof.txt; cat gpurun_out/front_ablate.txt
