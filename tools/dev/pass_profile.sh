#!/bin/bash
# Developer tool (GPU box): per-pass clocks inside the tail kernels (tools/dev/prof.so from build_prof.sh; tools/dev/gpu_pass_profile.py).  usage: bash tools/dev/pass_profile.sh [quality]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
cp nhwcodec_amd/libnhwhip.so /tmp/base.so
cp tools/dev/prof.so nhwcodec_amd/libnhwhip.so
python tools/dev/gpu_pass_profile.py 4096 ${1:-20} > gpurun_out/r5q/prof.log 2>&1
cp /tmp/base.so nhwcodec_amd/libnhwhip.so
cat gpurun_out/r5q/prof.log
