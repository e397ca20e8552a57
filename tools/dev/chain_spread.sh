cd $GRAFT_REPO_ROOT; cp nhwcodec_amd/libnhwhip.so /tmp/orig.so; cp tools/dev/dev.so nhwcodec_amd/libnhwhip.so
timeout 600 python tools/dev/gpu_chain_spread.py 10 8 1 13 16 2>&1 | tail -20
cp /tmp/orig.so nhwcodec_amd/libnhwhip.so
