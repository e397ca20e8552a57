cd $GRAFT_REPO_ROOT
cp nhwcodec_amd/libnhwhip.so /tmp/base.so
for k in 1 2 3 4; do
  cp tools/dev/ll2s$k.so nhwcodec_amd/libnhwhip.so
  (timeout 300 bash profiles/quick.sh r5ll2_$k 1 > /dev/null 2>&1); echo stop$k; grep "k_low_ll2" gpurun_out/r5ll2_$k/table.txt
done
cp /tmp/base.so nhwcodec_amd/libnhwhip.so
