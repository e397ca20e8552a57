cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
cp nhwcodec_amd/libnhwhip.so /tmp/base.so
for v in sl8 sl16; do
  cp tools/dev/$v.so nhwcodec_amd/libnhwhip.so
  (timeout 300 bash profiles/quick.sh r5n_$v 20 > gpurun_out/r5n/quick_$v.log 2>&1); echo $v; grep "k_phase<12>\|k_final\|total" gpurun_out/r5n_$v/table.txt
done
cp /tmp/base.so nhwcodec_amd/libnhwhip.so
