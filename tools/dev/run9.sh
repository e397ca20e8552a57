cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
cp nhwcodec_amd/libnhwhip.so /tmp/base.so
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep= --quality"
for rep in 1 2; do
for v in base c5w5 c5w6; do
  if [ $v = base ]; then cp /tmp/base.so nhwcodec_amd/libnhwhip.so; else cp tools/dev/$v.so nhwcodec_amd/libnhwhip.so; fi
  for q in 20 23; do
    $B $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v q$q', round(d['ms_per_step'],3), d['images_ok'])"
  done
done
done > gpurun_out/r5p/ab.log 2>&1
cat gpurun_out/r5p/ab.log
for v in c5w5 c5w6; do
  cp tools/dev/$v.so nhwcodec_amd/libnhwhip.so
  (timeout 300 bash profiles/quick.sh r5p_$v 20 > gpurun_out/r5p/quick_$v.log 2>&1); echo $v; grep "k_c5\|total" gpurun_out/r5p_$v/table.txt
done
cp /tmp/base.so nhwcodec_amd/libnhwhip.so
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_encoder or symbol_list" > gpurun_out/r5p/pytest.log 2>&1); tail -2 gpurun_out/r5p/pytest.log
