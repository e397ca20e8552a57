cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
cp nhwcodec_amd/libnhwhip.so /tmp/base.so
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep= --quality"
for rep in 1 2; do
for v in base l4a_lw8 l4a_w8 l4a_lw8n; do
  if [ $v = base ]; then cp /tmp/base.so nhwcodec_amd/libnhwhip.so; else cp tools/dev/$v.so nhwcodec_amd/libnhwhip.so; fi
  for q in 20 23; do
    $B $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v q$q', round(d['ms_per_step'],3), d['images_ok'])"
  done
done
done > gpurun_out/r5m/ab.log 2>&1
cat gpurun_out/r5m/ab.log
for v in l4a_lw8 l4a_lw8n; do
  cp tools/dev/$v.so nhwcodec_amd/libnhwhip.so
  (timeout 300 bash profiles/quick.sh r5m_$v 20 > gpurun_out/r5m/quick_$v.log 2>&1); echo $v; grep "k_phase<3>\|total" gpurun_out/r5m_$v/table.txt
done
cp /tmp/base.so nhwcodec_amd/libnhwhip.so
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_encoder" > gpurun_out/r5m/pytest.log 2>&1); tail -2 gpurun_out/r5m/pytest.log
