# Developer tool (GPU box): the decode leg for variant builds (e.g. the row walks' prefetch depths: MK_AHEAD / SH_AHEAD in nhw_dec.hip, 2 / 4 / 8 measured: 4 stays) (tools/dev/<name>.so), interleaved.  usage: VARIANTS="a b" bash tools/dev/dec_ab.sh
cd $GRAFT_REPO_ROOT; cp nhwcodec_amd/libnhwhip.so /tmp/orig.so
for rep in 1 2; do for v in orig $VARIANTS; do
  if [ $v = orig ]; then cp /tmp/orig.so nhwcodec_amd/libnhwhip.so; else cp tools/dev/$v.so nhwcodec_amd/libnhwhip.so; fi
  echo "$v: $(timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-config4-shape --no-chroma-l1 --sweep= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)['decode']; print('decode ms', d['ms_per_step'])")"
done; done
cp /tmp/orig.so nhwcodec_amd/libnhwhip.so
