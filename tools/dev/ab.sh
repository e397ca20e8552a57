#!/bin/bash
# Developer tool (GPU box): the encode step with two builds of the library on the same box, alternating.  usage: bash tools/dev/ab.sh <old.so> [q ...]
OLD=$1; shift; QS=${@:-20}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep="
cp nhwcodec_amd/libnhwhip.so /tmp/new.so
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp $OLD nhwcodec_amd/libnhwhip.so; else cp /tmp/new.so nhwcodec_amd/libnhwhip.so; fi
    for q in $QS; do
      $B --quality $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v q$q', round(d['ms_per_step'],3), r.get('frac'))"
    done
  done
done
cp /tmp/new.so nhwcodec_amd/libnhwhip.so
