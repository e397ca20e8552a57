#!/bin/bash
# Developer tool (GPU box): the decode leg with two builds of the library on the same box, alternating.  usage: bash tools/dev/ab_dec.sh <old.so> [q]
OLD=$1; Q=${2:-20}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-chroma-l1 --no-config4-shape --sweep= --quality $Q"
cp nhwcodec_amd/libnhwhip.so /tmp/new.so
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp $OLD nhwcodec_amd/libnhwhip.so; else cp /tmp/new.so nhwcodec_amd/libnhwhip.so; fi
    $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); x=d['decode']; print('$v q$Q decode', x['ms_per_step'], 'final', x['roofline']['ms_per_launch'], x['roofline']['frac'], 'enc', round(d['ms_per_step'],3))"
  done
done
cp /tmp/new.so nhwcodec_amd/libnhwhip.so
