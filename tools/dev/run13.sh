cd $GRAFT_REPO_ROOT
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep= --quality"
for rep in 1 2 3; do
for v in none high low; do
  for q in 20 23; do
    NHW_SIDE_PRIO=$v $B $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v q$q', round(d['ms_per_step'],3), d['images_ok'])"
  done
done
done
