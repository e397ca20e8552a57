cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 4 0 1 2 3 5; do
  NHW_DEC_FORK=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-chroma-l1 --no-config4-shape --sweep= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fork $v', d['decode']['ms_per_step'], d['decode']['files_ok_rank0'])"
done
done
