for p in 1 2 3 4; do
  NHW_PARTS=$p python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parts', $p, d['value'], d['ms_per_step'])"
done
