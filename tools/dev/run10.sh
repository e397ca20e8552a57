cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_encoder or many_seeds or robustness or compat_mode or every_image or stream_modes or stages_match or repeated" > gpurun_out/r5q/pytest.log 2>&1); tail -5 gpurun_out/r5q/pytest.log
bash tools/dev/ab.sh tools/dev/old.so 20 23 17 > gpurun_out/r5q/ab.log 2>&1; cat gpurun_out/r5q/ab.log
for q in 20 23; do NHW_LL_FORK=0 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-host-path --no-chroma-l1 --no-config4-shape --sweep= --quality $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('noll q$q', round(d['ms_per_step'],3))"; done
