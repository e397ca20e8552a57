cd $GRAFT_REPO_ROOT
for q in 1 8 10; do bash profiles/quick.sh round5_q$q $q > /dev/null 2>&1; tail -1 gpurun_out/round5_q$q/table.txt; done
python bench.py --steps 20 --warmup 5 > gpurun_out/round5_bench.json 2> gpurun_out/round5_bench.err
tail -c 300 gpurun_out/round5_bench.json
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "low_quality or whole_encoder or golden or every_image" > gpurun_out/round5_pytest_low.log 2>&1); tail -2 gpurun_out/round5_pytest_low.log
