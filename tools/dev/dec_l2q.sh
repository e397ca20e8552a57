# Developer tool (GPU box): decoder tests, then the decode leg of the bench with and without the quarter form of the luma level-2 kernel (tools/dev/dec_l2_quarters_experiment.patch applied).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_decode.py -x -q -m gpu > gpurun_out/dec_tests.txt 2>&1; tail -1 gpurun_out/dec_tests.txt
for v in 1 1; do echo "== NHW_DEC_L2Q=$v $(NHW_DEC_L2Q=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-config4-shape --no-chroma-l1 --sweep= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)['decode']; print('decode ms', d['ms_per_step'])")"; done
