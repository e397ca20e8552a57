# Developer tool (GPU box): parity of the qualities that use the chroma level-1 quarter kernel, then the bench line with and without it.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "many_seeds or golden or mixed_class or shifted_chroma or pair_mark" > gpurun_out/l1q_tests.txt 2>&1; tail -3 gpurun_out/l1q_tests.txt
for v in 1 0; do echo "== NHW_CHROMA_L1Q=$v"; NHW_CHROMA_L1Q=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-host-path --no-config4-shape --sweep= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('ms', d['ms_per_step'], 'frac', r['frac'], 'front', r.get('ms_per_launch_group'), 'chroma_l1_ms', r.get('chroma_l1_ms'))"; done
