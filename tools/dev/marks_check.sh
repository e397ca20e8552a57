# Developer tool (GPU box): the q <= 16 pre-filter after a change to its kernels -- parity of the rationed qualities, then kernel times at q1 / q8 / q10.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "many_seeds or rationed or fallback_paths or mixed_class" ${SKIPTESTS:+--co} > gpurun_out/marks_tests.txt 2>&1; tail -3 gpurun_out/marks_tests.txt
cd /tmp && export TMPDIR=/tmp
for q in 8 10 1; do
  rm -rf /tmp/prof_q$q
  NHW_LOW_PARTS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_q$q -o p -- python $GRAFT_REPO_ROOT/tools/dev/gpu_q_timing.py $q > $GRAFT_REPO_ROOT/gpurun_out/marks_q$q.log 2>&1
  echo "== q$q"; grep "^q$q" $GRAFT_REPO_ROOT/gpurun_out/marks_q$q.log | cut -c1-60
  python $GRAFT_REPO_ROOT/profiles/summarise_rocpd.py $(ls /tmp/prof_q$q/*.db | head -1) 2>&1 | grep -i "k_low\|k_color" | cut -c1-130
done
