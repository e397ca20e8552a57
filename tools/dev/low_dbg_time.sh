# Developer tool (GPU box, tools/dev/dev.so = an NHW_DEV build): kernel times of the q <= 16 pre-filter under NHW_LOW_DBG switches.   usage: low_dbg_time.sh "<dbg values>" "<qualities>"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; cp nhwcodec_amd/libnhwhip.so /tmp/orig.so; cp tools/dev/dev.so nhwcodec_amd/libnhwhip.so
cd /tmp && export TMPDIR=/tmp
for d in $1; do for q in $2; do
  rm -rf /tmp/prof_x
  NHW_LOW_DBG=$d NHW_LOW_PARTS=${PARTS:-1} NHW_CHROMA_FORK=${FORK:-0} timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/tools/dev/gpu_q_timing.py $q > /tmp/x.log 2>&1
  echo "== dbg $d q$q: $(grep "^q$q" /tmp/x.log | cut -c1-40)"
  python $GRAFT_REPO_ROOT/profiles/summarise_rocpd.py $(ls /tmp/prof_x/*.db | head -1) 2>&1 | grep -i "k_low_chain\|k_low_apply\|k_low_marks\|k_low_markrows\|k_low_post" | cut -c1-120
done; done
cp /tmp/orig.so $GRAFT_REPO_ROOT/nhwcodec_amd/libnhwhip.so
