# Developer tool (GPU box): k_low_pre's time with and without its serial row replays (tools/dev/preNS.so = -DPRE_TIMING_NO_SERIAL) and other variants.
cd $GRAFT_REPO_ROOT; cp nhwcodec_amd/libnhwhip.so /tmp/orig.so
cd /tmp && export TMPDIR=/tmp
for v in orig ${VARIANTS:-preNS}; do
  if [ $v = orig ]; then cp /tmp/orig.so $GRAFT_REPO_ROOT/nhwcodec_amd/libnhwhip.so; else cp $GRAFT_REPO_ROOT/tools/dev/$v.so $GRAFT_REPO_ROOT/nhwcodec_amd/libnhwhip.so; fi
  for q in ${QS:-10 1}; do
    rm -rf /tmp/prof_x
    NHW_LOW_PARTS=1 NHW_CHROMA_FORK=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/tools/dev/gpu_q_timing.py $q > /tmp/x.log 2>&1
    echo "== $v q$q: $(python $GRAFT_REPO_ROOT/profiles/summarise_rocpd.py $(ls /tmp/prof_x/*.db | head -1) 2>&1 | grep -i "k_low_pre" | cut -c1-120)"
  done
done
cp /tmp/orig.so $GRAFT_REPO_ROOT/nhwcodec_amd/libnhwhip.so
