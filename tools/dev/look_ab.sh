# Developer tool (GPU box): front kernel time (hipEvents, mean of 30 batches) for builds with different look-back lengths, interleaved twice.  The builds:
# bash tools/dev/build_variant.sh look16b -DFI_LOOK2=32 ; ... look12b "-DFI_LOOK=12 -DFI_LOOK2=32" ; -DFI_TIMING_NO_SERIAL: what the serial replay still costs (nothing measurable).
cd $GRAFT_REPO_ROOT; cp nhwcodec_amd/libnhwhip.so /tmp/orig.so
cat > /tmp/ft.py <<'P'
import sys, torch
sys.path.insert(0, ".")
import nhwcodec_amd as na
enc = na.Encoder(0, 4096); img = enc.synth_device(4096, 0); out = enc.alloc_out(4096)
for q in (20, 17, 19):
    for _ in range(3): enc.encode_device(img, q, out)
    torch.cuda.synchronize(); acc = 0.0
    for _ in range(30):
        enc.encode_device(img, q, out); torch.cuda.synchronize(); acc += enc.timing().front_ms
    print(f"q{q} front {acc / 30:.3f}", end="  ")
print()
P
for rep in 1 2; do for v in orig look16b look12b; do
  if [ $v = orig ]; then cp /tmp/orig.so nhwcodec_amd/libnhwhip.so; else cp tools/dev/$v.so nhwcodec_amd/libnhwhip.so; fi
  echo "$v: $(timeout 200 python /tmp/ft.py 2>&1 | tail -1)"
done; done
cp /tmp/orig.so nhwcodec_amd/libnhwhip.so
