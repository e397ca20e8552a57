"""Developer tool (GPU box): decode throughput at batch N (files produced by the GPU encoder), hipEvent-timed."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nhwcodec_amd as na

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    enc = na.Encoder(0, n)
    img = enc.synth_device(n, 0)
    out, sizes, status = enc.encode_device(img, q)
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0
    offs = torch.arange(n, dtype=torch.int64, device="cuda") * na.OUT_STRIDE
    total = int(sizes.sum())
    blob = out
    enc.close(); del img
    torch.cuda.empty_cache()
    dec = na.Decoder(0, n)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    pix = torch.empty((n, 512, 512, 3), dtype=torch.uint8, device="cuda")
    dec.decode_device(blob, offs, sizes, pix)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        _, st, qq = dec.decode_device(blob, offs, sizes, pix)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    assert int(st.abs().sum()) == 0 and int((qq != q).sum()) == 0
    print(f"decode n={n} q={q}: {ms:.2f} ms/batch, {n * 0.262144 / ms * 1000:.0f} Mpix/s, input {total/1e6:.1f} MB")

if __name__ == "__main__":
    main()
