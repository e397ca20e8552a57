"""Developer tool (GPU box): ms per 4096-image batch for a list of quality settings (inputs and outputs resident in HBM).
usage: python tools/dev/gpu_q_timing.py [q ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nhwcodec_amd as na

def main(qs, n=4096):
    enc = na.Encoder(0, n)
    img = enc.synth_device(n, 0)
    out = enc.alloc_out(n)
    for q in qs:
        for _ in range(2): enc.encode_device(img, q, out)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5): enc.encode_device(img, q, out)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 5 * 1e3
        t = enc.timing()
        print(f"q{q}: {ms:.1f} ms / batch = {n * 0.262144 / ms:.1f} Gpixel/s  stages", {k: round(getattr(t, k), 2) for k, _ in t._fields_}, flush=True)

if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [1, 10, 20, 23])
