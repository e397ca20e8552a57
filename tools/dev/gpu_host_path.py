"""Developer tool (GPU box): throughput of the host-buffer entry point nhw_enc_batch (H2D + encode + compaction + D2H included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nhwcodec_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
e = nhwcodec_amd.Encoder(0, max_batch=n)
imgs = e.synth_device(n, seed_base=1).cpu().numpy().reshape(n, 512, 512, 3)
e.encode(imgs[:64], 20)
for rep in range(3):
    t0 = time.perf_counter()
    out = e.encode(imgs, 20)
    dt = time.perf_counter() - t0
    print(f"host path: {n} images in {dt * 1e3:.1f} ms = {n * 0.262144 / dt:.0f} Mpixel/s ({sum(map(len, out)) / 1e6:.1f} MB out)", flush=True)
