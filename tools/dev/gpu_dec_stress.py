"""Developer tool (GPU box): decode one resident batch many times, report files whose pixels ever change.
usage: python tools/dev/gpu_dec_stress.py [n] [q] [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nhwcodec_amd as na

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
enc = na.Encoder(0, n); img = enc.synth_device(n, 5); out, sizes, status = enc.encode_device(img, q); torch.cuda.synchronize(); enc.close(); del img
dec = na.Decoder(0, n)
offs = torch.arange(n, dtype=torch.int64, device="cuda") * na.OUT_STRIDE
first, bad = None, 0
for r in range(reps):
    px, st, _ = dec.decode_device(out, offs, sizes); torch.cuda.synchronize()
    if first is None: first = px.clone(); continue
    d = (px != first).flatten(1).any(dim=1)
    if bool(d.any()): bad += 1; print("repeat", r, "changed files", d.nonzero().flatten().tolist()[:10])
print(f"n={n} q={q} repeats={reps}: runs that differed: {bad}")
