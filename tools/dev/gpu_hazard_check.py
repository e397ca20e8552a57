"""Developer tool (GPU box): find images of a synthetic batch whose chroma mark walk ends a row in a pair mark (the
running-index case of nhw_encoder.c:2372-2427) and compare exactly those against the oracle."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nhwcodec_amd
from oracle.oraclepy import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
e = nhwcodec_amd.Encoder(0, max_batch=n)
bgr = e.synth_device(n, seed_base=seed)
o, sizes, status = e.encode_device(bgr, q)
torch.cuda.synchronize()
sz = sizes.cpu().numpy()
ev = []
for i in range(n):
    m = np.zeros(32, np.int32)
    assert e.lib.nhw_debug_read(e.h, 54, i, ctypes.c_void_p(m.ctypes.data), ctypes.c_size_t(128)) == 0
    if m[31]:
        ev.append((i, int(m[31]) & 255, int(m[31]) >> 8))
print(f"{len(ev)} of {n} images have such rows (image, U rows, V rows):", ev[:20])
orc = Oracle()
bad = 0
for i, u, v in ev[:60]:
    want = orc.encode(orc.synth(seed + i), q)
    got = o[i, : sz[i]].cpu().numpy().tobytes()
    if got != want:
        bad += 1
        print(f"image {i} (U {u}, V {v}): MISMATCH ({len(got)} vs {len(want)} bytes)")
print(f"checked {min(len(ev), 60)} such images against the oracle: {bad} mismatches")
