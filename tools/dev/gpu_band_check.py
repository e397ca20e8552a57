"""Developer tool (GPU box): the compact band plane of Y29 (q >= 22) against a plain walk over the oracle's quantised plane."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nhwcodec_amd
from oracle.oraclepy import Oracle

def big_index(a):
    # nhw_tail_dev.h big_index: index k with |k|*8+... ; restated from band_value's use
    raise NotImplementedError

def main(q=22, seeds=(7, 0)):
    import torch
    orc = Oracle()
    enc = nhwcodec_amd.Encoder(0, max_batch=len(seeds))
    imgs = np.stack([orc.synth(s) for s in seeds])
    enc.lib.nhw_debug_stop_after(enc.h, 12)      # the luma tail up to Y31; the chroma phases reuse the band plane
    enc.encode_device(torch.from_numpy(imgs).cuda(), q)
    torch.cuda.synchronize()
    for i, s in enumerate(seeds):
        tr = orc.encode(imgs[i], q, trace=True)[1]
        proc = np.frombuffer([b for n, b in tr if n == "offsetY"][0][0], np.int16).reshape(512, 512)
        band = np.zeros(65536 + 1024, np.int32); written = np.zeros(65536 + 1024, bool)
        t = 0
        ends = [0] * 256
        for r in range(256):
            skip = False
            for j in range(256):
                a = int(proc[r, 256 + j])
                if skip:
                    skip = False; t += 1 if False else 0
                    continue
                if a == 128:
                    t += 1; continue
                if a in (127, 129):
                    e = 5 if a == 127 else -5
                    band[t - 1] = e; band[t] = 6 if a == 127 else -7; band[t + 1] = e; written[t-1:t+2] = True
                    t += 2; skip = True
                    if j == 255: ends[r] = 1
                    continue
                band[t] = -99999; written[t] = True   # plain code: value not restated here
                t += 1
        out = np.empty(2 * 65536, np.uint8)
        assert enc.lib.nhw_debug_read(enc.h, 12, i, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(out.size)) == 0
        got = out.view(np.int16).astype(np.int32)
        known = written[:65536] & (band[:65536] != -99999)
        bad = np.nonzero(known & (got != band[:65536]))[0]
        unw = np.nonzero(~written[:65536] & (got != 0))[0]
        print(f"seed {s}: mark-slot mismatches {len(bad)} first {bad[:5].tolist()} (row {bad[0] // 256 if len(bad) else '-'}); slots the walk never writes but are non-zero: {len(unw)} {unw[:5].tolist()}")
        for x in list(bad[:3]) + list(unw[:3]):
            rr, cc = x // 256, x % 256
            print("   slot", x, "row", rr, "col", cc, "gpu", got[x - 4:x + 5].tolist(), "want", band[x - 4:x + 5].tolist(), "cells", proc[rr, 256 + max(0, cc - 6):256 + cc + 6].tolist(), "end marks above", sum(1 for r2 in range(rr) if ends[r2]))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 22)
