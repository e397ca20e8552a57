"""Developer tool (GPU box): the q <= 16 luma pre-filter as a stage on a full batch -- ms per batch and a check of some images against the oracle.
usage: python tools/dev/gpu_prefilter_time.py [q ...]   (NHW_N=<images>; NHW_LOW_DBG acts on developer builds)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nhwcodec_amd as na
from oracle.oraclepy import Oracle

def main(qs, n=4096, check=6):
    orc = Oracle()
    enc = na.Encoder(0, n)
    img = enc.synth_device(n, 0)
    y = torch.empty((n, 512 * 512), dtype=torch.int16, device="cuda")
    u = torch.empty((n, 65536), dtype=torch.uint8, device="cuda"); v = torch.empty_like(u)
    for q in qs:
        assert enc.lib.nhw_stage_color(enc.h, img.data_ptr(), n, q, y.data_ptr(), u.data_ptr(), v.data_ptr(), None) == 0
        torch.cuda.synchronize()
        y0 = y[:check].cpu().numpy()
        work = y.clone()
        assert enc.lib.nhw_stage_prefilter(enc.h, work.data_ptr(), n, q, None) == 0
        torch.cuda.synchronize()
        got = work[:check].cpu().numpy()
        bad = [i for i in range(check) if not np.array_equal(got[i], orc.prefilter(y0[i], q))]
        ts = []
        for _ in range(3):
            work.copy_(y); torch.cuda.synchronize()
            t0 = time.time()
            enc.lib.nhw_stage_prefilter(enc.h, work.data_ptr(), n, q, None)
            torch.cuda.synchronize()
            ts.append((time.time() - t0) * 1e3)
        try:
            import ctypes
            acc = np.zeros(14)
            for i in range(0, n, max(1, n // 64)):
                m = np.zeros(512 * 512, np.uint8)
                enc.lib.nhw_debug_read(enc.h, 17, i, ctypes.c_void_p(m.ctypes.data), ctypes.c_size_t(512 * 512))
                acc += m[511 * 512: 511 * 512 + 112].view(np.int64)
            if acc.sum() > 0:
                names = ["load+sync", "A parallel", "A serial", "copy/codes/hits", "chain", "apply+marker rows"]
                per_row = (n // max(1, n // 64)) * 510
                print("   phase cycles per row (mean over sampled images):", {k: int(v / per_row) for k, v in zip(names, acc)})
                kinds = ["fast single pair", "machine_step + cache", "burst taken", "burst declined"]
                print("   chain steps per row, cycles a step:", {k: (round(acc[7 + 2 * i] / per_row, 1), int(acc[6 + 2 * i] / max(1, acc[7 + 2 * i]))) for i, k in enumerate(kinds)})
        except Exception as ex:
            print("   (no phase clocks:", ex, ")")
        print(f"q{q}: prefilter stage {min(ts):.1f} ms / {n} images (runs {', '.join(f'{t:.1f}' for t in ts)}); oracle check of {check}: {'OK' if not bad else 'MISMATCH ' + str(bad)}", flush=True)

if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [1, 10, 16], n=int(os.environ.get("NHW_N", "4096")))
