"""Developer tool (GPU box): run the HIP batch driver stage by stage and compare every live plane with the
oracle's checkpoint trace.  usage: python tests/gpu_stage_debug.py [q ...]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhwcodec_amd  # noqa: E402
from oracle.oraclepy import Oracle  # noqa: E402

B = dict(JPEG=0, PROC=1, PU=2, PV=3, CJPEG=4, CPROC=5, LL1=6, YIN=14, SCAN=17)   # YIN: the fused front's luma input plane (B_KMAP)


def read(enc, buf, img, nbytes):
    out = np.empty(nbytes, np.uint8)
    rc = enc.lib.nhw_debug_read(enc.h, buf, img, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(nbytes))
    assert rc == 0, rc
    return out.tobytes()


def first_diff(a, b, dtype, stride):
    x, y = np.frombuffer(a, dtype), np.frombuffer(b, dtype)
    bad = np.nonzero(x != y)[0]
    return f"{len(bad)} diffs, first at {bad[0]} (row {bad[0] // stride}, col {bad[0] % stride}): hip={x[bad[0]]} oracle={y[bad[0]]}"


def main(qs, seeds=(0, 1)):
    import torch
    orc = Oracle()
    enc = nhwcodec_amd.Encoder(0, max_batch=len(seeds))
    imgs = np.stack([orc.synth(s) for s in seeds])
    d_in = torch.from_numpy(imgs).cuda()
    allok = True
    for q in qs:
        traces = [orc.encode(imgs[i], q, trace=True) for i in range(len(seeds))]
        shift = 0 if q < 22 else -1
        # (stage, trace-record index among same-named records, name, [(buffer, blob index, bytes)])
        plan = [(1, 0, "downsample_YUV420", [("YIN", 0, 8 * 65536), ("PU", 1, 65536), ("PV", 2, 65536)])]
        # (stage 2, the pre-filtered luma plane, never reaches HBM with the fused front kernel)
        L = [(3, 0, "wavelet_analysis_512"), (5, 0, "wavelet_analysis_256"), (6, 0, "offsetY_recons256_p1"), (7, 0, "wavelet_synthesis_256"),
             (9, 1, "wavelet_analysis_256"), (11, 0, "offsetY_recons256_p0"), (12, 1, "wavelet_synthesis_256")]
        for st, k, nm in L:
            plan.append((st + shift, k, nm, [("JPEG", 0, 8 * 65536), ("PROC", 1, 8 * 65536)]))
        if q >= 22:   # below that the quantiser writes the symbol stream directly and leaves the plane alone
            plan.append((13 + shift, 0, "offsetY", [("PROC", 0, 8 * 65536)]))
        for comp in (0, 1):
            base = 13 + shift + 12 * comp
            # the chroma records named wavelet_analysis_256 come after the two luma ones
            C = [(2, 2 + comp, "wavelet_analysis_256"), (4, 2 * comp, "wavelet_analysis_128"), (5, comp, "offsetUV_recons256_c1"),
                 (6, 2 * comp, "wavelet_synthesis_128"), (8, 2 * comp + 1, "wavelet_analysis_128"), (10, comp, "offsetUV_recons256_c0"),
                 (11, 2 * comp + 1, "wavelet_synthesis_128")]
            for st, k, nm in C:
                plan.append((base + st, k, nm, [("CJPEG", 0, 2 * 65536), ("CPROC", 1, 2 * 65536)]))
            plan.append((base + 12, comp, "offsetUV", [("CPROC", 0, 2 * 65536)]))
        ok = True
        for st, k, nm, bufs in plan:
            enc.lib.nhw_debug_stop_after(enc.h, st)
            enc.encode_device(d_in, q)
            torch.cuda.synchronize()
            for i in range(len(seeds)):
                recs = [b for n, b in traces[i][1] if n == nm]
                for bname, bi, nbytes in bufs:
                    got = read(enc, B[bname], i, nbytes)
                    want = recs[k][bi]
                    if bname == "JPEG" and st > 1:      # only the 256x256 corner of the luma jpeg plane is live after level 1
                        got = np.frombuffer(got, np.int16).reshape(512, 512)[:256, :256].tobytes()
                        want = np.frombuffer(want, np.int16).reshape(512, 512)[:256, :256].tobytes()
                    if got != want:
                        ok = False
                        dt = np.uint8 if bname in ("PU", "PV") else np.int16
                        print(f"q{q} stage {st} {nm}#{k} img{i} {bname}: MISMATCH {first_diff(got, want, dt, 256 if bname.startswith(('C', 'PU', 'PV')) else 512)}")
            if not ok:
                break
        enc.lib.nhw_debug_stop_after(enc.h, 0)
        if ok:
            got = enc.encode(imgs, q)
            for i in range(len(seeds)):
                if got[i] != traces[i][0]:
                    ok = False
                    a, b = np.frombuffer(got[i], np.uint8), np.frombuffer(traces[i][0], np.uint8)
                    m = min(len(a), len(b)); bad = np.nonzero(a[:m] != b[:m])[0]
                    print(f"q{q} final img{i}: sizes hip={len(a)} oracle={len(b)} first diff {bad[0] if len(bad) else m}")
        print(f"q{q}: {'ALL STAGES + .nhw IDENTICAL' if ok else 'FAILED'}")
        allok &= ok
    t = enc.timing()
    print("timing ms", {n: round(getattr(t, n), 3) for n, _ in t._fields_})
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main([int(a) for a in sys.argv[1:]] or [20]))
