"""Developer tool (GPU box): quality 1..16 -- whole encoder against the oracle, and on a mismatch the batch driver stage by stage against
the oracle's checkpoint trace.  usage: python tools/dev/gpu_low_debug.py [q ...]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nhwcodec_amd  # noqa: E402
from oracle.oraclepy import Oracle  # noqa: E402
from oracle.harness import class_image  # noqa: E402

B = dict(JPEG=0, PROC=1, PU=2, PV=3, CJPEG=4, CPROC=5, LL1=6, L2SAVE=7, YIN=14, SCAN=17)
PL, CP = 8 * 65536, 2 * 65536


def read(enc, buf, img, nbytes):
    out = np.empty(nbytes, np.uint8)
    rc = enc.lib.nhw_debug_read(enc.h, buf, img, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(nbytes))
    assert rc == 0, rc
    return out.tobytes()


def first_diff(a, b, dtype, stride):
    x, y = np.frombuffer(a, dtype), np.frombuffer(b, dtype)
    bad = np.nonzero(x != y)[0]
    return f"{len(bad)} diffs, first at {bad[0]} (row {bad[0] // stride}, col {bad[0] % stride}): hip={x[bad[0]]} oracle={y[bad[0]]}"


def plan_for(q):
    """(stage, record name, k-th record of that name, [(buffer, blob, bytes, live corner or None)])"""
    plan = [(1, "downsample_YUV420", 0, [("JPEG", 0, PL, None), ("PU", 1, 65536, None), ("PV", 2, 65536, None)]),
            (2, "pre_processing", 0, [("YIN", 0, PL, None)]),
            (4, "wavelet_analysis_512", 0, [("JPEG", 0, PL, 256), ("PROC", 1, PL, None)]),
            (5, "wavelet_analysis_256", 0, [("JPEG", 0, PL, 256), ("PROC", 1, PL, None)])]
    st = 5
    if q > 6:
        plan += [(6, "offsetY_recons256_p1", 0, [("JPEG", 0, PL, 256), ("PROC", 1, PL, None)]),
                 (7, "wavelet_synthesis_256", 0, [("JPEG", 0, PL, 256), ("PROC", 1, PL, None)]),
                 (9, "wavelet_analysis_256", 1, [("JPEG", 0, PL, 256), ("PROC", 1, PL, None)])]
        st = 9
    st += 1                                   # Y11 / Y12 / Y13
    if q > 12:
        plan += [(st + 1, "offsetY_recons256_p0", 0, [("JPEG", 0, PL, 256), ("PROC", 1, PL, None)]),
                 (st + 2, "wavelet_synthesis_256", 1 if q > 6 else 0, [("JPEG", 0, PL, 256), ("PROC", 1, PL, None)])]
        st += 2
    st += 1                                   # Y19 .. Y31
    plan.append((st, "offsetY", 0, [("PROC", 0, PL, None)]))
    nl = 2 if q > 6 else 1                    # luma records named wavelet_analysis_256
    for comp in (0, 1):
        base = st + 12 * comp
        C = [(3, nl + comp, "wavelet_analysis_256"), (4, 2 * comp, "wavelet_analysis_128"), (5, comp, "offsetUV_recons256_c1"),
             (6, 2 * comp, "wavelet_synthesis_128"), (9, 2 * comp + 1, "wavelet_analysis_128"), (10, comp, "offsetUV_recons256_c0"),
             (11, 2 * comp + 1, "wavelet_synthesis_128")]
        if q <= 14:
            plan.append((base + 1, "pre_processing_UV", comp, [("CJPEG", 0, CP, None)]))
        for s_, k, nm in C:
            # the level-1 thinning of q<=16 sits between the first analysis and the next record, so only the jpeg side of it is compared
            bufs = [("CJPEG", 0, CP, None)] if (nm == "wavelet_analysis_256") else [("CJPEG", 0, CP, None), ("CPROC", 1, CP, None)]
            plan.append((base + s_, nm, k, bufs))
        plan.append((base + 12, "offsetUV", comp, [("CPROC", 0, CP, None)]))
    return plan


def main(qs, seeds=(0, 1), kinds=("synth",)):
    import torch
    orc = Oracle()
    imgs = []
    for kind in kinds:
        for s in seeds:
            imgs.append(orc.synth(s) if kind == "synth" else class_image(kind, s))
    imgs = np.stack(imgs)
    enc = nhwcodec_amd.Encoder(0, max_batch=len(imgs))
    d_in = torch.from_numpy(imgs).cuda()
    allok = True
    for q in qs:
        want = [orc.encode(imgs[i], q) for i in range(len(imgs))]
        enc.lib.nhw_debug_stop_after(enc.h, 0)
        got = enc.encode(imgs, q)
        bad = [i for i in range(len(imgs)) if got[i] != want[i]]
        if not bad:
            print(f"q{q}: {len(imgs)} images .nhw IDENTICAL")
            continue
        allok = False
        print(f"q{q}: images {bad} differ; walking the stages")
        traces = [orc.encode(imgs[i], q, trace=True)[1] for i in range(len(imgs))]
        done = False
        for st, nm, k, bufs in plan_for(q):
            enc.lib.nhw_debug_stop_after(enc.h, st)
            enc.encode_device(d_in, q)
            torch.cuda.synchronize()
            for i in bad[:2]:
                recs = [b for n, b in traces[i] if n == nm]
                for bname, bi, nbytes, corner in bufs:
                    g = read(enc, B[bname], i, nbytes)
                    w = recs[k][bi]
                    if corner:
                        g = np.frombuffer(g, np.int16).reshape(512, 512)[:corner, :corner].tobytes()
                        w = np.frombuffer(w, np.int16).reshape(512, 512)[:corner, :corner].tobytes()
                    if g != w:
                        dt = np.uint8 if bname in ("PU", "PV") else np.int16
                        strd = 256 if (corner or bname.startswith(("C", "PU", "PV"))) else 512
                        print(f"  q{q} stage {st} {nm}#{k} img{i} {bname}: MISMATCH {first_diff(g, w, dt, strd)}")
                        done = True
            if done:
                break
        if not done:
            print(f"  q{q}: every plane checkpoint matches; the difference is behind them (streams / container)")
            for i in bad[:2]:
                a, b = np.frombuffer(got[i], np.uint8), np.frombuffer(want[i], np.uint8)
                m = min(len(a), len(b)); d = np.nonzero(a[:m] != b[:m])[0]
                print(f"  img{i}: sizes hip={len(a)} oracle={len(b)} first diff {d[0] if len(d) else m} header hip={a[:40].tolist()} oracle={b[:40].tolist()}")
        enc.lib.nhw_debug_stop_after(enc.h, 0)
    return 0 if allok else 1


if __name__ == "__main__":
    args = sys.argv[1:]
    kinds = ("synth",)
    if args and args[0] == "--classes":
        kinds = ("synth", "noise", "blocks", "tiles"); args = args[1:]
    sys.exit(main([int(a) for a in args] or [10], kinds=kinds))
