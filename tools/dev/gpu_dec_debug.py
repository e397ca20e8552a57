"""Developer tool (GPU box): run the HIP decoder stage by stage against the oracle's checkpoints.
usage: python tools/dev/gpu_dec_debug.py [golden|q,seed ...]"""
import ctypes, glob, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nhwcodec_amd as na
from oracle.oraclepy import Oracle

D = dict(META=0, LL=1, PK=2, P1=3, P3=4, P5=5, P6=6, MARKS=7, A=8, B=9, CA=10, CB=11, CU=12)
O = Oracle()

def files(args):
    out = []
    for a in args:
        if a == "golden":
            for p in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "dec", "*.nhw"))):
                out.append((os.path.basename(p), open(p, "rb").read()))
        else:
            q, s = a.split(",")
            out.append((a, O.encode(O.synth(int(s)), int(q))))
    return out

def rd(dec, what, img, nbytes, dtype):
    buf = np.empty(nbytes, np.uint8)
    rc = dec.lib.nhw_dec_debug_read(dec.h, D[what], img, buf.ctypes.data, nbytes)
    assert rc == 0, rc
    return buf.view(dtype)

def plane(dec, what, img, comp=0):
    if what in ("A", "B"):
        return rd(dec, what, img, 8 * 65536 + 8192, np.int16)[2048:2048 + 262144].reshape(512, 512)
    raw = rd(dec, what, img, 2 * (2 * 65536 + 4096), np.int16)
    o = 1024 + comp * (65536 + 2048)
    return raw[o:o + 65536].reshape(256, 256)

def cmp(name, tag, g, o):
    if g.shape != o.shape:
        print(f"  {name} {tag}: SHAPE {g.shape} {o.shape}"); return False
    d = g != o
    if d.any():
        idx = np.argwhere(d)
        print(f"  {name} {tag}: {int(d.sum())} diffs, first {idx[0].tolist()} gpu {g[tuple(idx[0])]} oracle {o[tuple(idx[0])]}, rows {idx[:,0].min()}..{idx[:,0].max()}" + (f" cols {idx[:,1].min()}..{idx[:,1].max()}" if idx.shape[1] > 1 else ""))
        return False
    return True

def main():
    fl = files(sys.argv[1:] or ["20,0"])
    dec = na.Decoder(0, max(len(fl), 1))
    blobs = [f for _, f in fl]
    bad = 0
    def run(stage):
        dec.lib.nhw_dec_debug_stop_after(dec.h, stage)
        try:
            return dec.decode(blobs)
        except na.NhwError as e:
            print("decode error", e); raise
    pr = lambda nhw, i, dt=np.int16: np.frombuffer(O.decode_probe(nhw, i), dt)
    checks = [
        # (the luma plane does not exist before the expansion: the walk leaves a list of values that k_dec_expand turns into rows)
        (3, "A after expand", lambda i, nhw: (plane(dec, "A", i), pr(nhw, 3).reshape(512, 512))),
        (3, "CA0 after expand", lambda i, nhw: (plane(dec, "CA", i, 0), pr(nhw, 30).reshape(256, 256))),
        (3, "CA1 after expand", lambda i, nhw: (plane(dec, "CA", i, 1), pr(nhw, 31).reshape(256, 256))),
        (4, "A after shrink", lambda i, nhw: (plane(dec, "A", i), pr(nhw, 4).reshape(512, 512))),
        (5, "C after L2", lambda i, nhw: (plane(dec, "A", i)[:256, :256], pr(nhw, 5).reshape(512, 512)[:256, :256])),
        (6, "C after residuals", lambda i, nhw: (plane(dec, "A", i)[:256, :256], pr(nhw, 6).reshape(512, 512)[:256, :256])),
        (7, "marks", lambda i, nhw: (rd(dec, "MARKS", i, 2 * 65536, np.uint16)[:len(pr(nhw, 7, np.uint16))], pr(nhw, 7, np.uint16))),
        (8, "Cc0 after L2", lambda i, nhw: (plane(dec, "CA", i, 0)[:128, :128], pr(nhw, 40).reshape(256, 256)[:128, :128])),
        (8, "Cc1 after L2", lambda i, nhw: (plane(dec, "CA", i, 1)[:128, :128], pr(nhw, 41).reshape(256, 256)[:128, :128])),
        (9, "Cc0 after pairs", lambda i, nhw: (plane(dec, "CA", i, 0)[:128, :128], pr(nhw, 42).reshape(256, 256)[:128, :128])),
        (9, "Cc1 after pairs", lambda i, nhw: (plane(dec, "CA", i, 1)[:128, :128], pr(nhw, 43).reshape(256, 256)[:128, :128])),
        (10, "chroma0 before sharpen", lambda i, nhw: (plane(dec, "CA", i, 0), pr(nhw, 44).reshape(256, 256))),
        (10, "chroma1 before sharpen", lambda i, nhw: (plane(dec, "CA", i, 1), pr(nhw, 45).reshape(256, 256))),
        (11, "chroma0 sharpened", lambda i, nhw: (rd(dec, "CU", i, 131072, np.uint8)[:65536].reshape(256, 256), pr(nhw, 46).reshape(256, 256).astype(np.uint8))),
        (11, "chroma1 sharpened", lambda i, nhw: (rd(dec, "CU", i, 131072, np.uint8)[65536:].reshape(256, 256), pr(nhw, 47).reshape(256, 256).astype(np.uint8))),
    ]
    last = None
    for stage, tag, fn in checks:
        if stage != last:
            run(stage); last = stage
        for i, (name, nhw) in enumerate(fl):
            g, o = fn(i, nhw)
            if not cmp(name, tag, g, o): bad += 1
    out, qs = run(0)
    for i, (name, nhw) in enumerate(fl):
        o, q = O.decode(nhw)
        if not cmp(name, "pixels", out[i], o): bad += 1
    print("files", len(fl), "bad checks", bad)

if __name__ == "__main__":
    main()
