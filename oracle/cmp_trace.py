"""Developer tool: diff the oracle's checkpoint trace against the reference's (oracle/_ref)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle.harness import RefEncoder, synth_image
from oracle.oraclepy import Oracle


def cmp(seed, q, verbose=True, img=None):
    r, o = RefEncoder(), Oracle()
    if img is None:
        img = o.synth(seed)
    d_ref, t_ref = r.encode(img, q, trace=True)
    d_or, t_or = o.encode(img, q, trace=True)
    ok = True
    for k, ((n1, b1), (n2, b2)) in enumerate(zip(t_ref, t_or)):
        if n1 != n2:
            print(f"  [{k}] name mismatch ref={n1} oracle={n2}"); ok = False; break
        for bi, (x, y) in enumerate(zip(b1, b2)):
            if x != y:
                ok = False
                if len(x) == len(y) and len(x) % 2 == 0 and len(x) >= 65536:
                    ax, ay = np.frombuffer(x, np.int16), np.frombuffer(y, np.int16)
                    bad = np.nonzero(ax != ay)[0]
                    st = 512 if len(ax) == 262144 else 256
                    print(f"  [{k}] {n1} blob{bi}: {len(bad)} diffs, first at {bad[0]} (row {bad[0]//st}, col {bad[0]%st}) ref={ax[bad[0]]} or={ay[bad[0]]}")
                else:
                    ax, ay = np.frombuffer(x, np.uint8), np.frombuffer(y, np.uint8)
                    m = min(len(ax), len(ay)); bad = np.nonzero(ax[:m] != ay[:m])[0]
                    print(f"  [{k}] {n1} blob{bi}: len ref={len(x)} or={len(y)} first diff {bad[0] if len(bad) else m}")
        if not ok:
            break
    if len(t_ref) != len(t_or) and ok:
        print("  trace lengths differ", len(t_ref), len(t_or)); ok = False
    same = d_ref == d_or
    if verbose or not (ok and same):
        print(f"seed {seed} q{q}: trace {'OK' if ok else 'MISMATCH'}; bytes ref={len(d_ref)} or={len(d_or)} {'IDENTICAL' if same else 'DIFFER'}")
    return ok and same


if __name__ == "__main__":
    qs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [20]
    seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    bad = 0
    for s in seeds:
        for q in qs:
            bad += not cmp(s, q)
    sys.exit(1 if bad else 0)
