"""Developer tool (GPU box): white-noise and hard-edge images (the classes where the quantisers' rare rules fire: values beyond +-127, the
`quant4` pushes, rationed low bits) through the GPU encoder at a range of qualities against the oracle.
usage: python tools/dev/gpu_fuzz_noise.py [first_seed] [n_per_class] [q_first] [q_last]"""
import hashlib, os, sys
import numpy as np
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle.harness import class_image


def want_chunk(args):
    from oracle.oraclepy import Oracle
    q, items = args
    o = Oracle()
    return [hashlib.sha1(o.encode(class_image(k, s), q)).hexdigest() for k, s in items]


def main(first=10, n=16, q0=1, q1=16):
    import nhwcodec_amd as na
    items = [(k, s) for k in ("noise", "blocks") for s in range(first, first + n)]
    imgs = np.stack([class_image(k, s) for k, s in items])
    enc = na.Encoder(0, len(items))
    total = 0
    for q in range(q0, q1 + 1):
        got = [hashlib.sha1(f).hexdigest() for f in enc.encode(imgs, q)]
        with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
            want = [h for part in ex.map(want_chunk, [(q, items[i:i + 2]) for i in range(0, len(items), 2)]) for h in part]
        bad = [items[i] for i in range(len(items)) if got[i] != want[i]]
        total += len(bad)
        print(f"q{q}: {len(bad)} of {len(items)} differ {bad[:6]}", flush=True)
    print("TOTAL differing:", total)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a)
