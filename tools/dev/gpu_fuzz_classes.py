"""Developer tool (GPU box): images of many classes (flat / gradient / lattice / sinusoid backgrounds with dots, lines, noise, blocks; grey and
colour) through the GPU encoder at every quality against the oracle.  usage: python tools/dev/gpu_fuzz_classes.py [n_images] [first_seed]"""
import hashlib, os, sys
import numpy as np
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def make(seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:512, 0:512]
    kind = rng.integers(0, 4)
    if kind == 0: base = np.full((512, 512), rng.integers(20, 230), float)
    elif kind == 1: base = (xx * rng.uniform(0, 0.4) + yy * rng.uniform(0, 0.4) + rng.integers(0, 50))
    elif kind == 2:
        lat = rng.integers(0, 256, (9, 9)).astype(float)
        gy = yy / 64.0; gx = xx / 64.0; iy = gy.astype(int); ix = gx.astype(int); fy = gy - iy; fx = gx - ix
        base = (lat[iy, ix] * (1 - fx) + lat[iy, ix + 1] * fx) * (1 - fy) + (lat[iy + 1, ix] * (1 - fx) + lat[iy + 1, ix + 1] * fx) * fy
    else: base = 128 + rng.uniform(10, 100) * np.sin(xx / rng.uniform(2, 40)) * np.cos(yy / rng.uniform(2, 40))
    img = base.copy()
    for _ in range(rng.integers(0, 4)):
        f = rng.integers(0, 5)
        amp = rng.integers(5, 200) * (1 if rng.random() < 0.5 else -1)
        if f == 0: img += np.where(rng.random((512, 512)) < 10 ** rng.uniform(-3.5, -1), amp, 0)
        elif f == 1: img += np.where((xx % rng.integers(3, 120)) == 0, amp, 0)
        elif f == 2: img += np.where((yy % rng.integers(3, 120)) == 0, amp, 0)
        elif f == 3: img += rng.integers(-abs(amp) // 4 - 1, abs(amp) // 4 + 2, (512, 512))
        else:
            for _ in range(rng.integers(1, 30)):
                y0, x0 = rng.integers(0, 480, 2); h, w = rng.integers(2, 60, 2); img[y0:y0 + h, x0:x0 + w] += rng.integers(-80, 80)
    img = np.clip(img, 0, 255).astype(np.uint8)
    if rng.random() < 0.5: return np.ascontiguousarray(np.repeat(img[:, :, None], 3, 2))
    c2 = np.clip(img.astype(int) + rng.integers(-30, 30), 0, 255).astype(np.uint8); c3 = np.roll(img, rng.integers(0, 3), 1)
    return np.ascontiguousarray(np.stack([img, c2, c3], 2))


def want_chunk(args):
    from oracle.oraclepy import Oracle
    q, seeds = args
    o = Oracle()
    out = []
    for s in seeds:
        try:
            out.append(hashlib.sha1(o.encode(make(s), q)).hexdigest())
        except RuntimeError as ex:                       # the reference's exit(-1) (code-book overflow): the GPU must report the same status
            out.append(str(ex).split("rc=")[-1])
    return out


def dec_chunk(files):
    from oracle.oraclepy import Oracle
    o = Oracle()
    return [hashlib.sha1(o.decode(f)[0].tobytes()).hexdigest() for f in files]


def encode_with_status(enc, imgs, q):
    """files (b"" where the encoder reports a per-image status) and the statuses"""
    import torch
    o, sizes, status = enc.encode_device(torch.from_numpy(imgs).cuda(), q)
    torch.cuda.synchronize()
    o, sizes, status = o.cpu().numpy().reshape(len(imgs), -1), sizes.cpu().numpy(), status.cpu().numpy()
    return [o[i, :sizes[i]].tobytes() if status[i] == 0 else b"" for i in range(len(imgs))], [int(x) for x in status]


def main(n=192, first=0):
    import nhwcodec_amd as na
    seeds = list(range(first, first + n))
    imgs = np.stack([make(s) for s in seeds])
    enc = na.Encoder(0, n)
    bad_total = 0
    for q in range(1, 24):
        files, status = encode_with_status(enc, imgs, q)
        got = [hashlib.sha1(f).hexdigest() if st == 0 else str(st) for f, st in zip(files, status)]
        chunks = [(q, seeds[i:i + 8]) for i in range(0, n, 8)]
        with ProcessPoolExecutor(max_workers=min(48, os.cpu_count() or 8)) as ex:
            want = [h for part in ex.map(want_chunk, chunks) for h in part]
        bad = [seeds[i] for i in range(n) if got[i] != want[i]]
        bad_total += len(bad)
        # ... and the files back through the GPU decoder against the oracle's decoder
        ok = [i for i in range(n) if status[i] == 0]
        okf = [files[i] for i in ok]
        dec = na.Decoder(0, n)
        px, qs = dec.decode(okf)
        dec.close()
        with ProcessPoolExecutor(max_workers=min(48, os.cpu_count() or 8)) as ex:
            dwant = [h for part in ex.map(dec_chunk, [okf[i:i + 8] for i in range(0, len(okf), 8)]) for h in part]
        dbad = [seeds[ok[i]] for i in range(len(ok)) if hashlib.sha1(px[i].tobytes()).hexdigest() != dwant[i] or qs[i] != q]
        bad_total += len(dbad)
        print(f"q{q}: encode {len(bad)} of {n} images differ {bad[:8]}; decode {len(dbad)} differ {dbad[:8]}", flush=True)
    print("TOTAL differing:", bad_total)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 192, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
