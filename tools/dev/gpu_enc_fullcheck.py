"""Developer tool (GPU box): every image of a full batch against the oracle's encoder, spread over processes.
usage: python tools/dev/gpu_enc_fullcheck.py [n] [q] [seed_base]"""
import hashlib, os, sys
import numpy as np, torch
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

def enc_chunk(args):
    from oracle.oraclepy import Oracle
    q, seeds = args
    O = Oracle()
    return [hashlib.sha1(O.encode(O.synth(s), q)).hexdigest() for s in seeds]

def full_encode_check(n, q, seed_base, workers=None):
    import nhwcodec_amd as na
    enc = na.Encoder(0, n); img = enc.synth_device(n, seed_base); out, sizes, status = enc.encode_device(img, q); torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0
    arena = out.cpu().numpy(); sz = sizes.cpu().numpy(); enc.close()
    got = [hashlib.sha1(arena[i, : int(sz[i])].tobytes()).hexdigest() for i in range(n)]
    seeds = list(range(seed_base, seed_base + n))
    chunks = [(q, seeds[i:i + 32]) for i in range(0, n, 32)]
    with ProcessPoolExecutor(max_workers=workers or min(56, os.cpu_count() or 8)) as ex:
        want = [h for part in ex.map(enc_chunk, chunks) for h in part]
    return [i for i in range(n) if got[i] != want[i]]

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sb = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
    bad = full_encode_check(n, q, sb)
    print(f"n={n} q={q} seeds {sb}..: images differing from the oracle: {len(bad)} {bad[:10]}")
