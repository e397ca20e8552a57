"""Developer tool (GPU box): every file of a full batch against the oracle's decoder (about 20 ms of CPU per file, spread over processes).
usage: python tests/gpu_dec_fullcheck.py [n] [q]"""
import os, sys
import numpy as np, torch
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

def check(args):
    from oracle.oraclepy import Oracle
    import hashlib
    O = Oracle()
    return [hashlib.sha1(O.decode(f)[0].tobytes()).hexdigest() for f in args]

def main():
    import hashlib
    import nhwcodec_amd as na
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    enc = na.Encoder(0, n); img = enc.synth_device(n, 123456); out, sizes, status = enc.encode_device(img, q); torch.cuda.synchronize(); enc.close(); del img
    dec = na.Decoder(0, n)
    offs = torch.arange(n, dtype=torch.int64, device="cuda") * na.OUT_STRIDE
    px, st, _ = dec.decode_device(out, offs, sizes); torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    arena = out.cpu().numpy(); sz = sizes.cpu().numpy(); px = px.cpu().numpy()
    files = [arena[i, : int(sz[i])].tobytes() for i in range(n)]
    chunks = [files[i:i + 64] for i in range(0, n, 64)]
    with ProcessPoolExecutor(max_workers=min(48, os.cpu_count() or 8)) as ex:
        want = [h for part in ex.map(check, chunks) for h in part]
    bad = [i for i in range(n) if hashlib.sha1(px[i].tobytes()).hexdigest() != want[i]]
    print(f"n={n} q={q}: files differing from the oracle: {len(bad)} {bad[:10]}")

if __name__ == "__main__":
    main()
