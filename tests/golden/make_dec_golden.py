"""Generates tests/golden/dec/: .nhw files written by the REFERENCE encoder (oracle/_ref/libnhwref_enc.so, built from
/root/reference/encoder) at qualities the GPU encoder does not cover yet (1..16) plus a few above, and the SHA-256 of the
BMP the REFERENCE decoder (oracle/_ref/libnhwref_dec.so) writes for each.  Run in the build container only
(needs /root/reference to have been built into oracle/_ref):   python tests/golden/make_dec_golden.py
The files are data (inputs and expected-output digests), not reference source."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.harness import RefEncoder, RefDecoder, class_image
from oracle.oraclepy import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "dec")

def main():
    O, RE, RD = Oracle(), RefEncoder(), RefDecoder()
    man = {}
    cases = [(q, 0) for q in range(1, 24)] + [(1, 1), (10, 1), (6, 2), (14, 3), (16, 4), (20, "blocks"), (10, "gradient"), (3, "blocks"), (10, "tiles"), (20, "tiles"), (23, "tiles")]
    for q, s in cases:
        img = O.synth(s) if isinstance(s, int) else class_image(s, 0)
        nhw = RE.encode(img, q)
        name = f"q{q:02d}_{s}.nhw"
        with open(os.path.join(OUT, name), "wb") as fh:
            fh.write(nhw)
        bmp = RD.bmp(nhw)
        man[name] = {"quality": q, "input": s, "nhw_sha256": hashlib.sha256(nhw).hexdigest(), "bmp_sha256": hashlib.sha256(bmp).hexdigest(), "bytes": len(nhw)}
    with open(os.path.join(OUT, "manifest.json"), "w") as fh:
        json.dump(man, fh, indent=1, sort_keys=True)
    print(len(man), "files", sum(v["bytes"] for v in man.values()), "bytes")

if __name__ == "__main__":
    main()
