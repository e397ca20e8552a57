"""Generates the committed golden vectors from the REAL reference (oracle/_ref, canonical build).

Run in the build container only (needs /root/reference to have produced oracle/_ref):
    python tests/golden/make_golden.py
Outputs (data only -- inputs are regenerated from integer seeds, outputs are the reference's bytes):
    tests/golden/nhw/<class>_s<seed>_q<q>.nhw       complete reference .nhw files
    tests/golden/manifest.json                      sha256 of every .nhw + FNV-1a64 of every checkpoint blob
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.harness import RefEncoder, class_image, synth_image  # noqa: E402


def fnv64(b: bytes) -> str:
    import numpy as np
    # FNV-1a over 64-bit little-endian words would need padding; plain sha1 is enough and portable
    return hashlib.sha1(b).hexdigest()[:16]


def main():
    ref = RefEncoder()
    os.makedirs(os.path.join(HERE, "nhw"), exist_ok=True)
    man = {"files": {}, "hashes": {}, "checkpoints": {}}
    full = [("synth", s, q) for s in (0, 1) for q in (17, 18, 19, 20, 21, 22, 23)]
    full += [("noise", 0, 20), ("blocks", 0, 20), ("flat", 0, 20), ("gradient", 0, 23)]
    traced = list(full[:14])
    # quality 1..16 (integer colour, the rationed pre-filter, LL2 smoothing, no closed loops at the bottom): complete files for
    # seed 0 at every setting and for BASELINE config 3's q1 / q10 on a second seed and on the hard classes; checkpoints for the
    # settings where the gate structure changes
    low_full = [("synth", 0, q) for q in range(1, 17)] + [("synth", 1, 1), ("synth", 1, 10), ("noise", 0, 1), ("noise", 0, 10),
                                                          ("blocks", 0, 10), ("tiles", 0, 1)]
    full += low_full
    traced += [("synth", 0, q) for q in (1, 6, 7, 10, 11, 12, 13, 14, 15, 16)]
    hashed = [("synth", s, q) for s in range(2, 8) for q in (17, 20, 23)]
    hashed += [(k, 0, q) for k in ("noise", "blocks", "flat", "black", "white", "gradient") for q in (17, 19, 21, 22, 23)]
    hashed += [("synth", s, q) for s in range(2, 6) for q in range(1, 17)]
    hashed += [(k, 0, q) for k in ("noise", "blocks", "flat", "black", "white", "gradient", "tiles") for q in (1, 3, 5, 7, 8, 9, 10, 12, 13, 15, 16)]
    for kind, seed, q in full + hashed:
        img = synth_image(seed) if kind == "synth" else class_image(kind, seed)
        key = f"{kind}_s{seed}_q{q}"
        want_trace = (kind, seed, q) in traced
        if want_trace:
            data, tr = ref.encode(img, q, trace=True)
            man["checkpoints"][key] = [[n, [fnv64(b) for b in blobs]] for n, blobs in tr]
        else:
            data = ref.encode(img, q)
        man["hashes"][key] = hashlib.sha256(data).hexdigest()
        if (kind, seed, q) in full:
            with open(os.path.join(HERE, "nhw", key + ".nhw"), "wb") as f:
                f.write(data)
            man["files"][key] = len(data)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("files", len(man["files"]), "hashes", len(man["hashes"]))


if __name__ == "__main__":
    main()
