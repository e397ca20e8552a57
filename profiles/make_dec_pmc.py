"""profiles/dec_pmc.json from a pmc_summarise.py dump of a run with the decode leg: HBM bytes per launch of every decoder kernel
(2 x FETCH_SIZE per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE, both in KiB), tied to nhw_dec.hip by its hash (bench.py refuses a
stale file).  usage: make_dec_pmc.py pmc.json <commit> <quality> <batch>"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import decoder_source_hash
d = json.load(open(sys.argv[1]))
commit, q, batch = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
ks = {}
for k, v in d.items():
    if "k_dec" not in k or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    name = k.replace("void ", "")
    ks[name] = {"fetch_x2_bytes": 2 * v["FETCH_SIZE"]["per_launch"] * 1024, "write_bytes": v["WRITE_SIZE"]["per_launch"] * 1024,
                "launches_per_batch": v["FETCH_SIZE"]["launches"] // max(1, d.get("k_dec_final", v)["FETCH_SIZE"]["launches"])}
fin = ks.get("k_dec_final", {})
print(json.dumps({"source_hash": decoder_source_hash(), "commit": commit, "quality": q, "batch": batch,
                  "final_bytes_per_file": (fin.get("fetch_x2_bytes", 0) + fin.get("write_bytes", 0)) / batch,
                  "decoder_bytes_per_file": sum((v["fetch_x2_bytes"] + v["write_bytes"]) * v["launches_per_batch"] for v in ks.values()) / batch,
                  "kernels": ks}, indent=1))
