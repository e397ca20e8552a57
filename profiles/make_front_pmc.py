"""profiles/front_pmc.json from a pmc_summarise.py dump: HBM bytes per image of the front launch group (2 x FETCH_SIZE per the gfx950
note of MI355X_MICROARCH.md + WRITE_SIZE, both in KiB), tied to the kernel sources by their hash (bench.py refuses a stale file).
usage: make_front_pmc.py pmc.json <commit> <quality> <batch>"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_hash
d = json.load(open(sys.argv[1]))
commit, q, batch = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
front, total = {}, 0.0
for k, v in d.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    b = (2 * v["FETCH_SIZE"]["per_launch"] + v["WRITE_SIZE"]["per_launch"]) * 1024
    if any(n in k for n in ("k_front_image", "k_front_plain", "k_front_stale")) or k.endswith("k_color"):
        front[k] = {"fetch_x2_bytes": 2 * v["FETCH_SIZE"]["per_launch"] * 1024, "write_bytes": v["WRITE_SIZE"]["per_launch"] * 1024}
        total += b
print(json.dumps({"source_hash": kernel_source_hash(), "commit": commit, "quality": q, "batch": batch, "file": "profiles/round6_pmc.json",
                  "front_bytes_per_image": total / batch, "kernels": front}, indent=1))
