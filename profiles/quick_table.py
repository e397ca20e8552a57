"""kernel table: calls per batch, ms per batch, GB per batch (2 x FETCH_SIZE + WRITE_SIZE, KiB counters), TB/s.
usage: quick_table.py kernel_stats.txt pmc.json <batches in the run>"""
import json, re, sys
stats, pmc, nb = sys.argv[1], json.load(open(sys.argv[2])), float(sys.argv[3])
rows, tot_ms, tot_gb = [], 0.0, 0.0
for ln in open(stats):
    m = re.match(r"(.{78})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", ln)
    if not m or "at::" in ln or "rocclr" in ln or "k_synth" in ln:
        continue
    name = m.group(1).strip().split("(")[0]
    calls, total = int(m.group(2)), float(m.group(3))
    key = max((k for k in pmc if len(k.strip()) >= 5 and name.startswith(k.strip())), key=len, default=None)
    gb = None
    if key and "FETCH_SIZE" in pmc[key] and "WRITE_SIZE" in pmc[key]:
        gb = (2 * pmc[key]["FETCH_SIZE"]["per_launch"] + pmc[key]["WRITE_SIZE"]["per_launch"]) * 1024 / 1e9 * calls / nb
    rows.append((total / nb, name, calls / nb, gb))
    tot_ms += total / nb; tot_gb += gb or 0
for ms, name, calls, gb in sorted(rows, reverse=True):
    print(f"{name[:44]:44s} x{calls:<4.0f} {ms:7.3f} ms {gb if gb is not None else float('nan'):7.2f} GB {((gb or 0) / ms if ms else 0):6.2f} TB/s")
print(f"{'total':44s}       {tot_ms:7.3f} ms {tot_gb:7.2f} GB")
