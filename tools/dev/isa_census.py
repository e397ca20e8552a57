#!/usr/bin/env python3
"""Developer tool: per-basic-block instruction census of one kernel in a gfx950 .s file (hipcc -S --cuda-device-only).
usage: isa_census.py file.s kernel-substring     -- prints, per block: label, valu / salu / lds / vmem counts, barriers, branch targets"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key.split()[-1]) is False and ":" in l and key in l.split(":")[0])
blocks = []; cur = {"label": "entry", "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "bar": 0, "br": [], "pk": 0}
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False: break
    if s.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur); cur = {"label": m.group(1), "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "bar": 0, "br": [], "pk": 0}; continue
    if not s or s.startswith(";") or s.startswith("."): continue
    op = s.split()[0]
    if op.startswith("v_"):
        cur["valu"] += 1
        if op.startswith("v_pk_"): cur["pk"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["vmem"] += 1
    elif op == "s_barrier": cur["bar"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch")): cur["br"].append(s.split()[-1])
blocks.append(cur)
tot = 0
for b in blocks:
    tot += b["valu"]
    print(f'{b["label"]:12s} valu {b["valu"]:5d} (pk {b["pk"]:3d}) salu {b["salu"]:4d} lds {b["lds"]:4d} vmem {b["vmem"]:3d} {"BARRIER " * b["bar"]}{" ".join(b["br"])}')
print("static valu total", tot)
