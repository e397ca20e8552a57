"""Regenerates the kernel table of DESIGN.md (between the KERNEL_TABLE markers) from the committed profile summaries, so that the numbers
cannot drift from the files they cite.   usage: python profiles/make_design_table.py [--check]
inputs: profiles/round6_kernel_stats_1stream.txt (rocprofv3 --kernel-trace --stats, one stream), profiles/round6_pmc.json (FETCH_SIZE and
WRITE_SIZE per kernel from separate --pmc passes; KiB; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATS = os.path.join(ROOT, "profiles", "round6_kernel_stats_1stream.txt")
PMC = os.path.join(ROOT, "profiles", "round6_pmc.json")
BATCHES = 7.0            # bench.py --steps 5 --warmup 2 in profiles/collect.sh
PH = ["L1", "L2", "L3", "L4A", "C0", "C2", "C3", "C4", "C5", "FINAL", "L4B", "L4C", "L4D", "LLC", "L4C2"]
WV = ["DQ1", "DQ0", "EMIT", "QUANT"]
WHAT = {
    "k_front_image": "a1 + a2 + Y2 + Y3 fused: BGR24 -> Y, 4:2:0 chroma planes out, pre-filter (carry chained in the kernel), both directions of the level-1 analysis, LL copy; a workgroup walks an image (32 rows a band, 512 threads, 78 KB LDS)",
    "k_front_plain": "the same without the pre-filter (q >= 22; q <= 16 and the analysis stage from a luma plane): horizontal pass from registers, 64 rows a band",
    "k_dwt_ana<256>": "level-2 luma analysis of both closed loops (whole block in LDS, persistent workgroups); level-1 chroma analysis only for q <= 14 and the stage checks",
    "k_chroma_l1q": "level-1 chroma analysis from the 4:2:0 byte plane, a quarter of the 256 x 256 block to a workgroup (four a CU); its coefficients are part of SURVEY's 6 B/pixel",
    "k_dwt_ana<128>": "level-2 chroma analysis",
    "k_dwt_syn<256>": "level-2 luma synthesis of the second closed loop",
    "k_dwt_syn<128>": "level-2 chroma synthesis",
    "k_l2_recon": "first closed loop: level-2 synthesis + Y8 (tags -> reconstruction) + Y9 (LL1 pre-compensation) on one LDS residency of the block",
    "L1": "Y5 tag level-2 details", "L2": "Y8, Y9 as a kernel of their own (only the tests' stage checks run it)", "L3": "Y16 LL2 coder (parse), on a stream of its own beside the second dequantiser simulation",
    "L4A": "Y19-Y23: small runs (wavefront per row), residual classification (one table-driven step for every kind) and coding (column walks on LDS tiles)",
    "L4B": "Y24, Y25 position lists + list packing", "L4C": "Y27 detail clean-up (a wavefront takes 64 consecutive rows, neighbours in registers; Y26 only for q >= 22)", "L4D": "Y31 rewrites, on the symbol LIST (non-zero map + values); leaves map and offsets in stream order",
    "L4C2": "Y29 (q >= 22): band reconstruction, half synthesis, res6 / char_res1 / qsetting3 lists",
    "C0": "chroma: copy", "C2": "chroma: dequantiser simulation 1", "C3": "chroma: tags", "C4": "chroma: dequantiser simulation 2",
    "C5": "chroma: marks (running-index fixed point), LL2 emission, quantiser (wavefront per row); V leaves the merged chroma stream as a list", "LLC": "Z1 chroma LL2 coder", "FINAL": "Z2 packetiser + container",
    "k_l4a": "Y19-Y23 of q >= 17: Y21 small runs (wavefront per row, rows in which nothing fires skipped), Y22 + Y23 as ONE column sweep on registers and class tables (eight workgroups a CU)",
    "k_y31": "Y31 rewrites on the symbol LIST (non-zero map + values), 512 threads an image; leaves map and offsets in stream order",
    "k_final": "Z2 packetiser (both parts from the symbol lists) + container; a kernel of its own at four wavefronts a SIMD",
    "DQ1": "a8 dequantiser simulation, first closed loop (wavefront per image)", "DQ0": "a8 dequantiser simulation, second closed loop",
    "EMIT": "Y14/Y15 LL2 emission", "QUANT": "Y28 luma quantiser + Y30: the symbols leave as a list in stream order (wavefront per image); reads the level-2 block from l2save (Y26)",
}


def label(name):
    m = re.match(r"void k_phase<(\d+)>", name)
    if m:
        return f"`k_phase<{PH[int(m.group(1))]}>`", WHAT.get(PH[int(m.group(1))], "")
    m = re.match(r"void k_wave<(\d+)>", name)
    if m:
        return f"`k_wave<{WV[int(m.group(1))]}>`", WHAT.get(WV[int(m.group(1))], "")
    short = name.replace("void ", "").replace("nhw::", "").split("(")[0]
    base = re.sub(r"<.*", "", short)
    return f"`{short}`", WHAT.get(short, WHAT.get(base, ""))


def table():
    pmc = json.load(open(PMC))
    rows, tot_ms, tot_gb = [], 0.0, 0.0
    for ln in open(STATS):
        m = re.match(r"(.{78})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", ln)
        if not m or "at::" in ln or "rocclr" in ln or "k_synth" in ln:
            continue
        name = m.group(1).strip().split("(")[0]
        calls, total = int(m.group(2)), float(m.group(3))
        key = max((k for k in pmc if len(k.strip()) >= 5 and name.startswith(k.strip())), key=len, default=None)
        rd = wr = None
        if key and "FETCH_SIZE" in pmc[key] and "WRITE_SIZE" in pmc[key]:
            rd = 2 * pmc[key]["FETCH_SIZE"]["per_launch"] * 1024 / 1e9 * calls / BATCHES
            wr = pmc[key]["WRITE_SIZE"]["per_launch"] * 1024 / 1e9 * calls / BATCHES
        rows.append((total / BATCHES, name, calls / BATCHES, rd, wr))
        tot_ms += total / BATCHES; tot_gb += (rd or 0) + (wr or 0)
    out = ["| kernel | passes | launches | ms | HBM read + written, GB | TB/s |", "|---|---|---|---|---|---|"]
    for ms, name, calls, rd, wr in sorted(rows, reverse=True):
        lab, what = label(name)
        gb = f"{rd:.2f} + {wr:.2f}" if rd is not None else "n/a"
        tbs = f"{(rd + wr) / ms:.2f}" if rd is not None and ms else ""
        out.append(f"| {lab} | {what} | {calls:.0f} | {ms:.2f} | {gb} | {tbs} |")
    out.append(f"| **sum (one stream, nothing overlaps)** | | | **{tot_ms:.1f}** | **{tot_gb:.1f}** | |")
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a, b = s.index("<!-- KERNEL_TABLE_BEGIN -->"), s.index("<!-- KERNEL_TABLE_END -->")
    new = s[:a] + "<!-- KERNEL_TABLE_BEGIN -->\n" + table() + "\n" + s[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == s else 1)
    open(path, "w").write(new)


if __name__ == "__main__":
    main()
