#!/usr/bin/env python3
"""bench.py -- encode Mpixels/s of the MI355X NHW encoder on batches of synthetic 512x512 RGB images.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run, one rank per GPU.  A "step" = one pass of the whole encode hot path (BGR24 in HBM ->
.nhw bytes in HBM) over one batch of `--batch` synthetic images per GPU (BASELINE.json configs[1]: 4096 images,
-q20).  Images are independent, so ranks shard the work with no data-path collective (weak scaling); the only
collectives are the 32-byte work descriptor broadcast and the all-gather of per-rank {bytes, checksum}.

Rank 0 prints ONE JSON line.  `roofline` is measured with HIP events on the launch stream inside the timed
region; `cpu_baseline` times the reference encoder (oracle/_ref, kind "reference") or, if that build is
absent, the plain-C port (oracle/, kind "port") on the host cores over a bounded sample.
"""
import argparse
import math
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MPIX_PER_IMAGE = 0.262144
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
FRONT_BYTES_PER_IMAGE = 786432 + 786432   # 3 B/px read + 3 B/px written (Y coeffs + 2 x chroma)
# HBM bytes of the front launch group per image from the PMC counters (profiles/round1_final_pmc.json, batch 4096, -q20: FETCH_SIZE x 2 per
# the gfx950 note + WRITE_SIZE, separate rocprofv3 --pmc passes): k_color 3.62+2.68 GB, k_front_rowtail 0.28+0.04, k_front_chain 0.04, k_front_band 2.60+3.22
FRONT_PMC_BYTES_PER_IMAGE = (3.618e9 + 2.684e9 + 0.276e9 + 0.040e9 + 0.040e9 + 2.600e9 + 3.224e9) / 4096


def _cpu_worker(args):
    kind, seeds, q = args
    import numpy as np  # noqa: F401
    from oracle.oraclepy import Oracle
    orc = Oracle()
    imgs = [orc.synth(s) for s in seeds]
    if kind == "reference":
        from oracle.harness import RefEncoder
        enc = RefEncoder()
        f = lambda im: enc.encode(im, q)
    else:
        f = lambda im: orc.encode(im, q)
    f(imgs[0])
    t0 = time.perf_counter()
    for im in imgs:
        f(im)
    return len(imgs), time.perf_counter() - t0


def _cpu_dec_worker(args):
    kind, files, reps = args
    from oracle.oraclepy import Oracle
    if kind == "reference":
        from oracle.harness import RefDecoder
        rd = RefDecoder()
        f = rd.bmp                      # the unmodified reference decoder incl. its BMP writer (a tmpfs file per call)
    else:
        orc = Oracle()
        f = lambda b: orc.decode(b)
    f(files[0])
    t0 = time.perf_counter()
    for _ in range(reps):
        for b in files:
            f(b)
    return reps * len(files), time.perf_counter() - t0


def cpu_decode_baseline(files, budget_s=6.0):
    """Reference decoder on the host cores over a sample of the files the GPU just decoded."""
    import multiprocessing as mp
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libnhwref_dec.so")) else "port"
    cores = max(1, min(os.cpu_count() or 1, 64))
    reps = max(1, int(budget_s / (0.012 * 4)))
    jobs = [(kind, files[(4 * w) % len(files):(4 * w) % len(files) + 4] or files[:4], reps) for w in range(cores)]
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_dec_worker, jobs)
    n = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": round(n * MPIX_PER_IMAGE / busy, 2), "unit": "Mpixels/s", "cores": cores, "kind": kind,
            "sample": f"{n} decodes of {min(len(files), 4 * cores)} of this run's .nhw files, one in-process decoder per core, {busy:.1f} s; "
                      f"{'unmodified reference sources + zero-guard allocator, BMP written to a temporary file' if kind == 'reference' else 'plain-C restatement'}"}


def valu_evidence():
    """VALU issue statistics of the front kernels from the committed PMC pass (profiles/round1_pmc_valu.json, batch 4096, -q20):
    wave-instructions issued / (CUs x kernel cycles) -- why these kernels sit where they do against the HBM roofline."""
    try:
        with open(os.path.join(ROOT, "profiles", "round1_pmc_valu.json")) as fh:
            d = json.load(fh)
    except OSError:
        return None
    out = {}
    for k, v in d.items():
        if "SQ_INSTS_VALU" not in v or "GRBM_GUI_ACTIVE" not in v:
            continue
        name = k.split("::")[-1].split("<")[0].replace("void ", "")
        if name not in ("k_color", "k_front_rowtail", "k_front_band"):
            continue
        cyc = v["GRBM_GUI_ACTIVE"]["per_launch"] / 8.0          # summed over the 8 XCDs
        valu = v["SQ_INSTS_VALU"]["per_launch"]
        out[name] = {"valu_wave_instructions": int(valu), "lane_ops_per_pixel": round(valu * 64 / (4096 * 262144), 1),
                     "issue_frac_of_1_per_cu_cycle": round(valu / (256 * cyc), 3)}
    return out or None


def cpu_baseline(q, budget_s=12.0):
    """Reference encoder on the host cores, bounded sample (about `budget_s` seconds of wall time)."""
    import multiprocessing as mp
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libnhwref_enc.so")) else "port"
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    cores = max(1, min(os.cpu_count() or 1, 64))
    per = max(4, int(budget_s / 0.03))          # ~30 ms per image per core
    per = min(per, 400)
    jobs = [(kind, list(range(w * per, (w + 1) * per)), q) for w in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": round(n * MPIX_PER_IMAGE / busy, 2), "unit": "Mpixels/s", "cores": cores, "kind": kind,
            "sample": f"{n} synthetic 512x512 images (SURVEY 8d generator, seeds 0..{n - 1}), -q{q}, one in-process encoder per core, "
                      f"{busy:.1f} s encode time ({wall:.1f} s incl. process start); {'unmodified reference sources + zero-guard allocator' if kind == 'reference' else 'plain-C restatement'}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="images per GPU per step")
    ap.add_argument("--quality", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode leg (BASELINE config 5) reported next to the encode metric")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import nhwcodec_amd
    from nhwcodec_amd.dist import broadcast_descriptor, gather_summaries, max_over_ranks
    # work descriptor {base, count, quality, seed}: rank 0 decides, everyone receives
    _, batch, q, seed = broadcast_descriptor(dist, dev, 0, args.batch, args.quality, 1234)

    enc = nhwcodec_amd.Encoder(local_rank, max_batch=batch)
    bgr = enc.synth_device(batch, seed_base=seed + rank * batch)     # inputs resident in HBM before timing
    out = enc.alloc_out(batch)
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        enc.encode_device(bgr, q, out)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    front_ms = 0.0
    color_ms = 0.0
    tim = None
    for _ in range(args.steps):
        enc.encode_device(bgr, q, out)
        tim = enc.timing()          # hipEvents recorded on the launch stream; waits for this step's last event
        front_ms += tim.front_ms
        color_ms += tim.color_dwt_ms
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dist, dev, dt)

    _, sizes, status = out
    ok = int((status == 0).sum().item())
    nbytes = int(sizes.to(torch.int64).sum().item())
    chk = int((sizes.to(torch.int64) * torch.arange(1, batch + 1, device=dev)).sum().item() % (1 << 61))
    gathered = gather_summaries(dist, dev, nbytes, chk, ok)

    # BASELINE config 5 beside the headline metric: the batch just encoded goes back through the decoder, HBM to HBM
    # (the encoder's output arena is the decoder's input arena).  Timed after, and apart from, the encode region.
    dec_line = None
    if not args.no_decode:
        dec = nhwcodec_amd.Decoder(local_rank, max_batch=batch)
        offs = torch.arange(batch, dtype=torch.int64, device=dev) * nhwcodec_amd.OUT_STRIDE
        pix = torch.empty((batch, 512, 512, 3), dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            for _ in range(max(1, args.warmup)):
                dec.decode_device(out[0], offs, sizes, pix)
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            s1 = s2 = sc = 0.0
            for _ in range(args.steps):
                _, dst, dq = dec.decode_device(out[0], offs, sizes, pix)
                dtm = dec.timing()
                s1 += dtm.synth1_ms; s2 += dtm.synth2_ms; sc += dtm.color_ms
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            ddt = max_over_ranks(dist, dev, time.perf_counter() - t1)
        dec_ok = int((dst == 0).sum().item())
        err = (pix[:64].float() - bgr[:64].float()).pow(2).mean().item()
        dec_line = {"metric": "decode Mpixels/s (512x512 .nhw batch -> BGR24, BASELINE config 5)", "value": round(batch * world * args.steps * MPIX_PER_IMAGE / ddt, 2),
                    "unit": "Mpixels/s", "ms_per_step": round(ddt / args.steps * 1e3, 3), "files_ok_rank0": dec_ok,
                    "psnr_db_first_64": round(10 * math.log10(255.0 ** 2 / max(err, 1e-9)), 2),
                    "workload": f"the {batch} .nhw files per GPU this run just encoded (-q{q}), decoder arena = encoder arena in HBM",
                    "stage_ms": {"entropy": round(dtm.entropy_ms, 3), "total": round(dtm.total_ms, 3)},
                    # SURVEY 8(d) for config 5: the final reconstruction (level-1 synthesis, both directions, + colour) has 3 B/px of coefficients in and 3 B/px of
                    # BGR out as its algorithmic traffic; here it is three kernels, each listed with its own bytes (hipEvents on the launch stream)
                    "roofline": {"bound": "hbm", "kernel": "k_dec_synth (level 1, first direction) + k_dec_synth (level 1, second direction, -> bytes) + k_dec_color",
                                 "achieved": round(batch * 1572864 / ((s1 + s2 + sc) / 1e3 / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(batch * 1572864 / ((s1 + s2 + sc) / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                 "algorithmic_bytes_per_image": 1572864,
                                 "kernels": [{"kernel": "level-1 synthesis, rows (int16 in, int16 out)", "ms": round(s1 / args.steps, 3), "algorithmic_bytes": batch * 1048576,
                                              "achieved": round(batch * 1048576 / (s1 / 1e3 / args.steps) / 1e9, 1)},
                                             {"kernel": "level-1 synthesis, columns (int16 in, clipped bytes out)", "ms": round(s2 / args.steps, 3), "algorithmic_bytes": batch * 786432,
                                              "achieved": round(batch * 786432 / (s2 / 1e3 / args.steps) / 1e9, 1)},
                                             {"kernel": "x2 chroma + colour matrix (Y + 4:2:0 U,V bytes in, BGR24 out)", "ms": round(sc / args.steps, 3), "algorithmic_bytes": batch * (262144 + 131072 + 786432),
                                              "achieved": round(batch * (262144 + 131072 + 786432) / (sc / 1e3 / args.steps) / 1e9, 1)}]}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            szh = sizes[:256].cpu().numpy(); ar = out[0][:256].cpu().numpy()
            dec_line["cpu_baseline"] = cpu_decode_baseline([ar[i, : int(szh[i])].tobytes() for i in range(min(256, batch))])
        dec.close()

    if rank == 0:
        total_images = batch * world * args.steps
        value = total_images * MPIX_PER_IMAGE / dt
        front_s = front_ms / 1e3 / args.steps
        front_images = tim.front_images or batch      # the batch runs as `parts` sub-batches on their own streams; the events bracket the first one's launch group
        achieved = front_images * FRONT_BYTES_PER_IMAGE / front_s / 1e9
        line = {
            "metric": "encode Mpixels/s (512x512 RGB batch)", "value": round(value, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": f"batch of {batch} synthetic 512x512 BGR24 images per GPU, -q{q}, whole encoder (BGR in HBM -> .nhw bytes in HBM)",
                       "images_per_gpu": batch, "quality": q, "parallelism": f"dp{world} (independent images, no data-path collective)"},
            "roofline": {"bound": "hbm", "kernel": "front = k_color + k_front_rowtail + k_front_chain + k_front_band (colour, pre-filter and level-1 analysis; the band kernel fuses pre-filter + both filter directions)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": int(front_images * FRONT_PMC_BYTES_PER_IMAGE) if q == 20 else None, "traffic_unit": "bytes per launch group (PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/round1_final_pmc.json, scaled from batch 4096)",
                         "kernels": [   # the members of the group, each with its own algorithmic bytes and live hipEvent time
                             {"kernel": "k_color (BGR24 -> Y int16 + 4:2:0 U,V)", "ms": round(color_ms / args.steps, 3), "algorithmic_bytes": front_images * (786432 + 524288 + 131072),
                              "achieved": round(front_images * (786432 + 524288 + 131072) / (color_ms / 1e3 / args.steps) / 1e9, 1), "frac": round(front_images * (786432 + 524288 + 131072) / (color_ms / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
                             {"kernel": "k_front_rowtail + k_front_chain + k_front_band (pre-filter + level-1 analysis)", "ms": round((front_ms - color_ms) / args.steps, 3), "algorithmic_bytes": front_images * (524288 + 786432),
                              "achieved": round(front_images * (524288 + 786432) / ((front_ms - color_ms) / 1e3 / args.steps) / 1e9, 1), "frac": round(front_images * (524288 + 786432) / ((front_ms - color_ms) / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}],
                         "algorithmic_bytes_per_launch": front_images * FRONT_BYTES_PER_IMAGE, "images_per_launch": front_images, "sub_batches": tim.parts, "algorithmic_bytes_per_image": FRONT_BYTES_PER_IMAGE, "ms_per_launch_group": round(front_s * 1e3, 3)},
            "stage_ms": {"front": round(tim.front_ms, 3), "luma_tail (chroma sequence alongside, on its own stream)": round(tim.luma_ms, 3), "chroma left over": round(tim.chroma_ms, 3),
                         "entropy+container": round(tim.entropy_ms, 3), "total": round(tim.total_ms, 3)},
            "images_ok": [int(g[2]) for g in gathered], "bytes_out": [int(g[0]) for g in gathered],
        }
        ev = valu_evidence()
        if ev:
            line["roofline"]["valu_pmc"] = ev
        if dec_line:
            line["decode"] = dec_line
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(q)
        print(json.dumps(line), flush=True)
    enc.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
