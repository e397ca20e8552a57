#!/usr/bin/env python3
"""bench.py -- encode Mpixels/s of the MI355X NHW encoder on batches of synthetic 512x512 RGB images.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N>1 the driver launches it under
torch.distributed.run, one rank per GPU; started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) it
re-executes itself under torch.distributed.run with N ranks, so the documented command works either way and the line always
says n_gpus = N.  A "step" = one pass of the whole encode hot path (BGR24 in HBM -> .nhw bytes in HBM) over one batch of
`--batch` synthetic images per GPU (BASELINE.json configs[1]: 4096 images, -q20).  Images are independent, so ranks shard the
work with no data-path collective; the only collectives are the 32-byte work descriptor broadcast and the all-gather of
per-rank {bytes, checksum, images ok}.  Default: weak scaling (`--batch` images per GPU).  `--total-images T` (BASELINE
configs[3]: 65536 over 8 GPUs) cuts ONE job of T images into contiguous per-rank ranges instead (strong scaling).
`--dry` runs the multi-process plumbing alone on CPU over gloo (tests).

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
FRONT_KERNEL_NAME = ("THREE launches priced together: k_front_image (one fused kernel = the whole front launch group: BGR24 -> Y + 4:2:0 chroma planes, luma pre-filter with its carry "
                     "chained inside the kernel, both directions of the luma level-1 analysis; a workgroup walks an image top to bottom with a rolling window of rows in LDS, the luma "
                     "plane never travels) + the two k_chroma_l1q launches of the chroma level-1 analysis (U, V; a quarter of a 256 x 256 block to a workgroup), whose coefficients are part of the 6 B/pixel; "
                     "frac_front_kernel_alone prices the same bytes over k_front_image only")
VALU_PMC_FILE = os.path.join(ROOT, "profiles", "round6_pmc_valu.json")   # profiles/collect_valu.sh: SQ_INSTS_VALU, GRBM_GUI_ACTIVE ... of the same bench command
PMC_FILE = os.path.join(ROOT, "profiles", "front_pmc.json")     # written by profiles/pmc_summarise.py from separate rocprofv3 --pmc passes


def kernel_source_hash():
    """sha256 over the sources of the front kernels (nhw_front.hip and the workspace header it includes): ties the committed PMC profile of
    the front launch group to the code it was taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nhwcodec_amd", "csrc")
    for f in ("nhw_front.hip", "nhw_front_image.h", "nhw_ws.h"):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


DEC_PMC_FILE = os.path.join(ROOT, "profiles", "dec_pmc.json")       # profiles/collect_dec.sh


def decoder_source_hash():
    """sha256 over nhw_dec.hip: ties the committed PMC profile of the decoder kernels to the code it was taken from"""
    import hashlib
    h = hashlib.sha256()
    h.update(open(os.path.join(ROOT, "nhwcodec_amd", "csrc", "nhw_dec.hip"), "rb").read())
    return h.hexdigest()[:16]


def final_traffic(q, files):
    """HBM bytes of one k_dec_final launch from the committed PMC file, scaled to this batch -- or None (stale file, other quality, absent)"""
    try:
        with open(DEC_PMC_FILE) as fh:
            d = json.load(fh)
    except OSError:
        return None, "no committed PMC file"
    if d.get("source_hash") != decoder_source_hash():
        return None, f"profiles/dec_pmc.json was taken from other kernel sources (commit {d.get('commit')}); not reused"
    if d.get("quality") != q:
        return None, "PMC file is for another quality"
    return int(d["final_bytes_per_file"] * files), (f"PMC: 2 x FETCH_SIZE + WRITE_SIZE of k_dec_final, profiles/dec_pmc.json, commit {d.get('commit')}, batch {d.get('batch')}; "
                                                    f"the whole decoder moves {d.get('decoder_bytes_per_file', 0) / 1e6:.2f} MB per file")


def front_traffic(q, images):
    """HBM bytes of the front launch group from the committed PMC file (2 x FETCH_SIZE per the gfx950 note + WRITE_SIZE, per image, batch
    4096), scaled to this launch -- or None when the file was taken from other kernel sources, another quality, or is absent."""
    try:
        with open(PMC_FILE) as fh:
            d = json.load(fh)
    except OSError:
        return None, "no committed PMC file"
    if d.get("source_hash") != kernel_source_hash():
        return None, f"profiles/front_pmc.json was taken from other kernel sources ({d.get('source_hash')}, commit {d.get('commit')}); not reused"
    if d.get("quality") != q:
        return None, "PMC file is for another quality"
    return int(d["front_bytes_per_image"] * images), f"PMC: 2 x FETCH_SIZE + WRITE_SIZE over the group's kernels, {d.get('file')}, commit {d.get('commit')}, batch {d.get('batch')}"


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
    """VALU issue statistics of the front kernel from the committed PMC pass (profiles/round6_pmc_valu.json, batch 4096, -q20):
    wave-instructions issued / (CUs x kernel cycles) -- the kernel's limiter is vector-instruction issue, not HBM; this is its fraction of
    that roofline (one wave64 instruction per CU and cycle: four 16-lane SIMDs)."""
    try:
        with open(VALU_PMC_FILE) as fh:
            d = json.load(fh)
    except OSError:
        return None
    out = {}
    for k, v in d.items():
        if "SQ_INSTS_VALU" not in v or "GRBM_GUI_ACTIVE" not in v:
            continue
        name = k.split("::")[-1].split("<")[0].replace("void ", "")
        if name not in ("k_front_image", "k_front_plain"):
            continue
        cyc = v["GRBM_GUI_ACTIVE"]["per_launch"] / 8.0          # summed over the 8 XCDs
        valu = v["SQ_INSTS_VALU"]["per_launch"]
        out[name] = {"valu_wave_instructions": int(valu), "lane_ops_per_pixel": round(valu * 64 / (4096 * 262144), 1),
                     "issue_frac_of_1_per_cu_cycle": round(valu / (256 * cyc), 3)}
    return out or None


def physical_cores():
    """physical cores of the box (distinct (package, core id) pairs of /proc/cpuinfo), or None"""
    try:
        seen, pkg, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if core is not None:
                    seen.add((pkg, core))
                pkg = core = None
        return len(seen) or None
    except OSError:
        return None


def vanilla_xargs_baseline(q, cores, budget_s=8.0):
    """the reference exactly as its README builds it (`gcc *.c -O3`, oracle/_ref/nhw-enc, no shim), one process per
    image under `xargs -P cores`, BMPs on tmpfs."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "nhw-enc")
    if not os.path.exists(exe) or not shutil.which("xargs"):
        return None
    from oracle.harness import bmp_bytes
    from oracle.oraclepy import Oracle
    orc = Oracle()
    n = max(cores, min(cores * 6, int(budget_s * cores / 0.035)))
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        for i in range(n):
            with open(os.path.join(td, f"{i}.bmp"), "wb") as fh:
                fh.write(bmp_bytes(orc.synth(i)))
        names = "\n".join(str(i) for i in range(n))
        t0 = time.perf_counter()
        subprocess.run(["xargs", "-P", str(cores), "-I", "{}", exe, f"-q{q}", os.path.join(td, "{}.bmp"), os.path.join(td, "{}.nhw")],
                       input=names.encode(), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        wall = time.perf_counter() - t0
        done = sum(os.path.exists(os.path.join(td, f"{i}.nhw")) for i in range(n))
    if not done:
        return None
    return {"value": round(done * MPIX_PER_IMAGE / wall, 2), "unit": "Mpixels/s", "images": done, "wall_s": round(wall, 2),
            "how": f"stock `gcc -O3` nhw-enc -q{q}, one process per image, xargs -P {cores}, BMP files on tmpfs (process start and file I/O included)"}


def cpu_baseline(q, budget_s=12.0):
    """Reference encoder on the host cores, bounded sample (about `budget_s` seconds of wall time)."""
    import multiprocessing as mp
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libnhwref_enc.so")) else "port"
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    logical = os.cpu_count() or 1
    phys = physical_cores()
    cores = max(1, min(phys or logical, 64))        # one worker per PHYSICAL core: SMT siblings would flatter nothing and slow every worker
    per = max(4, int(budget_s / 0.03))          # ~30 ms per image per core
    per = min(per, 400)
    jobs = [(kind, list(range(w * per, (w + 1) * per)), q) for w in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    one = _cpu_worker((kind, list(range(40)), q))       # the same encoder alone on one core (BASELINE.md's single-core figure)
    out = {"value": round(n * MPIX_PER_IMAGE / busy, 2), "unit": "Mpixels/s", "cores": cores, "kind": kind,
           "physical_cores": phys, "logical_cpus": logical,
           "one_core": {"value": round(one[0] * MPIX_PER_IMAGE / one[1], 2), "unit": "Mpixels/s", "ms_per_image": round(one[1] / one[0] * 1e3, 2)},
           "sample": f"{n} synthetic 512x512 images (SURVEY 8d generator, seeds 0..{n - 1}), -q{q}, one in-process encoder per physical core, "
                     f"{busy:.1f} s encode time ({wall:.1f} s incl. process start); {'unmodified reference sources + zero-guard allocator' if kind == 'reference' else 'plain-C restatement'}"}
    try:
        v = vanilla_xargs_baseline(q, cores)
        if v:
            out["vanilla_xargs"] = v
    except Exception as ex:     # the baseline is a courtesy figure: never let it take the bench line down
        out["vanilla_xargs"] = {"error": str(ex)[:200]}
    return out


def cpu_baseline_light(q, budget_s=4.0):
    """The sweep legs' own baseline: the same reference encoder at that quality, same method as cpu_baseline(), a smaller sample."""
    import multiprocessing as mp
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libnhwref_enc.so")) else "port"
    logical = os.cpu_count() or 1
    phys = physical_cores()
    cores = max(1, min(phys or logical, 64))
    per = min(max(4, int(budget_s / 0.03)), 160)
    jobs = [(kind, list(range(w * per, (w + 1) * per)), q) for w in range(cores)]
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    n = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    one = _cpu_worker((kind, list(range(24)), q))
    return {"value": round(n * MPIX_PER_IMAGE / busy, 2), "unit": "Mpixels/s", "cores": cores, "kind": kind,
            "one_core": {"value": round(one[0] * MPIX_PER_IMAGE / one[1], 2), "unit": "Mpixels/s", "ms_per_image": round(one[1] / one[0] * 1e3, 2)},
            "sample": f"{n} synthetic images (seeds 0..{n - 1}), -q{q}, one in-process encoder per physical core, {busy:.1f} s encode time"}


def chroma_l1_ms(enc, n, repeats=3):
    """SURVEY 8(d) counts the chroma level-1 coefficients among the fused front kernel's 6 B/pixel, but that analysis runs as two launches of
    the 256 x 256 filterbank kernel on the chroma stream.  Their time for a batch of n images, measured here on its own
    (hipEvents round the two launches the way the encoder makes them -- from the 4:2:0 byte planes of the batch just encoded, nhw_stage_chroma_l1 --
    on the current stream), so that the line can carry a roofline figure that includes it."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    best = None
    for _ in range(repeats + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if enc.lib.nhw_stage_chroma_l1(enc.h, n, st) != 0:
            return None
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b)
        best = t if best is None else min(best, t)
    return best


def hbm_copy_gbs(nbytes=1 << 30, repeats=5):
    """the peak this box achieves on a plain device-to-device copy (read + written bytes over the time between two events on the
    stream the copy runs on), so that `frac` can be read against what the memory system delivers and not only against the 8 TB/s of the data sheet."""
    import torch
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(1); b.copy_(a)
    best = 0.0
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record()
        torch.cuda.synchronize()
        best = max(best, 2 * nbytes / (e0.elapsed_time(e1) / 1e3) / 1e9)
    del a, b
    return best


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def timed_steps(enc, bgr, q, out, steps, warmup, dist, dev, max_over_ranks):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; returns (seconds, max over ranks; stage sums)"""
    import torch
    for _ in range(warmup):
        enc.encode_device(bgr, q, out)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    front_ms = color_ms = pre_ms = 0.0
    tim = None
    for _ in range(steps):
        enc.encode_device(bgr, q, out)
        tim = enc.timing()          # hipEvents recorded on the launch stream; waits for this step's last event
        front_ms += tim.front_ms
        color_ms += tim.color_dwt_ms
        pre_ms += tim.prefilter_ms
    torch.cuda.synchronize()
    local = time.perf_counter() - t0            # this rank's own K steps (reported per rank; the metric uses the max below)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = max_over_ranks(dist, dev, time.perf_counter() - t0)
    timed_steps.last = {"local_s": local, "color_ms": color_ms, "prefilter_ms": pre_ms}
    return dt, front_ms, color_ms, tim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="images per GPU per step (weak scaling)")
    ap.add_argument("--total-images", type=int, default=0, help="strong scaling: ONE job of this many images per step, cut into contiguous per-rank ranges (BASELINE configs[3]: 65536 over 8 GPUs)")
    ap.add_argument("--quality", type=int, default=20)
    ap.add_argument("--sweep", type=str, default="1,8,10,17,23", help="BASELINE configs[2]: quality settings timed after the headline measurement (N=1 only; 8 stands for the slowest band of qualities, 6 .. 9: DESIGN 4.7); '' = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive leg (host buffers through nhw_enc_batch) reported next to the metric")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode leg (BASELINE config 5) reported next to the encode metric")
    ap.add_argument("--no-chroma-l1", action="store_true", help="skip the separate timing of the two chroma level-1 launches (roofline.frac_incl_chroma_l1); the profile scripts use it so that kernel tables hold the encoder's own launches only")
    ap.add_argument("--no-config4-shape", action="store_true", help="skip the leg that times BASELINE config 4's PER-GPU shape (8192 images = 65536 / 8) on this one GPU")
    ap.add_argument("--dry", action="store_true", help="CPU only: rendezvous over gloo, descriptor broadcast, sharding, gather -- no encode (tests of the N>1 plumbing)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC: without it RCCL's buffer exchange fails (also set when this file re-executes itself under torchrun)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"launched with {world} ranks but --gpus {args.gpus}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cpu") if args.dry else torch.device("cuda", local_rank)
    if not args.dry:
        torch.cuda.set_device(local_rank)

    from nhwcodec_amd.dist import broadcast_descriptor, gather_summaries, max_over_ranks, shard_range
    # work descriptor {base, count, quality, seed}: rank 0 decides, everyone receives
    strong = args.total_images > 0
    _, count, q, seed = broadcast_descriptor(dist, dev, 0, args.total_images if strong else args.batch, args.quality, 1234)
    if strong:
        lo, hi = shard_range(0, count, rank, world)       # this rank's contiguous range of the one job
        batch, first_seed = hi - lo, seed + lo
    else:
        batch, first_seed = count, seed + rank * count

    if args.dry:
        # the plumbing alone: every rank reports its range; no encode, no timing claim
        gathered = gather_summaries(dist, dev, batch, first_seed, batch)
        if dist:
            dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "encode Mpixels/s (512x512 RGB batch)", "value": None, "unit": "Mpixels/s", "n_gpus": world, "dry": True,
                              "scaling": "strong" if strong else "weak", "images_per_rank": [int(g[0]) for g in gathered],
                              "first_seed_per_rank": [int(g[1]) for g in gathered]}), flush=True)
        if dist:
            dist.destroy_process_group()
        return

    import nhwcodec_amd
    enc = nhwcodec_amd.Encoder(local_rank, max_batch=batch)
    bgr = enc.synth_device(batch, seed_base=first_seed)              # inputs resident in HBM before timing
    out = enc.alloc_out(batch)
    torch.cuda.synchronize()

    dt, front_ms, color_ms, tim = timed_steps(enc, bgr, q, out, args.steps, args.warmup, dist, dev, max_over_ranks)

    # the two chroma level-1 launches, timed on the planes the headline batch just left (the hook refuses anything but the handle's last whole batch)
    cl1 = chroma_l1_ms(enc, tim.front_images or batch) if (rank == 0 and q >= 17 and not args.no_chroma_l1) else None

    _, sizes, status = out
    ok = int((status == 0).sum().item())
    nbytes = int(sizes.to(torch.int64).sum().item())
    chk = int((sizes.to(torch.int64) * torch.arange(1, batch + 1, device=dev)).sum().item() % (1 << 61))
    gathered = gather_summaries(dist, dev, nbytes, chk, ok)
    head_local = timed_steps.last["local_s"]
    rank_ms = [g[0] / 1e3 for g in gather_summaries(dist, dev, int(head_local / args.steps * 1e6), rank, 0)]      # every rank's own ms per step
    total_per_step = sum(shard_range(0, count, r, world)[1] - shard_range(0, count, r, world)[0] for r in range(world)) if strong else batch * world

    # BASELINE config 2/3: the other quality settings, each under the same timed contract (barrier + synchronize on both sides), on the
    # same resident batch.  The rationed settings (q <= 16) take much longer per step, so they get fewer steps; the count is reported.
    sweep = []
    if world == 1 and args.sweep:
        for sq in [int(v) for v in args.sweep.split(",") if v.strip()]:
            if sq == q:
                continue
            k = args.steps if sq > 16 else max(3, args.steps // 2)
            sdt, sfront, _, stim = timed_steps(enc, bgr, sq, out, k, 1, None, dev, max_over_ranks)
            sok = int((out[2] == 0).sum().item())
            split = timed_steps.last
            sweep.append({"quality": sq, "steps": k, "warmup": 1, "ms_per_step": round(sdt / k * 1e3, 3), "value": round(batch * k * MPIX_PER_IMAGE / sdt, 2), "unit": "Mpixels/s",
                          **({"front_kernels_ms": {"colour + 4:2:0 (k_color)": round(split["color_ms"] / k, 3), "rationed pre-filter (k_low_pre, k_low_mapfix, k_low_chain, k_low_apply, k_low_markrows, k_low_marks)": round(split["prefilter_ms"] / k, 3),
                                                   "level-1 analysis (k_front_plain)": round((sfront - split["color_ms"] - split["prefilter_ms"]) / k, 3)}} if sq <= 16 else
                             {"front_kernels_ms": {("k_front_plain (colour + 4:2:0 + level-1 analysis, no pre-filter)" if sq >= 22 else "k_front_image (colour + 4:2:0 + pre-filter + level-1 analysis)"): round(sfront / k, 3)},
                              "front_achieved_GBs": round(batch * FRONT_BYTES_PER_IMAGE / (sfront / k / 1e3) / 1e9, 1)}),
                          **({"cpu_baseline": cpu_baseline_light(sq)} if not args.no_cpu_baseline else {}),
                          "images_ok": sok, "bytes_out": int(out[1].to(torch.int64).sum().item()),
                          "stage_ms": {"front": round(stim.front_ms, 3), "luma_tail": round(stim.luma_ms, 3), "entropy+container": round(stim.entropy_ms, 3)},
                          # the same roofline figure as the headline's, for this quality's front launch group (q >= 22: no pre-filter in the fused kernel;
                          # q <= 16: colour kernel + the serial pre-filter + the band kernel)
                          "front_roofline_frac": round((stim.front_images or batch) * FRONT_BYTES_PER_IMAGE / (stim.front_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)})
        enc.encode_device(bgr, q, out)      # leave the headline quality's files in the arena for the decode leg
        torch.cuda.synchronize()
        sizes, status = out[1], out[2]

    # BASELINE config 4 is 65536 images over 8 GPUs = 8192 per GPU: what ONE rank of that job costs, on record from a one-GPU box (the 8-GPU run is
    # the driver's).  Twice the headline batch is not twice the time for the kernels that hold a wavefront per image, so it is measured.
    c4_line = None
    if world == 1 and not strong and not args.no_config4_shape:
        n8 = 8192
        enc8 = nhwcodec_amd.Encoder(local_rank, max_batch=n8)
        bgr8 = enc8.synth_device(n8, seed_base=first_seed)
        out8 = enc8.alloc_out(n8)
        legs = []
        for sq, k in ((q, max(2, args.steps)), (10, 2)):
            sdt, sfront, _, stim = timed_steps(enc8, bgr8, sq, out8, k, 1, None, dev, max_over_ranks)
            split = timed_steps.last
            legs.append({"quality": sq, "steps": k, "warmup": 1, "ms_per_step": round(sdt / k * 1e3, 3), "value": round(n8 * k * MPIX_PER_IMAGE / sdt, 2), "unit": "Mpixels/s",
                         "images_ok": int((out8[2] == 0).sum().item()), "front_ms": round(sfront / k, 3),
                         **({"prefilter_ms (k_low_pre .. k_low_marks)": round(split["prefilter_ms"] / k, 3)} if sq <= 16 else {})})
        c4_line = {"workload": f"{n8} synthetic images on ONE GPU = the per-rank share of BASELINE config 4 (65536 over 8), whole encoder, inputs and outputs in HBM", "legs": legs}
        enc8.close()
        del bgr8, out8

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
            rec = 0.0
            for _ in range(args.steps):
                _, dst, dq = dec.decode_device(out[0], offs, sizes, pix)
                dtm = dec.timing()
                rec += dtm.recon_ms
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            ddt = max_over_ranks(dist, dev, time.perf_counter() - t1)
        dec_ok = int((dst == 0).sum().item())
        err = (pix[:64].float() - bgr[:64].float()).pow(2).mean().item()
        dec_line = {"metric": "decode Mpixels/s (512x512 .nhw batch -> BGR24, BASELINE config 5)", "value": round(total_per_step * args.steps * MPIX_PER_IMAGE / ddt, 2),
                    "unit": "Mpixels/s", "ms_per_step": round(ddt / args.steps * 1e3, 3), "files_ok_rank0": dec_ok,
                    "psnr_db_first_64": round(10 * math.log10(255.0 ** 2 / max(err, 1e-9)), 2),
                    "workload": f"the {batch} .nhw files per GPU this run just encoded (-q{q}), decoder arena = encoder arena in HBM",
                    "stage_ms": {"entropy": round(dtm.entropy_ms, 3), "total": round(dtm.total_ms, 3)},
                    # SURVEY 8(d) for config 5: the final reconstruction (level-1 synthesis, both directions, + colour) has 3 B/px of coefficients in and 3 B/px of
                    # BGR out as its algorithmic traffic; it is ONE kernel (k_dec_final: the intermediate plane and the luma bytes stay in LDS), hipEvents on the launch stream
                    "roofline": {"bound": "hbm", "kernel": "k_dec_final (level-1 synthesis in both directions + q>21 corrections + smoothing at the marks + x2 chroma + colour matrix)",
                                 "achieved": round(batch * 1572864 / (rec / 1e3 / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(batch * 1572864 / (rec / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                 "traffic": final_traffic(q, batch)[0], "traffic_unit": "bytes per launch; " + final_traffic(q, batch)[1],
                                 "algorithmic_bytes_per_image": 1572864, "ms_per_launch": round(rec / args.steps, 3)}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            szh = sizes[:256].cpu().numpy(); ar = out[0][:256].cpu().numpy()
            dec_line["cpu_baseline"] = cpu_decode_baseline([ar[i, : int(szh[i])].tobytes() for i in range(min(256, batch))])
        dec.close()

    # the same encoder behind the host-buffer entry point (page-locked BGR in host memory -> chunked H2D on a copy stream next
    # to the encode of the chunk before -> files compacted on the device -> D2H).  PCIe-inclusive, reported apart; never `value`.
    host_line = None
    if world == 1 and not args.no_host_path:
        import numpy as np
        hn = min(batch, 2048)
        pinned = enc.pinned_images(hn)
        pinned[:] = bgr[:hn].cpu().numpy()
        enc.encode(pinned[:64], q)
        times = []
        for _ in range(5):                   # a single shot of a PCIe-bound leg spreads 2x from box to box and run to run: the median of five, with the extremes
            th = time.perf_counter()
            files = enc.encode(pinned, q)
            times.append(time.perf_counter() - th)
        hdt = sorted(times)[len(times) // 2]
        host_line = {"metric": "encode Mpixels/s incl. PCIe both ways (page-locked host BGR in, .nhw bytes on the host out)", "value": round(hn * MPIX_PER_IMAGE / hdt, 2),
                     "unit": "Mpixels/s", "images": hn, "ms": round(hdt * 1e3, 2), "repeats": len(times), "value_min": round(hn * MPIX_PER_IMAGE / max(times), 2),
                     "value_max": round(hn * MPIX_PER_IMAGE / min(times), 2), "pcie_GBs_in": round(hn * 786432 / hdt / 1e9, 1),
                     "bytes_in": hn * 786432, "bytes_out": int(sum(len(f) for f in files)),
                     "same_files_as_resident_run": bool(all(files[i] == bytes(out[0][i, : int(sizes[i])].cpu().numpy().tobytes()) for i in (0, hn // 2, hn - 1)))}
        enc.free_pinned()

    if rank == 0:
        total_images = total_per_step * args.steps
        value = total_images * MPIX_PER_IMAGE / dt
        front_s = front_ms / 1e3 / args.steps
        front_images = tim.front_images or batch      # the batch runs as `parts` sub-batches on their own streams; the events bracket the first one's launch group
        achieved = front_images * FRONT_BYTES_PER_IMAGE / front_s / 1e9
        achieved_all = front_images * FRONT_BYTES_PER_IMAGE / (front_s + (cl1 or 0.0) / 1e3) / 1e9
        traffic, traffic_note = front_traffic(q, front_images)
        copy_gbs = hbm_copy_gbs()
        ev = valu_evidence()
        vk = (ev or {}).get("k_front_image" if q < 22 else "k_front_plain")
        line = {
            "metric": "encode Mpixels/s (512x512 RGB batch)", "value": round(value, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": (f"one job of {count} synthetic 512x512 BGR24 images cut into contiguous ranges over {world} GPU(s)" if strong else
                                    f"batch of {batch} synthetic 512x512 BGR24 images per GPU") + f", -q{q}, whole encoder (BGR in HBM -> .nhw bytes in HBM)",
                       "images_per_gpu": batch if not strong else [shard_range(0, count, r, world)[1] - shard_range(0, count, r, world)[0] for r in range(world)],
                       "images_per_step": total_per_step, "quality": q, "parallelism": f"dp{world} (independent images, no data-path collective)"},
            # `achieved` / `frac`: SURVEY 8(d)'s 6 B/pixel over the time of EVERYTHING that produces them -- the fused front kernel AND the two chroma
            # level-1 launches (0.5 B/pixel of the 6 are their output; they are still launches of their own).  `frac_front_kernel_alone`
            # is the same bytes over the fused kernel's time only, the figure rounds 1-4 reported as `frac`.
            "roofline": {"bound": "valu_issue", "priced_against": "hbm", "kernel": FRONT_KERNEL_NAME,
                         "achieved": round(achieved_all, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_all / HBM_PEAK_GBS, 4),
                         "frac_front_kernel_alone": round(achieved / HBM_PEAK_GBS, 4), "achieved_front_kernel_alone": round(achieved, 1),
                         # what limits the kernel is not the roofline it is priced against: it issues vector instructions most of its cycles (PMC, below)
                         "limiter": "valu_issue", **({"valu_roofline_frac": vk["issue_frac_of_1_per_cu_cycle"], "lane_ops_per_pixel": vk["lane_ops_per_pixel"]} if vk else {}),
                         # the peak confirmed with a device copy on this box (1 GiB, read + written bytes, best of 5) and `frac` against it
                         "hbm_copy_measured": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved_all / copy_gbs, 4),
                         "traffic": traffic, "traffic_unit": "bytes per launch group; " + traffic_note,
                         "algorithmic_bytes_per_launch": front_images * FRONT_BYTES_PER_IMAGE, "images_per_launch": front_images,
                         # SURVEY 8(d) also states its >= 0.50 target on the READ side alone (n x 786 432 B of BGR / t): half of `frac`
                         "frac_on_reads_only": round(front_images * 786432 / front_s / 1e9 / HBM_PEAK_GBS, 4),
                         "reads_only_target_note": ("north_star's literal target, >= 0.50 of the HBM READ roofline on this kernel, means 4 TB/s of reads next to as many bytes written = 8 TB/s of "
                                                    f"combined traffic, {8000.0 / copy_gbs:.2f} x what a plain device copy reaches on this box (hbm_copy_measured): out of reach for any kernel; the figure held to is 0.50 of the 6 B/pixel (<= 1.61 ms per 4096 images)"),
                         # the same bytes over the front group PLUS the two chroma level-1 launches whose output the 6 B/pixel include (measured apart, see chroma_l1_ms)
                         "chroma_l1_in_frac": bool(cl1),
                         **({"frac_incl_chroma_l1": round(front_images * FRONT_BYTES_PER_IMAGE / ((front_s + cl1 / 1e3)) / 1e9 / HBM_PEAK_GBS, 4), "chroma_l1_ms": round(cl1, 3)} if cl1 else {}), "sub_batches": tim.parts, "algorithmic_bytes_per_image": FRONT_BYTES_PER_IMAGE, "ms_per_launch_group": round(front_s * 1e3, 3)},
            "stage_ms": {"front": round(tim.front_ms, 3), "luma_tail incl. the join with the chroma stream in front of the luma quantiser": round(tim.luma_ms, 3), "chroma left over behind it (0 with that join)": round(tim.chroma_ms, 3),
                         "entropy+container": round(tim.entropy_ms, 3), "total": round(tim.total_ms, 3)},
            "images_ok": [int(g[2]) for g in gathered], "bytes_out": [int(g[0]) for g in gathered],
            "ms_per_step_per_rank": [round(v, 3) for v in rank_ms],      # each rank's own clock over its K steps; ms_per_step is the max-over-ranks figure with the barriers
            "world_size_from_collective": len(gathered),                # what the all-gather (RCCL for N > 1) saw, next to n_gpus from the launcher
        }
        if sweep:
            line["sweep"] = sweep
        if ev:
            line["roofline"]["valu_pmc"] = ev
        if c4_line:
            line["config4_per_gpu_shape"] = c4_line
        if host_line:
            line["host_path"] = host_line
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
