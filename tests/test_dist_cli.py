"""CPU tests: multi-process sharding logic over gloo (world_size 2), and the nhw-enc command line contract."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_batch():
    from nhwcodec_amd.dist import shard_range
    for count in (1, 7, 8, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            rs = [shard_range(5, count, r, world) for r in range(world)]
            assert rs[0][0] == 5 and rs[-1][1] == 5 + count
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nhwcodec_amd.dist import broadcast_descriptor, gather_summaries, max_over_ranks, shard_range
    dev = torch.device("cpu")
    # only rank 0 knows the job; everyone must end up with it
    desc = broadcast_descriptor(dist, dev, *( (100, 4097, 20, 77) if rank == 0 else (0, 0, 0, 0)))
    lo, hi = shard_range(desc[0], desc[1], rank, world)
    summ = gather_summaries(dist, dev, bytes_out=(hi - lo) * 10, checksum=lo, images_ok=hi - lo)
    tmax = max_over_ranks(dist, dev, 1.0 + rank)
    q.put((rank, desc, (lo, hi), summ, tmax))
    dist.destroy_process_group()


def test_gloo_world2_descriptor_broadcast_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, d0, s0, g0, t0), (r1, d1, s1, g1, t1) = res
    assert d0 == d1 == (100, 4097, 20, 77)
    assert s0 == (100, 2149) and s1 == (2149, 4197)
    assert g0 == g1 == [(20490, 100, 2049), (20480, 2149, 2048)]
    assert t0 == t1 == 2.0


def _bench_dry(*extra):
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", *extra], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout       # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in the environment re-executes itself under torch.distributed.run: two ranks meet
    (gloo in this CPU test), the line says n_gpus 2 and every rank reports its share."""
    d = _bench_dry("--gpus", "2")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["images_per_rank"] == [4096, 4096]
    assert d["first_seed_per_rank"][1] - d["first_seed_per_rank"][0] == 4096      # disjoint seed ranges


def test_bench_strong_split_is_baseline_config_4():
    """--total-images 65536 over 8 ranks = 8192 contiguous images per rank (BASELINE configs[3]); here 3 ranks to keep the CPU test light"""
    d = _bench_dry("--gpus", "3", "--total-images", "65536")
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and sum(d["images_per_rank"]) == 65536
    assert max(d["images_per_rank"]) - min(d["images_per_rank"]) <= 1
    seeds = d["first_seed_per_rank"]
    assert [seeds[i + 1] - seeds[i] for i in range(2)] == d["images_per_rank"][:2]


# ---------------------------------------------------------------- CLI contract (no GPU needed for argument handling)
CLI = os.path.join(ROOT, "tools", "nhw-enc")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "nhw-enc")


def _run(exe, *a):
    p = subprocess.run([exe, *a], capture_output=True, text=True)
    return p.returncode, p.stdout, p.stderr


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    return CLI


@pytest.mark.parametrize("args", [["-q99", "a.bmp", "b.nhw"], ["-qx", "a.bmp", "b.nhw"], ["-z", "a", "b"], ["only_one"], ["same", "same"], ["-h"]])
def test_cli_argument_handling_matches_reference_binary(cli, args):
    if not os.path.exists(REF_CLI):
        pytest.skip("reference binary not built")
    rc, out, err = _run(cli, *args)
    rrc, rout, rerr = _run(REF_CLI, *args)
    assert rc == rrc
    assert out.splitlines()[:1] == rout.splitlines()[:1] and err == rerr


def test_cli_rejects_unimplemented_quality_loudly(cli):
    rc, out, err = _run(cli, "-q0", "a.bmp", "b.nhw")       # accepted by the reference's argument loop, but it has no tables for it
    assert rc == 3 and "not implemented" in err


@pytest.mark.parametrize("args,msg", [(["--devices", "foo"], "--devices wants"), (["--devices", "0,"], "--devices wants"), (["--devices", ","], "--devices wants"),
                                      (["--devices", "0,x"], "--devices wants"), (["--devices", ",".join(["0"] * 17)], "at most 16"),
                                      (["--chunk", "x"], "--chunk wants"), (["--chunk", "0"], "--chunk wants"), (["--chunk", "99999"], "--chunk wants")])
def test_cli_rejects_malformed_device_lists_and_chunks(cli, tmp_path, args, msg):
    """A malformed --devices / --chunk is an error with a message and exit code 1 (before any GPU work), not a silent `device 0` / clamp."""
    rc, out, err = _run(cli, *args, "--synthetic", "1", "--outdir", str(tmp_path))
    assert rc == 1 and msg in err, (rc, err)
    assert not list(tmp_path.iterdir())


def test_cli_bad_bmp_exit_codes_match_reference(cli, tmp_path):
    if not os.path.exists(REF_CLI):
        pytest.skip("reference binary not built")
    cases = {"nosig.bmp": b"XX" + bytes(60), "short.bmp": b"BM" + bytes(10),
             "wrongsize.bmp": b"BM" + bytes(8) + (54).to_bytes(4, "little") + (40).to_bytes(4, "little") + (256).to_bytes(4, "little") + (256).to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little") + bytes(4)}
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        rc, out, _ = _run(cli, str(p), str(tmp_path / "o.nhw"))
        rrc, rout, _ = _run(REF_CLI, str(p), str(tmp_path / "r.nhw"))
        assert (rc, out) == (rrc, rout), name


@pytest.mark.gpu
def test_cli_end_to_end_files(cli, oracle, tmp_path):
    """nhw-enc file in -> file out equals the oracle, incl. a top-down (negative height) BMP and a 108-byte V4 header."""
    import struct
    from oracle.harness import bmp_bytes
    img = oracle.synth(21)
    (tmp_path / "a.bmp").write_bytes(bmp_bytes(img))
    assert _run(cli, "-q20", str(tmp_path / "a.bmp"), str(tmp_path / "a.nhw"))[0] == 0
    assert (tmp_path / "a.nhw").read_bytes() == oracle.encode(img, 20)
    # negative height: rows stored top-down -> the reader flips them
    hdr = struct.pack("<2sIHHIIiiHHIIiiII", b"BM", 54 + img.size, 0, 0, 54, 40, 512, -512, 1, 24, 0, img.size, 0, 0, 0, 0)
    (tmp_path / "b.bmp").write_bytes(hdr + img.tobytes())
    assert _run(cli, "-q23", str(tmp_path / "b.bmp"), str(tmp_path / "b.nhw"))[0] == 0
    assert (tmp_path / "b.nhw").read_bytes() == oracle.encode(img[::-1], 23)
    # V4 header (108 bytes) with a larger data offset, truncated pixel data (short read -> zero tail)
    off = 14 + 108 + 20
    hdr = struct.pack("<2sIHHI", b"BM", off + img.size, 0, 0, off) + struct.pack("<IiiHHI", 108, 512, 512, 1, 24, 0) + bytes(108 - 20) + bytes(20)
    cut = img.copy(); cut.reshape(-1)[-30000:] = 0
    (tmp_path / "c.bmp").write_bytes(hdr + img.tobytes()[:-30000])
    assert _run(cli, str(tmp_path / "c.bmp"), str(tmp_path / "c.nhw"))[0] == 0
    assert (tmp_path / "c.nhw").read_bytes() == oracle.encode(cut, 20)
    # batch directory mode
    d = tmp_path / "dir"; d.mkdir()
    for s in (1, 2, 3):
        (d / f"i{s}.bmp").write_bytes(bmp_bytes(oracle.synth(s)))
    assert _run(cli, "-q21", "--batch", str(d))[0] == 0
    for s in (1, 2, 3):
        assert (d / f"i{s}.nhw").read_bytes() == oracle.encode(oracle.synth(s), 21)
    # synthetic mode: SURVEY 8d images generated on the device, one file per seed; --gpus 1 = one worker thread on device 0
    o = tmp_path / "syn"; o.mkdir()
    assert _run(cli, "-q10", "--gpus", "1", "--synthetic", "5", "--seed", "40", "--outdir", str(o))[0] == 0
    for s in range(40, 45):
        assert (o / f"synth_{s}.nhw").read_bytes() == oracle.encode(oracle.synth(s), 10)
    rc, _, err = _run(cli, "--gpus", "99", "--synthetic", "2", "--outdir", str(o))
    assert rc == 1 and "device(s) visible" in err


@pytest.mark.gpu
def test_cli_two_workers_on_one_device_share_the_queue(cli, oracle, tmp_path):
    """`--devices 0,0`: two worker threads, two encoder handles on device 0, one queue of chunks behind a mutex (what `--gpus G` runs with G > 1,
    on the one GPU there is): every file equals the single-worker run's and the oracle's, whichever worker took its chunk -- at a rationed
    quality and at the headline one, so that both front kernels and their per-handle attributes run twice in one process."""
    for q, n in ((20, 40), (10, 24)):
        one = tmp_path / f"one{q}"; two = tmp_path / f"two{q}"; one.mkdir(); two.mkdir()
        assert _run(cli, f"-q{q}", "--gpus", "1", "--chunk", "7", "--synthetic", str(n), "--seed", "900", "--outdir", str(one))[0] == 0
        rc, out, err = _run(cli, f"-q{q}", "--devices", "0,0", "--chunk", "7", "--synthetic", str(n), "--seed", "900", "--outdir", str(two))
        assert rc == 0, err
        for s in range(900, 900 + n):
            a, b = (one / f"synth_{s}.nhw").read_bytes(), (two / f"synth_{s}.nhw").read_bytes()
            assert a == b, f"q{q} seed {s}: the two-worker run differs from the one-worker run"
        for s in (900, 900 + n // 2, 900 + n - 1):
            assert (two / f"synth_{s}.nhw").read_bytes() == oracle.encode(oracle.synth(s), q)
    rc, _, err = _run(cli, "--devices", "0,7", "--synthetic", "2", "--outdir", str(tmp_path))
    assert rc == 1 and "device(s) visible" in err


# ---------------------------------------------------------------- nhw-dec
DEC_CLI = os.path.join(ROOT, "tools", "nhw-dec")
REF_DEC_CLI = os.path.join(ROOT, "oracle", "_ref", "nhw-dec")


@pytest.fixture(scope="module")
def dec_cli():
    if not os.path.exists(DEC_CLI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    return DEC_CLI


def test_dec_cli_usage_matches_reference_binary(dec_cli):
    if not os.path.exists(REF_DEC_CLI):
        pytest.skip("reference binary not built")
    for args in ([], ["only_one.nhw"]):
        rc, out, err = _run(dec_cli, *args)
        rrc, rout, rerr = _run(REF_DEC_CLI, *args)
        assert rc == rrc == 0
        assert out.splitlines()[:5] == rout.splitlines()[:5] and err == rerr


@pytest.mark.gpu
def test_dec_cli_end_to_end_files(dec_cli, oracle, tmp_path):
    """nhw-dec file in -> BMP out equals the oracle's BMP (= the reference decoder's, by the golden digests); batch mode; a foreign file"""
    import hashlib
    import json
    import shutil
    gold = os.path.join(ROOT, "tests", "golden", "dec")
    man = json.load(open(os.path.join(gold, "manifest.json")))
    for name in ("q20_0.nhw", "q07_0.nhw", "q23_0.nhw"):
        shutil.copy(os.path.join(gold, name), tmp_path / name)
        assert _run(dec_cli, str(tmp_path / name), str(tmp_path / (name + ".bmp")))[0] == 0
        assert hashlib.sha256((tmp_path / (name + ".bmp")).read_bytes()).hexdigest() == man[name]["bmp_sha256"]
    d = tmp_path / "dir"; d.mkdir()
    names = ["q01_0.nhw", "q10_0.nhw", "q16_4.nhw", "q20_blocks.nhw"]
    for n in names:
        shutil.copy(os.path.join(gold, n), d / n)
    assert _run(dec_cli, "--batch", str(d))[0] == 0
    for n in names:
        assert hashlib.sha256((d / (n[:-4] + ".bmp")).read_bytes()).hexdigest() == man[n]["bmp_sha256"]
    (tmp_path / "junk.nhw").write_bytes(b"BM" + bytes(500))
    rc, out, _ = _run(dec_cli, str(tmp_path / "junk.nhw"), str(tmp_path / "junk.bmp"))
    assert rc == 3 and "Not an .nhw file" in out and not (tmp_path / "junk.bmp").exists()


# ---------------------------------------------------------------- pictures larger than 512x512 as independent tiles
def _big_bmp(img, negative_height=False):
    import struct
    h, w = img.shape[:2]
    return struct.pack("<2sIHHIIiiHHIIiiII", b"BM", 54 + img.size, 0, 0, 54, 40, w, -h if negative_height else h, 1, 24, 0, img.size, 0, 0, 0, 0) + img.tobytes()


def test_tile_helpers_round_trip_and_reject_odd_sizes():
    import nhwcodec_amd as na
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, (1024, 1536, 3), dtype=np.uint8)
    tiles, (ny, nx) = na.tile_images(big)
    assert (ny, nx) == (2, 3) and tiles.shape == (6, 512, 512, 3)
    assert np.array_equal(tiles[4], big[512:, 512:1024])            # tile (1, 1)
    assert np.array_equal(na.untile_images(tiles, ny, nx), big)
    for shape in [(512, 700, 3), (300, 512, 3), (1024, 1024, 4)]:
        with pytest.raises(na.NhwError):
            na.tile_images(np.zeros(shape, np.uint8))


def test_cli_tiles_rejects_sizes_that_are_not_multiples_of_512(cli, tmp_path):
    """the size check comes before any GPU work: same "invalid image file." line and exit code as a wrong-size BMP in the one-tile path"""
    (tmp_path / "odd.bmp").write_bytes(_big_bmp(np.zeros((512, 600, 3), np.uint8)))
    rc, out, err = _run(cli, "--tiles", str(tmp_path / "odd.bmp"), str(tmp_path / "odd"))
    assert rc == (-16) % 256 and "invalid image file." in out and "multiples of 512" in err


@pytest.mark.gpu
def test_cli_tiles_encode_and_join(cli, dec_cli, oracle, tmp_path):
    """nhw-enc --tiles: every tile of a 1024x1536 picture is the file the oracle writes for that crop; nhw-dec --tiles puts the oracle's
    decodes of the tiles side by side; a top-down file is flipped as a whole first.  The python helpers give the same files."""
    import nhwcodec_amd as na
    big = na.untile_images(np.stack([oracle.synth(700 + t) for t in range(6)]), 2, 3)
    (tmp_path / "big.bmp").write_bytes(_big_bmp(big))
    rc, out, _ = _run(cli, "-q20", "--tiles", str(tmp_path / "big.bmp"), str(tmp_path / "t"))
    assert rc == 0 and "2 x 3 tiles" in out
    files = []
    for r in range(2):
        for c in range(3):
            f = (tmp_path / f"t_y{r}_x{c}.nhw").read_bytes()
            assert f == oracle.encode(big[512 * r:512 * r + 512, 512 * c:512 * c + 512], 20), (r, c)
            files.append(f)
    assert _run(dec_cli, "--tiles", "2", "3", str(tmp_path / "t"), str(tmp_path / "joined.bmp"))[0] == 0
    joined = (tmp_path / "joined.bmp").read_bytes()
    want = na.untile_images(np.stack([oracle.decode(f)[0] for f in files]), 2, 3)
    assert len(joined) == 54 + want.size and joined[54:] == want.tobytes()
    assert int.from_bytes(joined[18:22], "little") == 1536 and int.from_bytes(joined[22:26], "little") == 1024
    # top-down file
    (tmp_path / "neg.bmp").write_bytes(_big_bmp(big, negative_height=True))
    assert _run(cli, "-q23", "--tiles", str(tmp_path / "neg.bmp"), str(tmp_path / "n"))[0] == 0
    assert (tmp_path / "n_y0_x2.nhw").read_bytes() == oracle.encode(big[::-1][:512, 1024:], 23)
    # python
    e = na.Encoder(0, 6)
    got, shape = e.encode_tiled(big, 20)
    assert shape == (2, 3) and got == files
    d = na.Decoder(0, 6)
    assert np.array_equal(d.decode_tiled(got, 2, 3), want)
    e.close(); d.close()


# ---------------------------------------------------------------- a tar archive as the batch container
def test_cli_tar_missing_archive_fails_before_any_gpu_work(cli, dec_cli, tmp_path):
    rc, out, _ = _run(cli, "--tar", str(tmp_path / "none.tar"), str(tmp_path / "o.tar"))
    assert rc == 255 and "Could not open file" in out
    assert _run(dec_cli, "--tar", str(tmp_path / "none.tar"), str(tmp_path / "o.tar"))[0] == 1


@pytest.mark.gpu
def test_cli_tar_round_trip(cli, dec_cli, oracle, tmp_path):
    """nhw-enc --tar: the .bmp members of a ustar archive (others are passed over, a broken one is reported and left out) come back as .nhw
    members that equal the oracle's files, in order; nhw-dec --tar turns them into the BMPs the oracle's decoder gives."""
    import io, tarfile
    from oracle.harness import bmp_bytes
    imgs = {f"dir/img{s}.bmp": oracle.synth(900 + s) for s in range(5)}
    src = tmp_path / "in.tar"
    with tarfile.open(src, "w", format=tarfile.USTAR_FORMAT) as tf:
        def add(name, data):
            ti = tarfile.TarInfo(name); ti.size = len(data); tf.addfile(ti, io.BytesIO(data))
        add("readme.txt", b"not an image\n" * 50)
        for name, im in imgs.items():
            add(name, bmp_bytes(im))
        add("broken.bmp", b"BM" + bytes(100))
    rc, out, err = _run(cli, "-q21", "--tar", str(src), str(tmp_path / "out.tar"))
    assert rc == 1 and "5 image(s) encoded" in out and "broken.bmp" in err          # the broken member makes the exit code 1, the others are there
    with tarfile.open(tmp_path / "out.tar") as tf:
        members = tf.getmembers()
        assert [m.name for m in members] == [n[:-4] + ".nhw" for n in imgs]
        files = [tf.extractfile(m).read() for m in members]
    for f, im in zip(files, imgs.values()):
        assert f == oracle.encode(im, 21)
    assert _run(dec_cli, "--tar", str(tmp_path / "out.tar"), str(tmp_path / "back.tar"))[0] == 0
    with tarfile.open(tmp_path / "back.tar") as tf:
        members = tf.getmembers()
        assert [m.name for m in members] == list(imgs)
        for m, f in zip(members, files):
            assert tf.extractfile(m).read() == oracle.bmp_header() + oracle.decode(f)[0].tobytes()
