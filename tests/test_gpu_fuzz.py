"""Bounded slices of the developer fuzzers in the driver's GPU run (the classes of bug they found in earlier rounds -- chroma pair-mark
rows whose running index shifts, a cross-XCD race between bands of one image, the rationed pre-filter's rare schedules -- stay guarded):
  * tools/dev/gpu_fuzz_classes.py: images of mixed classes through the encoder and back through the decoder at q 1, 10, 20, 23 against the oracle;
  * tools/dev/gpu_hazard_check.py: the images of a synthetic batch whose chroma mark walk ends a row in a pair mark, against the oracle;
  * bench.py's strong-scaling split at config 4's per-GPU shape (8192 images on one GPU) through the real launcher path.
Seeded and time-boxed: about a minute of GPU + host time in all."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpu_fuzz_classes import dec_chunk, make, want_chunk  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 10, 20, 23])
def test_mixed_class_images_encode_and_decode_like_the_oracle(q):
    from concurrent.futures import ProcessPoolExecutor
    import nhwcodec_amd as na
    n, first = 48, 20000 + 100 * q                         # fresh seeds per quality; earlier rounds ran 0..9255 by hand
    seeds = list(range(first, first + n))
    imgs = np.stack([make(s) for s in seeds])
    enc = na.Encoder(0, n)
    files = enc.encode(imgs, q)
    enc.close()
    dec = na.Decoder(0, n)
    px, qs = dec.decode(files)
    dec.close()
    workers = min(24, os.cpu_count() or 4)
    with ProcessPoolExecutor(max_workers=workers) as ex:
        want = [h for part in ex.map(want_chunk, [(q, seeds[i:i + 4]) for i in range(0, n, 4)]) for h in part]
        dwant = [h for part in ex.map(dec_chunk, [files[i:i + 4] for i in range(0, n, 4)]) for h in part]
    bad = [seeds[i] for i in range(n) if hashlib.sha1(files[i]).hexdigest() != want[i]]
    dbad = [seeds[i] for i in range(n) if hashlib.sha1(px[i].tobytes()).hexdigest() != dwant[i] or qs[i] != q]
    assert not bad, f"q{q}: .nhw bytes differ from the oracle for seeds {bad[:8]}"
    assert not dbad, f"q{q}: decoded pixels differ from the oracle's decoder for seeds {dbad[:8]}"


@pytest.mark.gpu
@pytest.mark.parametrize("q", [5, 8, 11, 14])
def test_noise_and_hard_edges_at_the_rationed_qualities(q):
    """White noise and hard-edge rectangles at quality 1..16: the images on which the quantisers' rare rules fire (values beyond +-127, the
    `quant4` pushes out of and into such values, rationed low bits) -- a slice of tools/dev/gpu_fuzz_noise.py, whose full run found a pusher
    that crossed 127 in round 3."""
    from concurrent.futures import ProcessPoolExecutor
    import nhwcodec_amd as na
    from oracle.harness import class_image
    from gpu_fuzz_noise import want_chunk as noise_want
    items = [(k, s) for k in ("noise", "blocks") for s in range(50 + q, 58 + q)]
    imgs = np.stack([class_image(k, s) for k, s in items])
    enc = na.Encoder(0, len(items))
    got = [hashlib.sha1(f).hexdigest() for f in enc.encode(imgs, q)]
    enc.close()
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        want = [h for part in ex.map(noise_want, [(q, items[i:i + 2]) for i in range(0, len(items), 2)]) for h in part]
    bad = [items[i] for i in range(len(items)) if got[i] != want[i]]
    assert not bad, f"q{q}: {bad} differ from the oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("q,seed", [(20, 31000), (10, 32000)])
def test_images_with_shifted_chroma_mark_rows(q, seed):
    """nhw_encoder.c:2372-2427: the index into the chroma LL1 block is not reset per row; a pair mark in a row's last column shifts every
    later row.  The kernel finds the shifts as a fixed point; the images of a batch that have such rows are compared with the oracle."""
    import torch
    import nhwcodec_amd
    from oracle.oraclepy import Oracle
    n = 2048
    e = nhwcodec_amd.Encoder(0, max_batch=n)
    bgr = e.synth_device(n, seed_base=seed)
    o, sizes, status = e.encode_device(bgr, q)
    torch.cuda.synchronize()
    sz = sizes.cpu().numpy()
    hits = []
    for i in range(n):
        m = np.zeros(32, np.int32)
        assert e.lib.nhw_debug_read(e.h, 54, i, ctypes.c_void_p(m.ctypes.data), ctypes.c_size_t(128)) == 0      # B_META: the image's scalar state
        if m[31]:
            hits.append(i)
    orc = Oracle()
    t0 = time.time()
    checked = 0
    for i in hits[:12]:
        if time.time() - t0 > 20:
            break
        assert o[i, : sz[i]].cpu().numpy().tobytes() == orc.encode(orc.synth(seed + i), q), f"image {i} of the batch (seed {seed + i})"
        checked += 1
    e.close()
    # such rows are rare (a few images in a thousand): the test is only meaningful if the batch holds some
    assert checked or not hits, "no image checked"


@pytest.mark.gpu
def test_bench_strong_split_at_config_4_shape():
    """BASELINE config 4 is 65536 images over 8 GPUs = 8192 per GPU: that per-GPU shape through bench.py's strong-scaling path (descriptor
    broadcast, contiguous ranges, gather) on the one GPU there is."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--total-images", "8192", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-decode", "--no-host-path", "--sweep="], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 1 and line["world_size_from_collective"] == 1
    assert line["config"]["images_per_step"] == 8192 and line["images_ok"] == [8192]
    assert line["value"] > 0 and len(line["ms_per_step_per_rank"]) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("q", [16, 17, 20, 23])
def test_code_book_overflow_status_matches_the_oracle(q):
    """The reference exits with -1 when the packetiser's code book overflows (compress_pixel.c:234,270,271); the C ABI reports
    NHW_E_CODEBOOK for THAT image and encodes its neighbours in the batch.  Image: tools/dev/gpu_fuzz_classes.py seed 50431 (overflows from
    quality 17 on, not at 16); oracle side: test_code_book_overflow_is_the_reference_exit."""
    import nhwcodec_amd
    from oracle.oraclepy import Oracle
    from gpu_fuzz_classes import make, encode_with_status
    o = Oracle()
    imgs = np.stack([o.synth(3), make(50431), o.synth(4), make(50430)])
    enc = nhwcodec_amd.Encoder(0, len(imgs))
    files, status = encode_with_status(enc, imgs, q)
    if q > 16:
        with pytest.raises(nhwcodec_amd.NhwError):                 # the host convenience path refuses the batch
            enc.encode(imgs, q)
    enc.close()
    for i, im in enumerate(imgs):
        try:
            want, rc = o.encode(im, q), 0
        except RuntimeError as ex:
            want, rc = b"", int(str(ex).split("rc=")[-1])
        assert status[i] == rc and files[i] == want, f"image {i}: status {status[i]} (oracle {rc})"
    assert status[1] == (nhwcodec_amd.NHW_E_CODEBOOK if q > 16 else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [7, 13, 16, 20, 23])
def test_production_batches_never_read_stale_planes(q):
    """The production launch sequence leaves out stores nothing reads (the transposed copy of the LL quadrant, the transposed first-direction
    planes of the chroma analyses, the natural-orientation copies of the syntheses ...), which the stage checks do not: their kernels write
    every plane.  So the work planes are filled with 0x7F7F garbage before a production batch: the files must still equal the oracle's --
    a cell that is read without having been written in THIS batch shows up here."""
    import nhwcodec_amd as na
    from oracle.oraclepy import Oracle
    orc = Oracle()
    seeds = [3000 + 17 * q + i for i in range(6)]
    imgs = np.stack([orc.synth(s) for s in seeds[:4]] + [make(seeds[4]), make(seeds[5])])
    enc = na.Encoder(0, len(imgs))
    enc.lib.nhw_debug_fill.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int]
    enc.encode(imgs, q)                                     # a batch before, as in a long-running encoder
    Q = 65536
    for buf, nbytes in ((0, 8 * Q), (1, 8 * Q), (4, 2 * Q), (5, 2 * Q)):     # JPEG, PROC, CJPEG, CPROC (nhw_ws.h)
        assert enc.lib.nhw_debug_fill(enc.h, buf, 0x7F, nbytes, len(imgs)) == 0
    got = enc.encode(imgs, q)
    enc.close()
    for i in range(len(imgs)):
        assert got[i] == orc.encode(imgs[i], q), f"q{q} image {i}: a production batch read a plane cell it had not written"
