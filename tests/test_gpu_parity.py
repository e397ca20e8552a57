"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI of libnhwhip.so, against the CPU
oracle on the same seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full batch
size -- through size-independent properties.  Integer/byte work: the bar is bit-exact everywhere."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from oracle.harness import class_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- CPU-side checks of the boundary
def test_c_abi_exports_every_declared_symbol():
    import nhwcodec_amd
    if not os.path.exists(nhwcodec_amd.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(nhwcodec_amd.LIB_PATH)      # loading needs no GPU
    for header, must in (("nhw_hip.h", {"nhw_enc_create", "nhw_enc_batch", "nhw_enc_batch_device", "nhw_enc_destroy"}),
                         ("nhw_hip_debug.h", {"nhw_debug_stop_after", "nhw_debug_read", "nhw_dec_debug_read"})):   # the boundary, and the hooks the tests call
        hdr = open(os.path.join(ROOT, "include", header)).read()
        names = set(re.findall(r"\b(nhw_[a-z_0-9]+)\s*\(", hdr))
        assert must <= names
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"
    # ... and nothing the tests call is missing from the headers
    declared = set()
    for header in ("nhw_hip.h", "nhw_hip_debug.h"):
        declared |= set(re.findall(r"\b(nhw_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", header)).read()))
    used = set()
    for dp, _, fs in os.walk(os.path.join(ROOT, "tests")):
        for f in fs:
            if f.endswith(".py"):
                used |= set(re.findall(r"\.lib\.(nhw_[a-z_0-9]+)", open(os.path.join(dp, f)).read()))
    assert used <= declared, f"called by the tests but declared in no header: {sorted(used - declared)}"


def test_product_never_touches_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "nhwcodec_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.replace("canonical-oracle semantics", ""), f"{f} refers to the oracle"


def test_no_gpu_means_loud_failure():
    import torch
    import nhwcodec_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nhwcodec_amd.NhwError):
        nhwcodec_amd.Encoder(0, 1)


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def enc():
    import nhwcodec_amd
    e = nhwcodec_amd.Encoder(0, max_batch=64)
    yield e
    e.close()


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 9, 16, 17, 18, 19, 20])
def test_colour_exhaustive_all_2_24_triples(enc, oracle, q):
    """a1: every (b0,b1,b2) triple once (64 images x 262144 pixels): Y in double, chroma through float (q >= 17; exact integer form
    for q >= 20), integer BT.601 scaled by the quality table below (the colour stage covers every quality)."""
    import torch
    idx = np.arange(1 << 24, dtype=np.uint32)
    imgs = np.stack([(idx >> 16).astype(np.uint8), (idx >> 8).astype(np.uint8), idx.astype(np.uint8)], axis=1).reshape(64, 512, 512, 3)
    d = _cuda(imgs)
    y = torch.empty((64, 512 * 512), dtype=torch.int16, device="cuda")
    u = torch.empty((64, 65536), dtype=torch.uint8, device="cuda")
    v = torch.empty_like(u)
    assert enc.lib.nhw_stage_color(enc.h, d.data_ptr(), 64, q, y.data_ptr(), u.data_ptr(), v.data_ptr(), None) == 0
    torch.cuda.synchronize()
    y, u, v = y.cpu().numpy(), u.cpu().numpy(), v.cpu().numpy()
    for i in range(64):
        oy, ou, ov = oracle.color(imgs[i], q)
        assert np.array_equal(y[i], oy), f"Y image {i}"
        assert np.array_equal(u[i], ou) and np.array_equal(v[i], ov), f"chroma image {i}"


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 6, 7, 8, 9, 10, 14, 16])
def test_prefilter_matches_oracle(enc, oracle, q):
    """The pre-filter is a stage of its own for quality 1..16 (k_low_machine + k_low_marks + the chroma pass, the kernels the encoder runs,
    behind nhw_stage_prefilter); for 17..21 it lives inside the fused front kernel k_front_image (test_fused_front_matches_oracle)."""
    import torch
    imgs = [oracle.synth(3), class_image("noise", 1), class_image("blocks", 2), class_image("flat")]
    ys = np.stack([oracle.color(im, q)[0] for im in imgs])
    d = _cuda(ys)
    assert enc.lib.nhw_stage_prefilter(enc.h, d.data_ptr(), len(imgs), q, None) == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    for i in range(len(imgs)):
        assert np.array_equal(got[i], oracle.prefilter(ys[i], q)), f"image {i}"


def _worst_case_plane(U, L, py, px, negate):
    """The plane that drives the 2-D low-pass [-1 2 6 2 -1] x [-1 2 6 2 -1] of the level-1 analysis to its extreme: U where the tap weight of
    the outputs of lattice phase (py, px) is positive, -L where it is negative (negate: the other way round -> the most negative sum).  The
    sign pattern of the 1-D taps around an output at c = 4m + p is (+ + - +) from c on, period 4."""
    sgn = np.array([1, 1, -1, 1])
    sy, sx = sgn[(np.arange(512) - py) % 4], sgn[(np.arange(512) - px) % 4]
    pos = (sy[:, None] * sx[None, :] > 0) ^ bool(negate)
    return np.where(pos, U, -L).astype(np.int16).reshape(-1)


@pytest.mark.gpu
def test_level1_analysis_domain_is_checked(enc, oracle):
    """nhw_stage_analysis(size 512) runs the encoder's level-1 kernel, whose filter passes are packed 16-bit arithmetic: it equals the
    reference's `int` accumulators (filters.c:203-287, 346-386; wavelet_filterbank.c:52) exactly on the stated domain -- adversarial planes
    AT the bound, every phase of the output lattice, both signs -- and refuses planes one past it with NHW_E_ARG instead of answering with
    a plane that differs (the round-4 hole: 104 * 315 + 40 * 4 = 32920 wraps)."""
    import torch
    NHW_E_ARG = -4
    at_bound = [(313, 4), (314, 0), (276, 100), (4, 313), (0, 314), (259, 4)]
    for U, L in at_bound:
        assert 104 * U + 40 * L <= 32720 and 104 * L + 40 * U <= 32720
        planes = np.stack([_worst_case_plane(U, L, py, px, neg) for py in (0, 2) for px in (0, 2) for neg in (0, 1)])
        j, p = _cuda(planes), _cuda(np.zeros_like(planes))
        assert enc.lib.nhw_stage_analysis(enc.h, j.data_ptr(), p.data_ptr(), len(planes), 512 * 512, 512, 512, 0, None) == 0, (U, L)
        torch.cuda.synchronize()
        gj, gp = j.cpu().numpy(), p.cpu().numpy()
        for i in range(len(planes)):
            oj, op = oracle.analysis(planes[i], 512, 512, 0)
            assert np.array_equal(gp[i], op), f"coefficients U={U} L={L} plane {i}"
            assert np.array_equal(gj[i].reshape(512, 512)[:256, :256], oj.reshape(512, 512)[:256, :256]), f"LL copy U={U} L={L} plane {i}"
    for U, L in [(314, 4), (315, 0), (4, 314), (32767, 0), (0, 32768)]:     # one past the bound (and the ends of `short`): refused, planes untouched
        planes = np.stack([_worst_case_plane(min(U, 32767), L, 0, 0, 0), np.zeros(512 * 512, np.int16)]).astype(np.int16)
        if L == 32768:
            planes[0] = np.where(planes[0] < 0, -32768, 0)
        j, p = _cuda(planes), _cuda(np.full_like(planes, 77))
        assert enc.lib.nhw_stage_analysis(enc.h, j.data_ptr(), p.data_ptr(), 2, 512 * 512, 512, 512, 0, None) == NHW_E_ARG, (U, L)
        torch.cuda.synchronize()
        assert np.array_equal(j.cpu().numpy(), planes) and (p.cpu().numpy() == 77).all()
    # the check is on the planes handed in, not a property of the handle: an in-domain call right after a refusal works
    ok = np.stack([_worst_case_plane(255, 0, 2, 0, 0)])
    j, p = _cuda(ok), _cuda(np.zeros_like(ok))
    assert enc.lib.nhw_stage_analysis(enc.h, j.data_ptr(), p.data_ptr(), 1, 512 * 512, 512, 512, 0, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(p.cpu().numpy()[0], oracle.analysis(ok[0], 512, 512, 0)[1])


@pytest.mark.gpu
@pytest.mark.parametrize("stride,size,final", [(512, 512, 0), (512, 256, 1), (256, 256, 0), (256, 128, 1)])
def test_filterbank_matches_oracle(enc, oracle, stride, size, final):
    import torch
    rng = np.random.default_rng(size + final)
    planes = rng.integers(-300, 300, (3, stride * stride)).astype(np.int16)
    planes[1] = rng.integers(0, 256, stride * stride)           # pixel-like
    planes[2] = rng.integers(-2000, 2600, stride * stride)      # pass-1-like range
    if size == 512:
        # the level-1 kernel (k_front_plain) works two columns to a dword in 16-bit arithmetic: exact on its stated domain (include/nhw_hip.h,
        # NHW_ANA512_BOUND: 104 U + 40 L <= 32720; proof in nhw_front_image.h) -- the luma it is built for is 0 .. 255 and what the pre-filters
        # add (at most +-4 a pixel); planes outside the domain are refused (test_level1_analysis_domain_is_checked)
        planes[0] = rng.integers(-4, 314, stride * stride)
        planes[2] = np.where(rng.random(stride * stride) < 0.5, 0, 255) + rng.integers(-4, 5, stride * stride)   # hard edges with pre-filter overshoot
    j, p = _cuda(planes), _cuda(np.zeros_like(planes))
    assert enc.lib.nhw_stage_analysis(enc.h, j.data_ptr(), p.data_ptr(), 3, stride * stride, stride, size, final, None) == 0
    torch.cuda.synchronize()
    gj, gp = j.cpu().numpy(), p.cpu().numpy()
    for i in range(3):
        oj, op = oracle.analysis(planes[i], stride, size, final)
        m = np.zeros((stride, stride), bool); m[:size, :size] = True
        assert np.array_equal(gp[i].reshape(stride, stride)[m], op.reshape(stride, stride)[m]), "coefficients"
        if size == 512:     # the band kernel (the product's level-1 kernel) leaves only what is read afterwards: the LL quadrant
            m[:] = False; m[:256, :256] = True
        assert np.array_equal(gj[i].reshape(stride, stride)[m], oj.reshape(stride, stride)[m]), "transposed plane / LL copy-back"
    if size == 512:
        return              # the encoder has no synthesis of that size
    # synthesis of small coefficients (decoder-simulation path of the encoder)
    coef = rng.integers(-40, 300, (2, stride * stride)).astype(np.int16)
    j, p = _cuda(coef), _cuda(np.zeros_like(coef))
    assert enc.lib.nhw_stage_synthesis(enc.h, j.data_ptr(), p.data_ptr(), 2, stride * stride, stride, size, None) == 0
    torch.cuda.synchronize()
    gj, gp = j.cpu().numpy(), p.cpu().numpy()
    for i in range(2):
        oj, op = oracle.synthesis(coef[i], stride, size)
        m = np.zeros((stride, stride), bool); m[:size, :size] = True
        assert np.array_equal(gp[i].reshape(stride, stride)[m], op.reshape(stride, stride)[m])
        assert np.array_equal(gj[i].reshape(stride, stride)[m], oj.reshape(stride, stride)[m])


@pytest.mark.gpu
@pytest.mark.parametrize("q", [17, 18, 19, 20, 21, 22, 23])
def test_fused_front_matches_oracle(oracle, q):
    """The fused front kernel (colour + 4:2:0 + pre-filter + level-1 analysis from the BGR bytes): coefficient plane, LL copy-back, ll1, the two
    4:2:0 chroma planes and (q>=22) the kept transposed horizontal-pass plane, against the oracle's stage functions, on smooth, noisy
    and blocky inputs."""
    import ctypes
    import torch
    import nhwcodec_amd
    imgs = [oracle.synth(9), class_image("noise", 3), class_image("blocks", 4), class_image("gradient")]
    e = nhwcodec_amd.Encoder(0, max_batch=len(imgs))
    e.lib.nhw_debug_stop_after(e.h, 4 if q < 22 else 3)      # colour, (pre-filter), analysis, ll1 copy
    e.encode_device(_cuda(np.stack(imgs)), q)
    torch.cuda.synchronize()

    def rd(buf, i, nbytes):
        out = np.empty(nbytes, np.uint8)
        assert e.lib.nhw_debug_read(e.h, buf, i, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(nbytes)) == 0
        return out.view(np.int16)
    for i, im in enumerate(imgs):
        y, u, v = oracle.color(im, q)
        assert np.array_equal(rd(2, i, 65536).view(np.uint8), u.ravel()) and np.array_equal(rd(3, i, 65536).view(np.uint8), v.ravel()), f"image {i}: 4:2:0 planes"
        if q < 22:
            y = oracle.prefilter(y, q)
        oj, op, ok = oracle.analysis(y, 512, 512, 0, keep=True)
        assert np.array_equal(rd(1, i, 8 * 65536), op), f"image {i}: level-1 coefficients"
        assert np.array_equal(rd(0, i, 8 * 65536).reshape(512, 512)[:256, :256], oj.reshape(512, 512)[:256, :256]), f"image {i}: LL copy-back"
        assert np.array_equal(rd(6, i, 2 * 65536), oj.reshape(512, 512)[:256, :256].ravel()), f"image {i}: ll1"
        if q >= 22:
            assert np.array_equal(rd(10, i, 4 * 65536), ok), f"image {i}: kept horizontal-pass plane"
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", list(range(1, 24)))
def test_whole_encoder_bit_exact_vs_oracle(enc, oracle, q):
    """The .nhw bytes of a mixed batch (synthetic seeds + every robustness class) equal the oracle's."""
    imgs = [oracle.synth(s) for s in (0, 7, 31)] + [class_image(k, 0) for k in ("noise", "blocks", "flat", "gradient", "black", "white")]
    got = enc.encode(np.stack(imgs), q)
    for i, im in enumerate(imgs):
        assert got[i] == oracle.encode(im, q), f"q{q} image {i}: .nhw differs from the oracle"


@pytest.mark.gpu
def test_golden_files(enc, oracle, manifest):
    keys = sorted(manifest["files"])
    imgs, qs = [], []
    for key in keys:
        kind, s, q = key.rsplit("_", 2)
        imgs.append(oracle.synth(int(s[1:])) if kind == "synth" else class_image(kind, int(s[1:])))
        qs.append(int(q[1:]))
    for q in sorted(set(qs)):
        sel = [i for i in range(len(keys)) if qs[i] == q]
        got = enc.encode(np.stack([imgs[i] for i in sel]), q)
        for g, i in zip(got, sel):
            want = open(os.path.join(GOLD, "nhw", keys[i] + ".nhw"), "rb").read()
            assert g == want, f"{keys[i]}: differs from the reference's own .nhw"


@pytest.mark.gpu
def test_batch_position_and_workspace_reuse_do_not_matter(enc, oracle):
    """Same image at every batch slot, after a batch of noise has dirtied the workspace."""
    a, b = oracle.synth(2), class_image("noise", 5)
    enc.encode(np.stack([b] * 64), 23)
    got = enc.encode(np.stack([a] * 64), 20)
    want = oracle.encode(a, 20)
    assert all(g == want for g in got)
    got = enc.encode(np.stack([b, a] * 8), 20)
    assert got[1] == want and got[15] == want and got[0] == oracle.encode(b, 20)


@pytest.mark.gpu
def test_device_generator_matches_definition(enc, oracle):
    import torch
    t = enc.synth_device(5, seed_base=40)
    torch.cuda.synchronize()
    t = t.cpu().numpy()
    for i in range(5):
        assert np.array_equal(t[i], oracle.synth(40 + i))


@pytest.mark.gpu
def test_unsupported_quality_fails_loudly(enc, oracle):
    import nhwcodec_amd
    for q in (0, 24, -1):     # -q0 is accepted by the reference CLI but has no tables downstream
        with pytest.raises(nhwcodec_amd.NhwError):
            enc.encode(np.stack([oracle.synth(0)]), q)


@pytest.mark.gpu
def test_full_batch_4096_properties(oracle):
    """BASELINE.json configs[1] size: 4096 synthetic images, -q20, one GPU.  Properties that do not need 4096
    oracle runs: every status 0, determinism (checksum of checksums over two runs), sizes in the natural-image
    band, and a sample of 24 images bit-exact against the oracle; a sample decodes with the reference decoder."""
    import torch
    import nhwcodec_amd
    n = 4096
    e = nhwcodec_amd.Encoder(0, max_batch=n)
    bgr = e.synth_device(n, seed_base=1000)
    out = e.alloc_out(n)
    e.encode_device(bgr, 20, out)
    torch.cuda.synchronize()
    o, sizes, status = out
    assert int((status != 0).sum()) == 0
    sz = sizes.cpu().numpy()
    assert sz.min() > 15000 and sz.max() < 60000

    def digest():
        idx = torch.arange(nhwcodec_amd.OUT_STRIDE, device="cuda")[None, :]
        masked = torch.where(idx < sizes[:, None].to(torch.int64), o, torch.zeros_like(o)).to(torch.int64)
        w = (idx % 251 + 1).to(torch.int64)
        per = (masked * w).sum(1) + sizes.to(torch.int64) * 1000003
        return int((per * torch.arange(1, n + 1, device="cuda")).sum().item())
    d1 = digest()
    e.encode_device(bgr, 20, out)
    torch.cuda.synchronize()
    assert digest() == d1, "two runs over the same resident batch differ"
    pick = list(range(0, n, 171))[:24]
    host = o[pick].cpu().numpy()
    for k, i in enumerate(pick):
        want = oracle.encode(oracle.synth(1000 + i), 20)
        assert host[k, : sz[i]].tobytes() == want, f"image {i}"
    dec = os.path.join(ROOT, "oracle", "_ref", "nhw-dec")
    if os.path.exists(dec):
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "x.nhw")
            open(p, "wb").write(host[0, : sz[pick[0]]].tobytes())
            subprocess.check_call([dec, p, os.path.join(td, "x.bmp")], stdout=subprocess.DEVNULL)
            px = np.frombuffer(open(os.path.join(td, "x.bmp"), "rb").read()[54:], np.uint8).reshape(512, 512, 3).astype(int)
            mse = ((px - oracle.synth(1000 + pick[0]).astype(int)) ** 2).mean()
            assert 10 * np.log10(255 ** 2 / mse) > 30
    e.close()


@pytest.mark.gpu
def test_repeated_batches_are_identical():
    """Race detector: 25 encodes of one resident 4096-image batch must give 25 identical outputs.  (The fused front
    kernel once read luma rows from the plane other workgroups of the same image write their LL rows into: about one
    run in thirty differed in a handful of images, depending on how far apart the XCDs' dispatchers had drifted.)"""
    import torch
    import nhwcodec_amd
    n = 4096
    e = nhwcodec_amd.Encoder(0, max_batch=n)
    bgr = e.synth_device(n, seed_base=777)
    out = e.alloc_out(n)
    idx = torch.arange(nhwcodec_amd.OUT_STRIDE, device="cuda")[None, :]
    w = (idx % 251 + 1).to(torch.int64)

    def digests():
        e.encode_device(bgr, 20, out)
        torch.cuda.synchronize()
        o, sizes, status = out
        assert int((status != 0).sum()) == 0
        per = torch.empty(n, dtype=torch.int64, device="cuda")
        for a in range(0, n, 512):
            m = torch.where(idx < sizes[a:a + 512, None].to(torch.int64), o[a:a + 512], torch.zeros_like(o[a:a + 512])).to(torch.int64)
            per[a:a + 512] = (m * w).sum(1) + sizes[a:a + 512].to(torch.int64) * 1000003
        return per.clone()
    ref = digests()
    for k in range(24):
        bad = torch.nonzero(digests() != ref).flatten().tolist()
        assert not bad, f"run {k + 1}: images {bad[:8]} differ from the first run"
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,q", [(2429, 20), (2928, 20), (3254, 20), (9205, 23), (9787, 23), (5862, 18)])
def test_chroma_rows_that_end_in_a_pair_mark(enc, oracle, seed, q):
    """Images (found with tools/dev/gpu_hazard_check.py) whose chroma mark walk takes a pair mark in the last column of a
    row: from there on the reference compares against cll1 one cell further on (nhw_encoder.c:2372-2427)."""
    im = oracle.synth(seed)
    assert enc.encode(im[None], q)[0] == oracle.encode(im, q)


@pytest.mark.gpu
@pytest.mark.parametrize("n,q", [(1001, 23), (517, 18)])
def test_sub_batches_on_streams_match_oracle(oracle, n, q):
    """With NHW_PARTS=2, batches of 512 images or more run their stages behind the front as two sub-batches on streams of
    their own (odd split here): images around the seam and at both ends must equal the oracle's."""
    import torch
    import nhwcodec_amd
    os.environ["NHW_PARTS"] = "2"
    try:
        e = nhwcodec_amd.Encoder(0, max_batch=n)
    finally:
        del os.environ["NHW_PARTS"]
    bgr = e.synth_device(n, seed_base=4000)
    o, sizes, status = e.encode_device(bgr, q)
    torch.cuda.synchronize()
    assert e.timing().parts == 2
    assert int((status != 0).sum()) == 0
    sz = sizes.cpu().numpy()
    for i in (0, 1, n // 2 - 1, n // 2, n // 2 + 1, n - 2, n - 1):
        assert o[i, : sz[i]].cpu().numpy().tobytes() == oracle.encode(oracle.synth(4000 + i), q), f"image {i}"
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", list(range(1, 24)))
def test_many_seeds_bit_exact(oracle, q):
    """40 more synthetic images per quality (seeds far from the ones used elsewhere), bit-exact against the oracle."""
    import torch
    import nhwcodec_amd
    n, base = 40, 70000 + 1000 * q
    e = nhwcodec_amd.Encoder(0, max_batch=n)
    o, sizes, status = e.encode_device(e.synth_device(n, seed_base=base), q)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0
    sz = sizes.cpu().numpy()
    host = o.cpu().numpy()
    bad = [i for i in range(n) if host[i, : sz[i]].tobytes() != oracle.encode(oracle.synth(base + i), q)]
    assert not bad, f"q{q}: images {bad} differ from the oracle"
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 4, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 20, 22, 23])
def test_robustness_classes_more_seeds(enc, oracle, q):
    """White noise and hard-edge rectangles (the worst cases for the order-dependent passes and the packetiser), six seeds each."""
    imgs = [class_image(k, s) for k in ("noise", "blocks") for s in range(1, 7)] + [class_image("tiles", s) for s in range(3)]   # tiles: the LL2 coder's mode 1
    got = enc.encode(np.stack(imgs), q)
    bad = [i for i, im in enumerate(imgs) if got[i] != oracle.encode(im, q)]
    assert not bad, f"q{q}: images {bad} differ from the oracle"


def _residual_range_image(kind, seed):
    """Pictures whose level-1 LL residuals (reconstruction - LL1) sit in the +-2 .. +-9 range where the rules of Y22 / Y23 fire, everywhere and
    in particular in the first and the last column and in the last rows: a smooth field + low-amplitude noise / stripes / a textured frame."""
    rng = np.random.RandomState(4000 + seed)
    yy, xx = np.mgrid[0:512, 0:512]
    base = 96 + 48 * np.sin(xx / (17.0 + seed)) * np.cos(yy / (23.0 + 2 * seed)) + 0.08 * xx
    amp = (3, 6, 12, 24, 40)[seed % 5]
    if kind == "grain":
        img = base[..., None] + rng.uniform(-amp, amp, (512, 512, 3))
    elif kind == "stripes":                                       # columns of alternating offsets: the column walks see a chain in every column
        img = base[..., None] + amp * (((xx // (1 + seed % 3)) & 1) * 2 - 1)[..., None] + rng.uniform(-2, 2, (512, 512, 3))
    else:                                                         # frame: the texture only in a border of 24 pixels (columns 0, 255 of the LL band; rows 254, 255)
        img = np.repeat(base[..., None], 3, 2)
        m = (xx < 24) | (xx >= 488) | (yy < 24) | (yy >= 488)
        img[m] += rng.uniform(-amp, amp, (int(m.sum()), 3))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [13, 15, 16, 17, 18, 19, 20, 21, 22, 23])
def test_residual_rules_under_load(enc, oracle, q):
    """Y22 and Y23 are one column sweep since round 5 (residuals_fused_par): a prologue settles what the reference's column order couples
    (step 0 of every column, column 0's and column 255's whole walks, sparsely), Y23 runs from a class table, and Y21 skips the rows in which no
    cell fires.  The generator's pictures fire these rules in a few per cent of the cells; these fire them everywhere -- grain of five amplitudes,
    column stripes, a textured frame that loads exactly the coupled columns and the last rows (the write to row 256) -- at every quality that
    runs the passes, q18 with its rule that also rewrites the cell below."""
    imgs = [_residual_range_image(k, s) for k in ("grain", "stripes", "frame") for s in range(5)]
    got = enc.encode(np.stack(imgs), q)
    bad = [i for i, im in enumerate(imgs) if got[i] != oracle.encode(im, q)]
    assert not bad, f"q{q}: images {bad} differ from the oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 8, 10, 20, 23])
def test_full_batch_4096_every_image_bit_exact(q):
    """BASELINE configs 2 and 3 (-q20; -q1 / -q10 / -q23) at their full size, every one of the 4096 outputs against the oracle (oracle
    side spread over the host cores); -q8 for the band of qualities whose bursts end through t17 (the cut bursts of the chain)"""
    from gpu_enc_fullcheck import full_encode_check
    bad = full_encode_check(4096, q, 900000 + q)
    assert not bad, f"q{q}: images {bad[:16]} differ from the oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 6, 10, 11, 13, 14, 16, 17, 18, 19, 20, 21, 22, 23])
def test_compat_mode_matches_oracle_and_stock_binary(oracle, q):
    """NHW_COMPAT_GLIBC_ONESHOT: bit-exact against the oracle in its GLIBC_ONESHOT mode, equal to the stock reference binary outside the
    positions that binary leaves un-initialised (on images where the luma heap neighbours are the only live out-of-bounds reads: DESIGN.md
    section 2); switching back gives the canonical output again (the guard behind ll1 is clean)."""
    import nhwcodec_amd
    from oracle.harness import STOCK_ENC, stock_encode, uninitialised_positions
    enc = nhwcodec_amd.Encoder(0, 16)
    imgs = np.stack([oracle.synth(i) for i in (110, 117, 924, 925, 931, 0, 1, 2, 3, 4)] + [class_image(k, q) for k in ("blocks", "noise", "tiles")]   # the first five: images where the chroma neighbour matters
                    + ([class_image("noise", 3), class_image("noise", 5)] if q <= 16 else []))   # below q14: loud enough for Y20 to read behind resIII
    enc.set_compat(True)
    got = enc.encode(imgs, q)
    got2 = enc.encode(imgs, q)                      # a second batch over the same workspace
    enc.set_compat(False)
    canon = enc.encode(imgs, q)
    enc.close()
    oracle.set_oob_mode(True)
    try:
        want = [oracle.encode(im, q) for im in imgs]
    finally:
        oracle.set_oob_mode(False)
    assert [i for i in range(len(imgs)) if got[i] != want[i]] == []
    assert got2 == got
    assert [i for i in range(len(imgs)) if canon[i] != oracle.encode(imgs[i], q)] == []
    if os.path.exists(STOCK_ENC):
        for i in list(range(6)) + list(range(13, len(imgs))):
            stock = stock_encode(imgs[i], q)
            assert len(stock) == len(got[i])
            pad = uninitialised_positions(stock)
            assert [k for k in range(len(stock)) if stock[k] != got[i][k] and k not in pad] == [], f"image {i}"


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 8, 10, 16, 17, 20, 21])
def test_front_fallback_paths_are_exact(oracle, q):
    """The pre-filter's carry normally comes from a short look-back (its 16 states merge within a dozen pixels); where they have not merged, a
    segment is replayed from the one before it, in raster order across the rows of a band (k_front_image).  That path is rare on real
    content, so a debug switch sends every segment down it: the output must not change.  Quality 1..16: the same switch makes every band of
    pass A (k_low_pre) go three rows back for the state its first row starts from and come forward again with the exact state."""
    import nhwcodec_amd
    enc = nhwcodec_amd.Encoder(0, 32)
    imgs = np.stack([oracle.synth(40 + i) for i in range(20)] + [class_image(k, s) for k in ("noise", "blocks", "tiles", "gradient") for s in (1, 2)])
    enc.lib.nhw_debug_front_fallback.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert enc.lib.nhw_debug_front_fallback(enc.h, 1) == 0
    forced = enc.encode(imgs, q)
    assert enc.lib.nhw_debug_front_fallback(enc.h, 0) == 0
    normal = enc.encode(imgs, q)
    enc.close()
    assert forced == normal
    bad = [i for i in range(len(imgs)) if forced[i] != oracle.encode(imgs[i], q)]
    assert not bad, f"q{q}: images {bad} differ from the oracle on the fallback paths"


@pytest.mark.gpu
def test_two_devices_in_one_process(oracle):
    """One process, an encoder (and a decoder) handle on device 0 AND on device 1, q10 and q20 on both: the per-device dynamic-LDS opt-ins
    (nhw_front_set_attrs / nhw_tail_set_attrs at nhw_enc_create), the per-handle streams and the workspace of a device other than 0.  Skipped on
    a one-GPU box; it runs the first time a multi-GPU box is leased (`nhw-enc --devices 0,1` is the C host's form of the same)."""
    import torch
    import nhwcodec_amd
    lib = nhwcodec_amd.load_library()
    if lib.nhw_device_count() < 2:
        pytest.skip("one GPU visible: the second-device path needs a multi-GPU box")
    imgs = np.stack([oracle.synth(s) for s in (0, 1, 2)])
    encs = [nhwcodec_amd.Encoder(d, max_batch=4) for d in (0, 1)]
    decs = [nhwcodec_amd.Decoder(d, max_batch=4) for d in (0, 1)]
    for q in (10, 20):
        want = [oracle.encode(im, q) for im in imgs]
        outs = []
        for d, e in enumerate(encs):                     # both devices in flight before either is read back
            with torch.cuda.device(d):
                outs.append(e.encode_device(torch.from_numpy(imgs).to(f"cuda:{d}"), q))
        for d, (o, sizes, status) in enumerate(outs):
            with torch.cuda.device(d):
                torch.cuda.synchronize(d)
                assert (status.cpu().numpy() == 0).all()
                files = [o[i, : int(sizes[i])].cpu().numpy().tobytes() for i in range(len(imgs))]
                assert files == want, f"device {d}, q{q}"
                px, qs = decs[d].decode(files)
                for i, f in enumerate(files):
                    wpx, wq = oracle.decode(f)
                    assert qs[i] == wq == q and np.array_equal(px[i], wpx), f"decode on device {d}, q{q}, image {i}"
    for h in encs + decs:
        h.close()


@pytest.mark.gpu
def test_host_path_pcie_inclusive(oracle):
    """nhw_enc_batch from page-locked host memory (nhw_host_alloc): 2048 images uploaded in chunks next to the encode of the chunk before,
    files compacted on the device and downloaded.  Bit-exact on a sample, and the PCIe-inclusive rate stays above what a pageable copy
    alone used to allow (the figure itself goes to bench.py's `host_path` field, never to `value`)."""
    import time
    import nhwcodec_amd
    n = 2048
    e = nhwcodec_amd.Encoder(0, max_batch=1024)
    a = e.pinned_images(n)
    base = e.synth_device(1024, seed_base=300).cpu().numpy()
    a[:1024] = base; a[1024:] = base[::-1]
    e2 = nhwcodec_amd.Encoder(0, max_batch=n)
    t0 = time.perf_counter()
    got = e2.encode(a, 20)                     # the FIRST call of the handle: its staging buffers exist since nhw_enc_create
    dt_first = time.perf_counter() - t0
    dt = 1e9
    for _ in range(5):                         # a PCIe-bound single shot spreads from 9.8 to 12 Gpixel/s on one box: the best of five
        t0 = time.perf_counter()
        got = e2.encode(a, 20)
        dt = min(dt, time.perf_counter() - t0)
    for i in (0, 1, 1023, 1024, 2047):
        assert got[i] == oracle.encode(a[i], 20)
    rate = n * 0.262144 / dt / 1e3
    print(f"host path: {n} images in {dt * 1e3:.1f} ms = {rate:.1f} Gpixel/s incl. PCIe both ways (first call of the handle: {dt_first * 1e3:.1f} ms)")
    assert rate > 9.0                          # the driver's box gives 11.8 (35 GB/s of PCIe); a pageable copy alone used to allow 5
    assert n * 0.262144 / dt_first / 1e3 > 6.0, "the first call of a handle no longer pays for gigabytes of hipMalloc (it was 5.2 against 11.6)"
    e.free_pinned(); e.close(); e2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", [17, 20, 23])
def test_tail_stages_match_the_oracle_trace(oracle, q):
    """Rows a8-a16 stage by stage: the batch driver is stopped after every launch group of the luma tail and of both chroma sequences
    (nhw_debug_stop_after) and the live planes are compared with the oracle's checkpoint of the same point of encode_image -- both filter
    banks of both closed loops, both dequantiser simulations, the quantisers.  A regression in one pass is reported at that pass."""
    import torch
    import nhwcodec_amd
    B = dict(JPEG=0, PROC=1, PU=2, PV=3, CJPEG=4, CPROC=5)
    seeds = (0, 1)
    e = nhwcodec_amd.Encoder(0, max_batch=len(seeds))
    imgs = np.stack([oracle.synth(s) for s in seeds])
    d_in = torch.from_numpy(imgs).cuda()
    traces = [oracle.encode(imgs[i], q, trace=True) for i in range(len(seeds))]
    shift = 0 if q < 22 else -1                                 # no pre-filter stage from q22
    plan = []                                                   # (stage, index among same-named trace records, record name, [(buffer, blob of the record, bytes)])
    for st, k, nm in [(3, 0, "wavelet_analysis_512"), (5, 0, "wavelet_analysis_256"), (6, 0, "offsetY_recons256_p1"), (7, 0, "wavelet_synthesis_256"),
                      (9, 1, "wavelet_analysis_256"), (11, 0, "offsetY_recons256_p0"), (12, 1, "wavelet_synthesis_256")]:
        plan.append((st + shift, k, nm, [("JPEG", 0, 8 * 65536), ("PROC", 1, 8 * 65536)]))
    if q >= 22:                                                 # below that the quantiser writes the symbol stream directly and leaves the plane alone
        plan.append((13 + shift, 0, "offsetY", [("PROC", 0, 8 * 65536)]))
    for comp in (0, 1):
        base = 13 + shift + 12 * comp
        for st, k, nm in [(2, 2 + comp, "wavelet_analysis_256"), (4, 2 * comp, "wavelet_analysis_128"), (5, comp, "offsetUV_recons256_c1"),
                          (6, 2 * comp, "wavelet_synthesis_128"), (8, 2 * comp + 1, "wavelet_analysis_128"), (10, comp, "offsetUV_recons256_c0"),
                          (11, 2 * comp + 1, "wavelet_synthesis_128")]:
            plan.append((base + st, k, nm, [("CJPEG", 0, 2 * 65536), ("CPROC", 1, 2 * 65536)]))
        plan.append((base + 12, comp, "offsetUV", [("CPROC", 0, 2 * 65536)]))
    try:
        for st, k, nm, bufs in plan:
            e.lib.nhw_debug_stop_after(e.h, st)
            e.encode_device(d_in, q)
            torch.cuda.synchronize()
            for i in range(len(seeds)):
                recs = [b for n, b in traces[i][1] if n == nm]
                for bname, bi, nbytes in bufs:
                    out = np.empty(nbytes, np.uint8)
                    assert e.lib.nhw_debug_read(e.h, B[bname], i, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(nbytes)) == 0
                    got, want = out.view(np.int16), np.frombuffer(recs[k][bi], np.int16)
                    if bname == "JPEG":                         # only the 256x256 corner of the luma jpeg plane is live after level 1
                        got, want = got.reshape(512, 512)[:256, :256], want.reshape(512, 512)[:256, :256]
                    bad = np.argwhere(got != want)
                    assert bad.size == 0, f"q{q} stage {st} ({nm} #{k}) image {i} plane {bname}: {len(bad)} cells differ, first {bad[0].tolist()}"
    finally:
        e.lib.nhw_debug_stop_after(e.h, 0)
    got = e.encode(imgs, q)
    assert [g == t[0] for g, t in zip(got, traces)] == [True] * len(seeds)
    e.close()


def _read_ws(e, buf, img, nbytes):
    out = np.empty(nbytes, np.uint8)
    assert e.lib.nhw_debug_read(e.h, buf, img, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(nbytes)) == 0
    return out


def _ws_index(name):
    """index of a workspace buffer (nhwcodec_amd/csrc/nhw_ws.h, the B_* list) for nhw_debug_read"""
    import re
    txt = open(os.path.join(ROOT, "nhwcodec_amd", "csrc", "nhw_ws.h")).read()
    body = re.sub(r"/\*.*?\*/", "", txt[txt.index("enum {"):txt.index("B_COUNT")], flags=re.S)
    names = re.findall(r"B_[A-Z0-9_]+", body)
    return names.index("B_" + name)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 10, 16, 17, 20, 23])
def test_symbol_list_equals_the_byte_stream(oracle, q):
    """Since round 5 the symbol stream travels as a LIST -- a non-zero map per 64-symbol slice and the symbols that are not 128 (the luma
    quantiser, Y31 on map + values, the chroma quantiser, the packetiser) -- and the byte stream of the reference (nhw_encoder.c:2108-2252,
    2553-2570) is only written for the stage checks.  Here both forms are made (the batch driver stopped behind Y31, and behind the second
    chroma sequence) and compared symbol for symbol: the luma part after all three rewrites of Y31 (chains of (+-8, 0, 0, 0, +-8) across slice
    edges, the long zero runs of the low qualities, the cleared first and last four symbols), the chroma part as the packetiser will order it.
    Images: the generator's, noise (every symbol non-zero: the worst case of the list), blocks, flat, gradient, tiles."""
    import torch
    import nhwcodec_amd
    Q = 65536
    imgs = np.stack([oracle.synth(7000 + 31 * q + i) for i in range(6)] + [class_image(k, q) for k in ("noise", "blocks", "flat", "gradient", "tiles")])
    e = nhwcodec_amd.Encoder(0, max_batch=len(imgs))
    d_in = torch.from_numpy(imgs).cuda()
    rd = lambda name, i, nbytes, dt: np.frombuffer(bytes(_read_ws(e, _ws_index(name), i, nbytes)), dt)
    luma_stage = 12 if q >= 22 else 13 if q > 12 else 11 if q > 6 else 7       # the stage count behind Y31 (fewer stages where a closed loop is missing)
    try:
        e.lib.nhw_debug_stop_after(e.h, luma_stage)
        e.encode_device(d_in, q)
        torch.cuda.synchronize()
        for i in range(len(imgs)):
            dense = rd("SCAN", i, 4 * Q, np.uint8)
            nzs, voff, vals = rd("NZS", i, 4 * Q // 8, np.uint64), rd("VOFF", i, 4 * Q // 16, np.uint32), rd("VALS", i, 4 * Q, np.uint8)
            bits = np.unpackbits(nzs.view(np.uint8), bitorder="little").astype(bool)          # bit k of slice g = symbol 64 g + k
            got = np.full(4 * Q, 128, np.uint8)
            cnt = bits.reshape(-1, 64).sum(1)
            off = (voff & 0x1FFFFFFF).astype(np.int64)
            idx = np.concatenate([np.arange(o, o + c) for o, c in zip(off, cnt)]) if cnt.sum() else np.zeros(0, np.int64)
            got[bits] = vals[idx]
            bad = np.flatnonzero(got != dense)
            assert bad.size == 0, f"q{q} image {i}: luma list and byte stream differ at {bad[:8].tolist()} ({bad.size} symbols)"
            assert not (vals[idx] == 128).any(), "a listed symbol is the zero symbol"
        if q >= 17:
            e.lib.nhw_debug_stop_after(e.h, luma_stage + 24)                                   # behind V's quantiser (12 stages a chroma sequence)
            e.encode_device(d_in, q)
            torch.cuda.synchronize()
            for i in range(len(imgs)):
                dense = rd("SCAN", i, 6 * Q, np.uint8)[4 * Q:]
                cnzq, cvals = rd("CNZQ", i, Q // 4 + 256, np.uint8), rd("CVALS", i, 2 * Q, np.uint8)
                maps, fbase = cnzq[:Q // 4].view(np.uint64).reshape(16, 64, 2), cnzq[Q // 4:Q // 4 + 64].view(np.uint32)
                got = np.full(2 * Q, 128, np.uint8)
                for F in range(16):
                    at = int(fbase[F])
                    for lane in range(64):
                        for half in range(2):
                            S = 64 * (lane >> 1) + 16 * (F >> 2) + 4 * (F & 3) + 2 * (lane & 1) + half    # the stream slice of (flush, lane, half): pack_chroma_order
                            b = np.unpackbits(maps[F, lane, half:half + 1].view(np.uint8), bitorder="little").astype(bool)
                            n = int(b.sum())
                            got[64 * S:64 * S + 64][b] = cvals[at:at + n]
                            at += n
                bad = np.flatnonzero(got != dense)
                assert bad.size == 0, f"q{q} image {i}: chroma list and byte stream differ at {bad[:8].tolist()} ({bad.size} symbols)"
    finally:
        e.lib.nhw_debug_stop_after(e.h, 0)
    files = e.encode(imgs, q)
    assert all(f == oracle.encode(imgs[i], q) for i, f in enumerate(files) if i < 3)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 10, 13, 16])
def test_low_quality_stages_match_the_oracle_trace(oracle, q):
    """The same walk for the quality 1..16 forms: colour, the rationed pre-filter, both filter banks of both closed loops (where the quality
    has them), both dequantiser simulations (offsetY_recons256 with the rationed low bits), Y11/Y12 (through the analysis that follows
    them), the quantiser's plane, and both chroma sequences with pre_processing_UV -- stage by stage against the oracle's checkpoints
    (the stage plan is the developer tool's, tools/dev/gpu_low_debug.py)."""
    import torch
    import nhwcodec_amd
    from gpu_low_debug import B, plan_for, read
    seeds = (0, 1)
    e = nhwcodec_amd.Encoder(0, max_batch=len(seeds))
    imgs = np.stack([oracle.synth(s) for s in seeds])
    d_in = torch.from_numpy(imgs).cuda()
    traces = [oracle.encode(imgs[i], q, trace=True) for i in range(len(seeds))]
    try:
        for st, nm, k, bufs in plan_for(q):
            e.lib.nhw_debug_stop_after(e.h, st)
            e.encode_device(d_in, q)
            torch.cuda.synchronize()
            for i in range(len(seeds)):
                recs = [b for n, b in traces[i][1] if n == nm]
                for bname, bi, nbytes, corner in bufs:
                    g, w = read(e, B[bname], i, nbytes), recs[k][bi]
                    dt = np.uint8 if bname in ("PU", "PV") else np.int16
                    ga, wa = np.frombuffer(g, dt), np.frombuffer(w, dt)
                    if corner:
                        ga, wa = ga.reshape(512, 512)[:corner, :corner], wa.reshape(512, 512)[:corner, :corner]
                    bad = np.argwhere(ga != wa)
                    assert bad.size == 0, f"q{q} stage {st} ({nm} #{k}) image {i} plane {bname}: {len(bad)} cells differ, first {bad[0].tolist()}"
    finally:
        e.lib.nhw_debug_stop_after(e.h, 0)
    got = e.encode(imgs, q)
    assert [g == t[0] for g, t in zip(got, traces)] == [True] * len(seeds)
    e.close()


@pytest.mark.gpu
def test_synthetic_entry_point_and_device_count(oracle):
    """nhw_enc_synth_batch (what `nhw-enc --synthetic` calls): generator seeds in, the oracle's files for those seeds out; nhw_device_count
    sees the GPU the suite runs on."""
    import nhwcodec_amd
    e = nhwcodec_amd.Encoder(0, max_batch=8)
    assert e.lib.nhw_device_count() >= 1
    got = e.encode_synthetic(5, 4242, 19)
    assert got == [oracle.encode(oracle.synth(4242 + i), 19) for i in range(5)]
    with pytest.raises(nhwcodec_amd.NhwError):
        e.encode_synthetic(2, 0, 24)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 10, 13, 20, 23])
def test_odd_batch_sizes(oracle, q):
    """Batches that do not fill the kernels' image groups (four images per workgroup in the wave kernels, two per wavefront in the q <= 16
    pre-filter, bands of an image spread over the XCDs): 1, 2, 3, 5 and 7 images in a workspace made for 7."""
    import nhwcodec_amd
    e = nhwcodec_amd.Encoder(0, max_batch=7)
    imgs = np.stack([oracle.synth(3100 + i) for i in range(7)])
    want = [oracle.encode(im, q) for im in imgs]
    for n in (1, 2, 3, 5, 7):
        assert e.encode(imgs[:n], q) == want[:n], f"q{q} n={n}"
    assert e.encode(imgs[4:7], q) == want[4:7]
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"NHW_CHROMA_FORK": "0"}, {"NHW_LISTS_FORK": "0"}, {"NHW_LL_FORK": "0"}, {"NHW_QUANT_JOIN": "0"}])
def test_encoder_stream_modes_give_the_same_files(oracle, env):
    """The chroma sequence, the position lists and the LL2 coder run on streams of their own beside the luma tail; NHW_CHROMA_FORK=0 /
    NHW_LISTS_FORK=0 / NHW_LL_FORK=0 put them back in line (the last one also moves the putting back of the verbatim LL2 samples from the
    synthesis kernel to the dequantiser simulation again).  Same arithmetic under another schedule: 512 images at q20 and q23, every file equal to the
    default schedule's, a sample of them to the oracle's."""
    import torch
    import nhwcodec_amd
    n = 512
    base = nhwcodec_amd.Encoder(0, n)
    os.environ.update(env)
    try:
        e = nhwcodec_amd.Encoder(0, n)
    finally:
        for k in env: del os.environ[k]
    bgr = base.synth_device(n, seed_base=61000)
    for q in (20, 23):
        o0, s0, st0 = base.encode_device(bgr, q)
        o1, s1, st1 = e.encode_device(bgr, q)
        torch.cuda.synchronize()
        assert int(st0.abs().sum()) == 0 and int(st1.abs().sum()) == 0 and torch.equal(s0, s1)
        sz = s0.cpu().numpy()
        a0, a1 = o0.cpu().numpy(), o1.cpu().numpy()
        assert [i for i in range(n) if not np.array_equal(a0[i, : sz[i]], a1[i, : sz[i]])] == []
        for i in (0, 255, 511):
            assert a1[i, : sz[i]].tobytes() == oracle.encode(oracle.synth(61000 + i), q), f"q{q} image {i}"
    e.close(); base.close()


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    """The driver's hooks back to back in one interpreter: build() loads the C ABI library, smoke() then brings torch in.  A process must
    have one HIP runtime -- load_library() imports torch first so that it is torch's (loaded the other way round, neither finds a device)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "smoke OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_device_only_handle_skips_the_host_staging(oracle):
    """nhw_enc_create_ex(NHW_CREATE_DEVICE_ONLY): a handle for nhw_enc_batch_device does not hold the host path's 1.8 MB of staging per image;
    the resident path gives the oracle's bytes, and a host-path call on such a handle still works (it allocates then).  Unknown flags are refused."""
    import ctypes
    import torch
    import nhwcodec_amd
    n = 512
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    e_full = nhwcodec_amd.Encoder(0, max_batch=n)
    free1 = torch.cuda.mem_get_info()[0]
    e_full.close()
    e = nhwcodec_amd.Encoder(0, max_batch=n, device_only=True)
    free2 = torch.cuda.mem_get_info()[0]
    held_full, held_dev = free0 - free1, free0 - free2
    assert held_full - held_dev > n * 1_700_000, (held_full, held_dev)      # 3 x 512 x 512 in + 2 x the output slot a picture
    img = e.synth_device(8, seed_base=40)
    o, sizes, status = e.encode_device(img, 20)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0
    sz = sizes.cpu().numpy()
    host = img.cpu().numpy()
    for i in (0, 7):
        assert o[i, : sz[i]].cpu().numpy().tobytes() == oracle.encode(host[i], 20)
    got = e.encode(host[:4], 10)                                            # the host path after all: allocates on this call
    assert got[3] == oracle.encode(host[3], 10)
    e.close()
    h = ctypes.c_void_p()
    assert nhwcodec_amd.load_library().nhw_enc_create_ex(0, 4, 2, ctypes.byref(h)) != 0
