"""Test-infrastructure helpers (oracle side only; never imported by the product).

* synth_image(seed): the SURVEY.md section 8d integer generator ("smooth+noise" class), numpy/pure-Python.
* RefEncoder: ctypes binding of oracle/_ref/libnhwref_enc.so = the UNMODIFIED reference encoder
  (/root/reference/encoder/*.c) linked with oracle/ref/ref_shim.c (canonical zero-guard allocator +
  --wrap checkpoints).  Built by `make -C oracle/ref`; prebuilt .so travels to the GPU box.
"""
import ctypes
import os
import struct
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libnhwref_enc.so")
IMG_BYTES = 512 * 512 * 3


def synth_image(seed: int) -> np.ndarray:
    """SURVEY.md section 8d generator. Returns uint8[512,512,3] in BMP file order (rows as stored, B,G,R)."""
    x = (0x9E3779B9 * (seed + 1)) & 0xFFFFFFFF
    if x == 0:
        x = 1
    n = 17 * 17 * 3 + IMG_BYTES
    out = np.empty(n, dtype=np.uint32)
    # xorshift32 is strictly sequential; plain loop (about 1 s / image). Tests cache results.
    for i in range(n):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        out[i] = x
    lat = (out[: 17 * 17 * 3] >> 24).astype(np.int64).reshape(17, 17, 3)
    noise = (out[17 * 17 * 3 :] % 13).astype(np.int64).reshape(512, 512, 3) - 6
    ys = np.arange(512)
    gy = ys >> 5
    wy = ((ys & 31) << 3)[:, None, None]
    gx = ys >> 5
    wx = ((ys & 31) << 3)[None, :, None]
    l00 = lat[gy][:, gx]
    l01 = lat[gy][:, gx + 1]
    l10 = lat[gy + 1][:, gx]
    l11 = lat[gy + 1][:, gx + 1]
    v = ((l00 * (256 - wx) + l01 * wx) * (256 - wy) + (l10 * (256 - wx) + l11 * wx) * wy + 32768) >> 16
    return np.clip(v + noise, 0, 255).astype(np.uint8)


def bmp_bytes(img: np.ndarray) -> bytes:
    """54-byte BITMAPINFOHEADER + pixel rows in file order, height +512."""
    hdr = struct.pack("<2sIHHIIiiHHIIiiII", b"BM", 54 + IMG_BYTES, 0, 0, 54, 40, 512, 512, 1, 24, 0, IMG_BYTES, 0, 0, 0, 0)
    return hdr + img.tobytes()


def parse_trace(buf: bytes, count: int):
    """-> list of (name, [blob bytes, ...]) in call order."""
    out, off = [], 0
    for _ in range(count):
        name = buf[off : off + 32].split(b"\0", 1)[0].decode()
        nblobs, *lens = struct.unpack_from("<6I", buf, off + 32)
        off += 32 + 24
        blobs = []
        for i in range(nblobs):
            blobs.append(bytes(buf[off : off + lens[i]]))
            off += lens[i]
        out.append((name, blobs))
    return out


class RefEncoder:
    def __init__(self, so_path: str = REF_SO):
        self.lib = ctypes.CDLL(so_path)
        self.lib.nhwref_encode.restype = ctypes.c_int
        self.lib.nhwref_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        self.lib.nhwref_trace_begin.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        self.lib.nhwref_trace_end.restype = ctypes.c_size_t
        self.lib.nhwref_trace_end.argtypes = [ctypes.POINTER(ctypes.c_int)]
        self.lib.nhwref_encode_file.restype = ctypes.c_int
        self.lib.nhwref_encode_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        fd, self.tmp = tempfile.mkstemp(suffix=".nhw")
        os.close(fd)
        self._out = ctypes.create_string_buffer(1 << 20)

    def __del__(self):
        try:
            os.unlink(self.tmp)
        except OSError:
            pass

    def encode(self, img: np.ndarray, quality: int, trace: bool = False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.size == IMG_BYTES
        tbuf = None
        if trace:
            tbuf = ctypes.create_string_buffer(96 << 20)
            self.lib.nhwref_trace_begin(tbuf, len(tbuf))
        n = ctypes.c_size_t(0)
        rc = self.lib.nhwref_encode(img.ctypes.data, quality, self.tmp.encode(), self._out, len(self._out), ctypes.byref(n))
        tr = None
        if trace:
            cnt = ctypes.c_int(0)
            tl = self.lib.nhwref_trace_end(ctypes.byref(cnt))
            tr = parse_trace(tbuf.raw[:tl], cnt.value)
        if rc != 0:
            raise RuntimeError(f"reference encoder failed rc={rc}")
        data = self._out.raw[: n.value]
        return (data, tr) if trace else data


REF_DEC_SO = os.path.join(HERE, "_ref", "libnhwref_dec.so")
STOCK_ENC = os.path.join(HERE, "_ref", "nhw-enc")      # the reference encoder exactly as its README builds it (gcc *.c -O3), no shim


def stock_encode(img: np.ndarray, quality: int) -> bytes:
    """Run the stock reference binary on one image (one process per image: the heap layout SURVEY.md App. D describes)."""
    import subprocess
    d = tempfile.mkdtemp(prefix="nhwstock")
    try:
        with open(os.path.join(d, "a.bmp"), "wb") as fh:
            fh.write(bmp_bytes(img))
        subprocess.run([STOCK_ENC, f"-q{quality}", os.path.join(d, "a.bmp"), os.path.join(d, "a.nhw")], capture_output=True, check=True)
        with open(os.path.join(d, "a.nhw"), "rb") as fh:
            return fh.read()
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def uninitialised_positions(nhw: bytes) -> set:
    """Byte offsets of a .nhw file whose value the stock encoder itself leaves to un-initialised memory (located from
    the header): the last byte of the res1 / res5 / res6 word sections, the last two of res3's, the last byte of the two select-word
    sections, and the last byte of the two code books (the collapse loop at compress_pixel.c:410-421 looks one stack byte past them)."""
    q = nhw[1]
    o = 2
    f = {}

    def g(n, name):
        nonlocal o
        f[name] = int.from_bytes(nhw[o:o + n], "little")
        o += n
    g(2, "book1"); g(2, "book2"); g(4, "data1"); g(4, "data2"); g(2, "tree_end"); g(2, "exw")
    if q > 12: g(2, "res1")
    if q >= 19: g(2, "res3"); g(2, "res3b")
    if q > 17: g(2, "res4")
    if q > 12: g(2, "res1b")
    if q >= 21: g(2, "res5"); g(2, "res5b")
    if q > 21: g(4, "res6"); g(2, "res6b"); g(2, "char")
    if q > 22: g(2, "qs3")
    g(2, "sel1"); g(2, "sel2")
    if q > 15: g(2, "llword")
    g(2, "chres")
    pad = set()
    o += f["book1"]; pad.add(o - 1)
    o += f["book2"]; pad.add(o - 1)
    o += f["exw"]
    if q > 12: o += f["res1"] + 2 * f["res1b"]; pad.add(o - 1)
    if q > 17: o += f["res4"]
    if q >= 19: o += f["res3"] + 3 * f["res3b"]; pad.update((o - 1, o - 2))
    if q >= 21: o += f["res5"] + 2 * f["res5b"]; pad.add(o - 1)
    if q > 21: o += f["res6"] + 2 * f["res6b"]; pad.add(o - 1); o += 2 * f["char"]
    if q > 22: o += 4 * f["qs3"]
    o += f["sel1"]; pad.add(o - 1)
    o += f["sel2"]; pad.add(o - 1)
    return pad


class RefDecoder:
    """ctypes binding of oracle/_ref/libnhwref_dec.so = the UNMODIFIED reference decoder (/root/reference/decoder/*.c)
    linked with oracle/ref/ref_dec_shim.c (zero-guard allocator)."""

    def __init__(self, so_path: str = REF_DEC_SO):
        self.lib = ctypes.CDLL(so_path)
        self.lib.nhwref_decode_planes.restype = ctypes.c_int
        self.lib.nhwref_decode_planes.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        self.lib.nhwref_decode_bmp.restype = ctypes.c_int
        self.lib.nhwref_decode_bmp.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        self.dir = tempfile.mkdtemp(prefix="nhwrefdec")

    def __del__(self):
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)

    def _file(self, nhw: bytes) -> str:
        p = os.path.join(self.dir, "in.nhw")
        with open(p, "wb") as fh:
            fh.write(nhw)
        return p

    def planes(self, nhw: bytes):
        out = np.empty((3, 512, 512), np.uint8)
        q = ctypes.c_int(0)
        rc = self.lib.nhwref_decode_planes(self._file(nhw).encode(), out.ctypes.data, ctypes.byref(q))
        if rc != 0:
            raise RuntimeError(f"reference decoder failed rc={rc}")
        return out, q.value

    def bmp(self, nhw: bytes) -> bytes:
        o = os.path.join(self.dir, "out.bmp")
        rc = self.lib.nhwref_decode_bmp(self._file(nhw).encode(), o.encode())
        if rc != 0:
            raise RuntimeError(f"reference decoder failed rc={rc}")
        with open(o, "rb") as fh:
            return fh.read()


# ---------------------------------------------------------------- deterministic robustness classes
def _hash_u32(idx: np.ndarray, seed: int) -> np.ndarray:
    """Stateless integer hash (vectorised, identical on every numpy): used for robustness inputs."""
    h = (idx.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64((seed * 0x85EBCA77 + 0x1234567) & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return h.astype(np.uint32)


def class_image(kind: str, seed: int = 0) -> np.ndarray:
    """Secondary input classes of SURVEY.md section 8d: 'noise', 'blocks', 'flat', 'gradient', 'black', 'white', 'tiles'."""
    idx = np.arange(IMG_BYTES, dtype=np.uint64)
    if kind == "noise":
        return (_hash_u32(idx, seed) >> 24).astype(np.uint8).reshape(512, 512, 3)
    if kind == "flat":
        return np.full((512, 512, 3), 77, np.uint8)
    if kind == "black":
        return np.zeros((512, 512, 3), np.uint8)
    if kind == "white":
        return np.full((512, 512, 3), 255, np.uint8)
    if kind == "gradient":
        return np.broadcast_to((np.arange(512) // 2).astype(np.uint8)[None, :, None], (512, 512, 3)).copy()
    if kind == "blocks":
        h = _hash_u32(np.arange(40 * 7 + 3, dtype=np.uint64), seed + 99)
        img = np.empty((512, 512, 3), np.uint8)
        img[:] = (h[-3:] >> 24).astype(np.uint8)
        for k in range(40):
            y0, x0 = int(h[7 * k] % 480), int(h[7 * k + 1] % 480)
            hh, ww = 8 + int(h[7 * k + 2] % 192), 8 + int(h[7 * k + 3] % 192)
            img[y0 : y0 + hh, x0 : x0 + ww] = (h[7 * k + 4 : 7 * k + 7] >> 24).astype(np.uint8)
        return img
    if kind == "tiles":                                   # flat 48x48 tiles: LL2 runs of 12 equal samples -> the LL2 coder's mode 1
        h = _hash_u32(np.arange(11 * 11 * 3, dtype=np.uint64), seed + 7).reshape(11, 11, 3)
        t = (64 + (h >> 25)).astype(np.uint8)
        ys = np.arange(512) // 48
        return t[ys][:, ys].copy()
    raise ValueError(kind)
