"""nhwcodec_amd -- MI355X-native NHW encoder hot path.

Host-side mirror of the reference's C interface (rcanut/nhwcodec encoder/codec.h:184-219: quality setting in,
512x512 BGR24 image in, .nhw bytes out) over the C ABI of libnhwhip.so (include/nhw_hip.h).  PyTorch is
used only as plumbing: device buffers, streams, torch.distributed.  There is no CPU path: if the HIP
library is missing or no GPU is visible, construction fails.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnhwhip.so")
IMG_BYTES = 786432
OUT_STRIDE = 512 << 10
QUALITY_DEFAULT = 20          # reference nhw_encoder_cli.c:95 (NORM)
# per-image / call status (include/nhw_hip.h)
NHW_OK, NHW_E_QUALITY, NHW_E_CODEBOOK, NHW_E_SPACE, NHW_E_ARG, NHW_E_HIP, NHW_E_FORMAT = 0, -1, -2, -3, -4, -5, -6

P = ctypes.c_void_p


class Timing(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("total_ms", "front_ms", "color_dwt_ms", "luma_ms", "chroma_ms", "entropy_ms")] + [("parts", ctypes.c_int), ("front_images", ctypes.c_int), ("prefilter_ms", ctypes.c_float)]


class NhwError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise NhwError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback")
    # torch brings its own HIP runtime; a process must have one.  Loaded after torch, libnhwhip.so binds to the runtime that is there; loaded
    # before it, the system's runtime comes in first and torch (or this library) then finds no device (`g.build(); g.smoke()` in one process).
    import torch  # noqa: F401
    L = ctypes.CDLL(path)
    L.nhw_last_error.restype = ctypes.c_char_p
    L.nhw_enc_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(P)]
    L.nhw_enc_create_ex.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(P)]
    L.nhw_enc_destroy.argtypes = [P]
    L.nhw_quality_supported.argtypes = [ctypes.c_int]
    L.nhw_enc_set_compat.argtypes = [P, ctypes.c_int]
    L.nhw_enc_batch_device.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, P, P, P]
    L.nhw_enc_batch.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, ctypes.c_size_t, P, P]
    L.nhw_synth_batch_device.argtypes = [P, P, ctypes.c_int, ctypes.c_uint32, P]
    L.nhw_enc_last_timing.argtypes = [P, ctypes.POINTER(Timing)]
    L.nhw_enc_synth_batch.argtypes = [P, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, P, ctypes.c_size_t, P, P]
    L.nhw_host_alloc.restype = P
    L.nhw_host_alloc.argtypes = [ctypes.c_size_t]
    L.nhw_host_free.argtypes = [P]
    L.nhw_device_count.restype = ctypes.c_int
    L.nhw_stage_color.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, P, P, P]
    L.nhw_stage_prefilter.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P]
    L.nhw_stage_chroma_l1.argtypes = [P, ctypes.c_int, P]
    L.nhw_stage_analysis.argtypes = [P, P, P, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    L.nhw_stage_synthesis.argtypes = [P, P, P, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, P]
    return L


class Encoder:
    """One encoder handle on one GPU.  encode_device() works on torch CUDA tensors already in HBM."""

    def __init__(self, device: int = 0, max_batch: int = 64, device_only: bool = False):
        """device_only: the handle is for encode_device() -- no staging buffers of the host path (nhw_enc_create_ex, NHW_CREATE_DEVICE_ONLY)"""
        import torch
        if not torch.cuda.is_available():
            raise NhwError("no GPU visible: nhwcodec_amd has no CPU path")
        self.torch = torch
        self.lib = load_library()
        self.device = device
        self.max_batch = max_batch
        h = P()
        self._chk(self.lib.nhw_enc_create_ex(device, max_batch, 1 if device_only else 0, ctypes.byref(h)))
        self.h = h

    def _chk(self, rc):
        if rc != 0:
            raise NhwError(f"libnhwhip rc={rc}: {self.lib.nhw_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.free_pinned()        # page-locked buffers handed out by pinned_images() end with the handle
            self.lib.nhw_enc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_compat(self, glibc_oneshot: bool):
        """False: canonical output (default).  True: reproduce the stock one-image-per-process binary (include/nhw_hip.h)."""
        self._chk(self.lib.nhw_enc_set_compat(self.h, 1 if glibc_oneshot else 0))

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def synth_device(self, n: int, seed_base: int = 0):
        t = self.torch.empty((n, 512, 512, 3), dtype=self.torch.uint8, device=f"cuda:{self.device}")
        with _OnTorchStream(self) as st:
            self._chk(self.lib.nhw_synth_batch_device(self.h, t.data_ptr(), n, seed_base, st))
        return t

    def alloc_out(self, n: int):
        dev = f"cuda:{self.device}"
        return (self.torch.empty((n, OUT_STRIDE), dtype=self.torch.uint8, device=dev),
                self.torch.empty(n, dtype=self.torch.int32, device=dev),
                self.torch.empty(n, dtype=self.torch.int32, device=dev))

    def encode_device(self, bgr, quality: int = QUALITY_DEFAULT, out=None):
        """bgr: uint8 CUDA tensor [n,512,512,3] (BMP file order).  Returns (out[n,OUT_STRIDE], sizes[n], status[n]) on device."""
        if not (bgr.is_cuda and bgr.dtype == self.torch.uint8 and bgr.is_contiguous() and bgr.dim() == 4 and tuple(bgr.shape[1:]) == (512, 512, 3)):
            raise NhwError("encode_device wants a contiguous uint8 CUDA tensor of shape [n, 512, 512, 3]")
        if bgr.device.index != self.device:
            raise NhwError(f"the batch is on cuda:{bgr.device.index}, this encoder on cuda:{self.device}")
        n = bgr.shape[0]
        if out is None:
            out = self.alloc_out(n)
        o, sizes, status = out
        for t_, dt_, cnt_ in ((o, self.torch.uint8, n * OUT_STRIDE), (sizes, self.torch.int32, n), (status, self.torch.int32, n)):
            if not (t_.is_cuda and t_.device.index == self.device and t_.dtype == dt_ and t_.is_contiguous() and t_.numel() >= cnt_):
                raise NhwError("encode_device: output tensors must be contiguous, on this encoder's device, uint8 [n, OUT_STRIDE] / int32 [n] / int32 [n]")
        with _OnTorchStream(self) as st:
            self._chk(self.lib.nhw_enc_batch_device(self.h, bgr.data_ptr(), n, quality, o.data_ptr(), sizes.data_ptr(), status.data_ptr(), st))
        return o, sizes, status

    def encode(self, images, quality: int = QUALITY_DEFAULT):
        """images: numpy uint8 [n,512,512,3] on the host -> list of .nhw byte strings (raises on a per-image failure)."""
        import numpy as np
        images = np.asarray(images)
        if images.dtype != np.uint8 or images.ndim != 4 or images.shape[1:] != (512, 512, 3):
            raise NhwError(f"encode wants uint8 [n, 512, 512, 3] (BMP file order), got {images.dtype} {images.shape}")
        images = np.ascontiguousarray(images)
        n = images.shape[0]
        arena = np.empty(n * OUT_STRIDE, np.uint8)
        offs = np.empty(n + 1, np.uint64)
        status = np.empty(n, np.int32)
        self._chk(self.lib.nhw_enc_batch(self.h, images.ctypes.data, n, quality, arena.ctypes.data, arena.size, offs.ctypes.data, status.ctypes.data))
        if (status != 0).any():
            raise NhwError(f"per-image status {status.tolist()}")
        return [arena[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]

    def encode_tiled(self, big, quality: int = QUALITY_DEFAULT):
        """a picture whose sides are multiples of 512 -> (list of .nhw byte strings, one per tile, row-major, (ny, nx)); `nhw-enc --tiles`"""
        tiles, shape = tile_images(big)
        return self.encode(tiles, quality), shape

    def timing(self) -> Timing:
        t = Timing()
        self._chk(self.lib.nhw_enc_last_timing(self.h, ctypes.byref(t)))
        return t

    def encode_synthetic(self, n: int, seed_base: int, quality: int = QUALITY_DEFAULT):
        """SURVEY 8(d) images seed_base.. generated on the device, encoded, files brought to the host (`nhw-enc --synthetic`)."""
        import numpy as np
        arena = np.empty(n * OUT_STRIDE, np.uint8)
        offs = np.empty(n + 1, np.uint64)
        status = np.empty(n, np.int32)
        self._chk(self.lib.nhw_enc_synth_batch(self.h, n, seed_base, quality, arena.ctypes.data, arena.size, offs.ctypes.data, status.ctypes.data))
        if (status != 0).any():
            raise NhwError(f"per-image status {status.tolist()}")
        return [arena[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]

    def pinned_images(self, n: int):
        """uint8 [n,512,512,3] in page-locked host memory (nhw_host_alloc): encode() uploads such a batch by DMA at PCIe speed while the
        chunk before is being encoded.  Keep the returned array alive only as long as this encoder; free with free_pinned()."""
        import numpy as np
        p = self.lib.nhw_host_alloc(n * IMG_BYTES)
        if not p:
            raise NhwError("nhw_host_alloc failed")
        buf = (ctypes.c_uint8 * (n * IMG_BYTES)).from_address(p)
        a = np.frombuffer(buf, np.uint8).reshape(n, 512, 512, 3)
        self._pinned = getattr(self, "_pinned", []) + [p]
        return a

    def free_pinned(self):
        for p in getattr(self, "_pinned", []):
            self.lib.nhw_host_free(p)
        self._pinned = []


def tile_images(big):
    """a uint8 [512*ny, 512*nx, 3] picture (BMP file row order) -> ([ny*nx, 512, 512, 3] independent tiles, row-major, (ny, nx)).
    Every tile is a picture of its own to the codec: no state crosses a tile edge, in either direction."""
    import numpy as np
    big = np.asarray(big)
    if big.dtype != np.uint8 or big.ndim != 3 or big.shape[2] != 3 or big.shape[0] < 512 or big.shape[1] < 512 or big.shape[0] % 512 or big.shape[1] % 512:
        raise NhwError(f"tiling wants uint8 [512*ny, 512*nx, 3], got {big.dtype} {big.shape}")
    ny, nx = big.shape[0] // 512, big.shape[1] // 512
    return np.ascontiguousarray(big.reshape(ny, 512, nx, 512, 3).transpose(0, 2, 1, 3, 4)).reshape(ny * nx, 512, 512, 3), (ny, nx)


def untile_images(tiles, ny: int, nx: int):
    """the inverse of tile_images"""
    import numpy as np
    tiles = np.asarray(tiles)
    if tiles.shape != (ny * nx, 512, 512, 3):
        raise NhwError(f"untile wants [{ny * nx}, 512, 512, 3], got {tiles.shape}")
    return np.ascontiguousarray(tiles.reshape(ny, nx, 512, 512, 3).transpose(0, 2, 1, 3, 4)).reshape(ny * 512, nx * 512, 3)


class _OnTorchStream:
    """Stream-ordered launch next to torch: the C ABI takes a hipStream_t and reads NULL as "the handle's own stream", which torch's
    default stream (handle 0) would select by accident.  On the default stream the work goes to a side stream that waits for it
    and that it waits for afterwards, so callers see ordinary stream semantics either way."""

    def __init__(self, owner):
        self.o = owner

    def __enter__(self):
        t, o = self.o.torch, self.o
        self.cur = t.cuda.current_stream(o.device)
        if self.cur.cuda_stream != 0:
            self.side = None
            return self.cur.cuda_stream
        if getattr(o, "_side", None) is None:
            o._side = t.cuda.Stream(o.device)
        self.side = o._side
        self.side.wait_stream(self.cur)
        return self.side.cuda_stream

    def __exit__(self, *exc):
        if self.side is not None:
            self.cur.wait_stream(self.side)
        return False


class DecTiming(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("total_ms", "entropy_ms", "recon_ms")]


class Decoder:
    """One decoder handle on one GPU: mirror of the reference's decode_image + write_image_bmp
    (decoder/nhw_decoder.c:54, decoder/nhw_decoder_cli.c:108) for batches of .nhw files."""

    def __init__(self, device: int = 0, max_batch: int = 64, device_only: bool = False):
        """device_only: the handle is for encode_device() -- no staging buffers of the host path (nhw_enc_create_ex, NHW_CREATE_DEVICE_ONLY)"""
        import torch
        if not torch.cuda.is_available():
            raise NhwError("no GPU visible: nhwcodec_amd has no CPU path")
        self.torch = torch
        self.lib = L = load_library()
        L.nhw_dec_last_error.restype = ctypes.c_char_p
        L.nhw_dec_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(P)]
        L.nhw_dec_destroy.argtypes = [P]
        L.nhw_dec_batch_device.argtypes = [P, P, P, P, ctypes.c_int, P, P, P, P]
        L.nhw_dec_batch.argtypes = [P, P, P, ctypes.c_int, P, P, P]
        L.nhw_dec_bmp_header.argtypes = [P]
        L.nhw_dec_last_timing.argtypes = [P, ctypes.POINTER(DecTiming)]
        L.nhw_dec_debug_stop_after.argtypes = [P, ctypes.c_int]
        L.nhw_dec_debug_read.argtypes = [P, ctypes.c_int, ctypes.c_int, P, ctypes.c_size_t]
        self.device = device
        self.max_batch = max_batch
        h = P()
        self._chk(L.nhw_dec_create(device, max_batch, ctypes.byref(h)))
        self.h = h

    def _chk(self, rc):
        if rc != 0:
            raise NhwError(f"libnhwhip rc={rc}: {self.lib.nhw_dec_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.nhw_dec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bmp_header(self) -> bytes:
        h = ctypes.create_string_buffer(54)
        self.lib.nhw_dec_bmp_header(ctypes.cast(h, P))
        return h.raw

    def decode_tiled(self, files, ny: int, nx: int):
        """the tiles written by encode_tiled -> uint8 [512*ny, 512*nx, 3]; `nhw-dec --tiles`"""
        if len(files) != ny * nx:
            raise NhwError(f"{len(files)} files for {ny} x {nx} tiles")
        px, _ = self.decode(files)
        return untile_images(px, ny, nx)

    def timing(self) -> DecTiming:
        t = DecTiming()
        self._chk(self.lib.nhw_dec_last_timing(self.h, ctypes.byref(t)))
        return t

    def decode_device(self, arena, offsets, lengths, out=None):
        """arena: uint8 CUDA tensor holding the files; offsets: int64 CUDA tensor [n]; lengths: int32 CUDA tensor [n]
        (the encoder's output arena with offsets i*OUT_STRIDE and its sizes tensor fits as is).
        Returns (pixels[n,512,512,3], status[n], quality[n]) on the device."""
        t = self.torch
        n = offsets.numel()
        dev = f"cuda:{self.device}"
        for name, x, dt in (("arena", arena, t.uint8), ("offsets", offsets, t.int64), ("lengths", lengths, t.int32)):
            if not (x.is_cuda and x.device.index == self.device and x.dtype == dt and x.is_contiguous()):
                raise NhwError(f"decode_device: `{name}` must be a contiguous {dt} tensor on cuda:{self.device}")
        if lengths.numel() != n or n < 1:
            raise NhwError("decode_device: offsets and lengths must have one entry per file")
        if out is None:
            out = t.empty((n, 512, 512, 3), dtype=t.uint8, device=dev)
        elif not (out.is_cuda and out.device.index == self.device and out.dtype == t.uint8 and out.is_contiguous() and out.numel() >= n * IMG_BYTES):
            raise NhwError("decode_device: `out` must be a contiguous uint8 tensor of n*786432 bytes on this decoder's device")
        status = t.empty(n, dtype=t.int32, device=dev)
        quality = t.empty(n, dtype=t.int32, device=dev)
        with _OnTorchStream(self) as st:
            self._chk(self.lib.nhw_dec_batch_device(self.h, arena.data_ptr(), offsets.data_ptr(), lengths.data_ptr(), n, out.data_ptr(), status.data_ptr(),
                                                    quality.data_ptr(), st))
        return out, status, quality

    def decode(self, files):
        """files: list of .nhw byte strings -> (uint8 [n,512,512,3] in nhw-dec's output byte order, quality list)."""
        import numpy as np
        n = len(files)
        offs = np.zeros(n + 1, np.uint64)
        offs[1:] = np.cumsum([len(f) for f in files])
        blob = np.frombuffer(b"".join(files), np.uint8)
        out = np.empty((n, 512, 512, 3), np.uint8)
        status = np.empty(n, np.int32)
        quality = np.empty(n, np.int32)
        self._chk(self.lib.nhw_dec_batch(self.h, blob.ctypes.data, offs.ctypes.data, n, out.ctypes.data, status.ctypes.data, quality.ctypes.data))
        if (status != 0).any():
            raise NhwError(f"per-file status {status.tolist()}")
        return out, quality.tolist()
