"""ctypes binding of oracle/liboracle.so (the plain-C CPU restatement).  Test infrastructure only."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle.so")
P = ctypes.c_void_p


class _Trace(ctypes.Structure):
    _fields_ = [("buf", P), ("cap", ctypes.c_size_t), ("len", ctypes.c_size_t), ("count", ctypes.c_int)]


class Oracle:
    def __init__(self, so_path: str = SO):
        L = self.lib = ctypes.CDLL(so_path)
        L.nhwo_synth_image.argtypes = [ctypes.c_uint32, P]
        L.nhwo_color.argtypes = [P, ctypes.c_int, P, P, P]
        L.nhwo_prefilter.argtypes = [P, ctypes.c_int]
        L.nhwo_analysis.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
        L.nhwo_synthesis.argtypes = [P, P, ctypes.c_int, ctypes.c_int]
        L.nhwo_quality_supported.argtypes = [ctypes.c_int]
        L.nhwo_encode.restype = ctypes.c_int
        L.nhwo_encode.argtypes = [P, ctypes.c_int, P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(_Trace)]
        L.nhwo_decode.restype = ctypes.c_int
        L.nhwo_decode.argtypes = [P, ctypes.c_size_t, P, ctypes.POINTER(ctypes.c_int)]
        L.nhwo_decode_planes.restype = ctypes.c_int
        L.nhwo_decode_planes.argtypes = [P, ctypes.c_size_t, P, ctypes.POINTER(ctypes.c_int)]
        L.nhwo_dec_bmp_header.argtypes = [P]
        self._out = ctypes.create_string_buffer(1 << 20)

    def decode(self, nhw: bytes, planes: bool = False):
        """.nhw bytes -> uint8[512,512,3] in nhw-dec's output byte order (planes=True: uint8[3,512,512] Y,U,V before the colour matrix)"""
        out = np.empty((3, 512, 512) if planes else (512, 512, 3), np.uint8)
        q = ctypes.c_int(0)
        buf = ctypes.create_string_buffer(bytes(nhw), len(nhw))
        fn = self.lib.nhwo_decode_planes if planes else self.lib.nhwo_decode
        rc = fn(ctypes.cast(buf, P), len(nhw), out.ctypes.data, ctypes.byref(q))
        if rc != 0:
            raise RuntimeError(f"oracle decode failed rc={rc}")
        return out, q.value

    def decode_probe(self, nhw: bytes, probe_id: int, cap: int = 1 << 20) -> bytes:
        """one intermediate buffer of the decode of `nhw` (ids: see probe() calls in nhwo_dec.c)"""
        buf = ctypes.create_string_buffer(cap)
        self.lib.nhwo_dec_probe.argtypes = [ctypes.c_int, P, ctypes.c_size_t]
        self.lib.nhwo_dec_probe_len.restype = ctypes.c_size_t
        self.lib.nhwo_dec_probe(probe_id, ctypes.cast(buf, P), cap)
        try:
            self.decode(nhw, planes=True)
            n = self.lib.nhwo_dec_probe_len()
        finally:
            self.lib.nhwo_dec_probe(0, None, 0)
        return buf.raw[:n]

    def bmp_header(self) -> bytes:
        h = ctypes.create_string_buffer(54)
        self.lib.nhwo_dec_bmp_header(ctypes.cast(h, P))
        return h.raw

    def set_oob_mode(self, glibc_oneshot: bool):
        """False: canonical (out-of-bounds reads see zeros).  True: NHWO_OOB_GLIBC_ONESHOT, the stock binary's heap neighbours (nhwo.h)."""
        ctypes.c_int.in_dll(self.lib, "nhwo_oob_mode").value = 1 if glibc_oneshot else 0

    def synth(self, seed: int) -> np.ndarray:
        b = np.empty((512, 512, 3), np.uint8)
        self.lib.nhwo_synth_image(seed, b.ctypes.data)
        return b

    def color(self, img, q):
        img = np.ascontiguousarray(img, np.uint8)
        y = np.empty(512 * 512, np.int16); u = np.empty(65536, np.uint8); v = np.empty(65536, np.uint8)
        self.lib.nhwo_color(img.ctypes.data, q, y.ctypes.data, u.ctypes.data, v.ctypes.data)
        return y, u, v

    def prefilter(self, y, q):
        y = np.ascontiguousarray(y, np.int16).copy()
        self.lib.nhwo_prefilter(y.ctypes.data, q)
        return y

    def analysis(self, jpeg, stride, n, final_level, keep=False):
        jpeg = np.ascontiguousarray(jpeg, np.int16).copy()
        proc = np.zeros_like(jpeg)
        k = np.zeros(2 * 65536, np.int16) if keep else None
        self.lib.nhwo_analysis(jpeg.ctypes.data, proc.ctypes.data, stride, n, final_level, k.ctypes.data if keep else None)
        return (jpeg, proc, k) if keep else (jpeg, proc)

    def synthesis(self, jpeg, stride, n):
        jpeg = np.ascontiguousarray(jpeg, np.int16).copy()
        proc = np.zeros_like(jpeg)
        self.lib.nhwo_synthesis(jpeg.ctypes.data, proc.ctypes.data, stride, n)
        return jpeg, proc

    def supported(self, q: int) -> bool:
        return bool(self.lib.nhwo_quality_supported(q))

    def encode(self, img, q, trace=False):
        from .harness import parse_trace
        img = np.ascontiguousarray(img, np.uint8)
        n = ctypes.c_size_t(0)
        tr = None
        tbuf = None
        if trace:
            tbuf = ctypes.create_string_buffer(96 << 20)
            tr = _Trace(ctypes.cast(tbuf, P), len(tbuf), 0, 0)
        rc = self.lib.nhwo_encode(img.ctypes.data, q, ctypes.cast(self._out, P), len(self._out), ctypes.byref(n), ctypes.byref(tr) if trace else None)
        if rc != 0:
            raise RuntimeError(f"oracle encode failed rc={rc}")
        data = self._out.raw[: n.value]
        if trace:
            return data, parse_trace(tbuf.raw[: tr.len], tr.count)
        return data
