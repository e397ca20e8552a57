"""Host-side pin of the decoder's marker / payload arithmetic (nhwcodec_amd/csrc/nhw_dec.hip alt_starts).

Three byte-serial walks of the decoder -- the LL2 DPCM bytes (nhw_decoder.c:1661-1877: a 64..127 byte takes the next byte as its payload),
the packed code book's repeat marker and the book's two-byte entries (compress_pixel.c:86-117, :456-478) -- share one rule: a marker-valued
byte that is not itself the payload of the marker before it is a marker.  The reference decides that byte by byte; the kernels take 64 bytes
at a time and find the real markers with two 64-bit additions (inside a run of marker-valued bytes every second one is a marker, counted
from the run's first byte).  This test restates both forms and compares them on random and on exhaustive short masks."""
import random

M64 = (1 << 64) - 1


def by_walk(T, pending):
    """the reference's order: visit the marker-valued bytes left to right, skip the ones that are payloads"""
    payload = 1 if pending else 0
    starters = 0
    for l in range(64):
        if not (T >> l) & 1 or (payload >> l) & 1:
            continue
        starters |= 1 << l
        payload |= 1 << (l + 1)
    return starters


def by_carry(T, pending):
    """the kernel's form (alt_starts)"""
    Tm = T & ~1 & M64 if pending else T
    rs = Tm & ~(Tm << 1) & M64
    ev = 0x5555555555555555
    return ((Tm & ~((Tm + (rs & ev)) & M64) & ev) | (Tm & ~((Tm + (rs & ~ev & M64)) & M64) & ~ev)) & M64


def test_marker_arithmetic_equals_the_byte_walk():
    for T in range(1 << 12):                                        # every pattern of the first 12 bytes, and of the last 12
        for pending in (False, True):
            assert by_walk(T, pending) == by_carry(T, pending)
            hi = T << 52
            assert by_walk(hi, pending) == by_carry(hi, pending)
    rnd = random.Random(7)
    for _ in range(50000):
        dens = rnd.random()
        T = sum(1 << b for b in range(64) if rnd.random() < dens)
        for pending in (False, True):
            assert by_walk(T, pending) == by_carry(T, pending), (hex(T), pending)
