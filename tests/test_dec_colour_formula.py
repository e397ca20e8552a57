"""Host-side pin of the decoder's integer colour matrix (quality >= 20, nhwcodec_amd/csrc/nhw_dec.hip yuv_to_bytes).

The kernel replaces the reference's double arithmetic (decoder/nhw_decoder_cli.c:133-283: (int)(Y + 1.402 V' + 0.5f) and its two sisters,
clipped to 0..255) by 24-bit integer multiplies and reciprocal multiplies.  This test restates exactly the kernel's integer arithmetic in
numpy and compares it with the double form on every one of the 2^24 (Y, U, V) triples; the ties the kernel hands to the double path are
counted.  (The kernel itself is compared with the oracle's decoder on whole files in tests/test_decode.py.)"""
import numpy as np


def test_integer_colour_matrix_equals_the_double_form_on_all_triples():
    n = np.arange(256000, dtype=np.uint64)
    assert ((n * 4294968) >> 32 == n // 1000).all()                 # the high half of the 24 x 24 bit product IS the quotient
    m = np.arange(800000, dtype=np.uint64)
    assert ((m * 1374390) >> 32 == m // 3125).all()
    half = np.float64(np.float32(0.5))
    U, V = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    U = U.ravel(); V = V.ravel()
    Ud = (U - 128).astype(np.float64); Vd = (V - 128).astype(np.float64)
    ties = 0
    for y in range(256):
        Yd = np.float64(y)
        ref = [np.clip(np.trunc(v).astype(np.int64), 0, 255) for v in (Yd + 1.402 * Vd + half, Yd - 0.34414 * Ud - 0.71414 * Vd + half, Yd + 1.772 * Ud + half)]
        nR = 1000 * y + 1402 * V + (500 - 1402 * 128)
        nB = 1000 * y + 1772 * U + (500 - 1772 * 128)
        nG = 100000 * y - 34414 * U - 71414 * V + (50000 + (34414 + 71414) * 128)
        assert (np.abs(nG) < 1 << 31).all() and (np.abs(nR) < 1 << 23).all()
        R = ((np.clip(nR, 0, 255999).astype(np.uint64) * 4294968) >> 32).astype(np.int64)
        B = ((np.clip(nB, 0, 255999).astype(np.uint64) * 4294968) >> 32).astype(np.int64)
        gq = (((np.clip(nG, 0, 25599999) >> 5).astype(np.uint64) * 1374390) >> 32).astype(np.int64)
        tie = (nG > 0) & (gq * 100000 == nG)                        # the exact value is an integer: the kernel takes the double path
        ties += int(tie.sum())
        G = np.where(tie, ref[1], gq)
        assert (R == ref[0]).all() and (G == ref[1]).all() and (B == ref[2]).all(), y
    assert 0 < ties < 1000
