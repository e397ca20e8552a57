/*
 * nhw_hip_debug.h -- the debug / test entry points of libnhwhip.so (C ABI).  Not part of the drop-in boundary (include/nhw_hip.h): the
 * parity tests use them to stop the batch driver after a stage and to read workspace buffers back, so that a mismatch against the
 * checkpoint trace of the reference ("checkpoint" flavour) is located at one stage instead of at the .nhw bytes.
 * tests/test_gpu_parity.py::test_c_abi_exports_every_declared_symbol checks every name declared here against the library, like the
 * boundary's own.
 */
#ifndef NHW_HIP_DEBUG_H
#define NHW_HIP_DEBUG_H

#include "nhw_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* encoder: the next batches stop after launch stage `stage` of the batch driver (0: run to the end).  Stage numbers follow the order of
 * the reference's cross-TU calls in encode_image (nhw_encoder.c:103-2878); tools/dev/gpu_low_debug.py lists them per quality. */
int nhw_debug_stop_after(nhw_enc *e, int stage);
/* encoder: every carry segment of the fused front kernel takes its exact replay (the path the look-back falls back to) */
int nhw_debug_front_fallback(nhw_enc *e, int on);
/* encoder: the first `bytes` bytes of workspace buffer `buf` (an index of the B_* list in nhwcodec_amd/csrc/nhw_ws.h) of image `img` -> host */
int nhw_debug_read(nhw_enc *e, int buf, int img, void *dst, size_t bytes);
/* encoder: an order-independent 64-bit digest of the first `bytes` bytes of buffer `buf`, one per image, into device memory (n x uint64) */
int nhw_debug_hash(nhw_enc *e, int buf, size_t bytes, int n, void *d_out, void *stream);

/* encoder: the first `bytes` bytes of buffer `buf` of the first n images set to `byte` (the zero guard behind the buffer is not touched): a test
 * that the production launch sequence -- which leaves out stores nothing reads -- never reads what an earlier batch left in a plane */
int nhw_debug_fill(nhw_enc *e, int buf, int byte, size_t bytes, int n);

/* decoder: the same two hooks (stage order: decode_image, decoder/nhw_decoder.c:54-1476; `what`: an index of the D_* list in nhw_dec.hip) */
void nhw_dec_debug_stop_after(nhw_dec *d, int stage);
int  nhw_dec_debug_read(nhw_dec *d, int what, int img, void *dst, size_t bytes);
/* decoder: the colour matrix of write_image_bmp (nhw_decoder_cli.c:133-283) for quality q on n (Y, U, V) byte triples in device memory (n a
 * multiple of 8) -> n x 3 bytes in the order the reference writes them, through the functions k_dec_final runs: the exhaustive 2^24 test */
int  nhw_dec_debug_colour(int quality, const void *d_yuv, void *d_rgb, int n);

#ifdef __cplusplus
}
#endif
#endif
