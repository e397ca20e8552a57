/*
 * nhwo.h -- CPU ORACLE for the NHW encode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference encoder's algorithm (rcanut/nhwcodec,
 * /root/reference/encoder/), written from scratch for this repo.  Every function cites the
 * reference file:line it follows.  It exists to CHECK the HIP product path; nothing under
 * nhwcodec_amd/ or include/ may include, link, load or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
 *
 * Parity pin: validated stage by stage (checkpoint trace) and end to end (.nhw bytes) against
 * oracle/_ref = the unmodified reference sources linked with a zero-fill / zero-guard allocator
 * ("canonical" build), and against the committed golden
 * vectors in tests/golden/ that were generated from that build.
 *
 * Out-of-bounds model: ZERO.  The reference reads a few hundred bytes outside several of its
 * heap blocks; here every logical buffer is carved with the reference's own size and has
 * NHWO_GUARD zero bytes on both sides, so those reads return 0 exactly as in the canonical build.
 *
 * Supported quality settings in this revision: 17..23 (see nhwo_quality_supported()).
 */
#ifndef NHWO_H
#define NHWO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NHWO_DIM   512              /* image is NHWO_DIM x NHWO_DIM */
#define NHWO_HALF  256              /* reference IM_DIM  (encoder/codec.h:61) */
#define NHWO_QSIZE 65536            /* reference IM_SIZE (encoder/codec.h:58) */
#define NHWO_IMG_BYTES (NHWO_DIM * NHWO_DIM * 3)
#define NHWO_GUARD 4096
#define NHWO_MAX_OUT (1 << 20)

enum {
	NHWO_OK = 0,
	NHWO_E_QUALITY = -1,      /* quality setting not supported by this revision */
	NHWO_E_CODEBOOK = -2,     /* reference would exit(-1): compress_pixel.c:234,270,271 */
	NHWO_E_SPACE = -3,        /* output buffer too small */
	NHWO_E_ALLOC = -4
};

/* checkpoint trace, same record layout as oracle/ref/ref_shim.c */
typedef struct {
	uint8_t *buf;
	size_t cap, len;
	int count;
} nhwo_trace;

int nhwo_quality_supported(int quality);

/* What out-of-bounds reads of the reference return.  NHWO_OOB_ZERO (default) is the canonical, normative model (every such read
 * is 0).  NHWO_OOB_GLIBC_ONESHOT reproduces the heap adjacency of the stock `gcc -O3` nhw-enc run on one image per process
 * (SURVEY.md App. D): a compatibility mode for comparing against that binary; its output equals the binary's except for the
 * un-initialised padding bytes at the end of the res*_word / select_word sections. */
enum { NHWO_OOB_ZERO = 0, NHWO_OOB_GLIBC_ONESHOT = 1 };
extern int nhwo_oob_mode;

/* Whole encoder: BGR24 (BMP file order, 786432 bytes) -> .nhw bytes.  trace may be NULL. */
int nhwo_encode(const uint8_t *bgr, int quality, uint8_t *out, size_t cap, size_t *out_len, nhwo_trace *trace);

/* SURVEY.md section 8d synthetic image (integer generator), writes 786432 bytes. */
void nhwo_synth_image(uint32_t seed, uint8_t *bgr);

/* ---- stage-level entry points (used by the per-kernel parity tests) ---- */

/* a1: colorspace.c:55-260.  y: short[512*512]; u,v: uint8[256*256]. */
void nhwo_color(const uint8_t *bgr, int quality, int16_t *y, uint8_t *u, uint8_t *v);
/* a2: image_processing.c:558-2426 (q 17..21 branch).  In place on y[512*512]. */
void nhwo_prefilter(int16_t *y, int quality);
/* a3..a6: wavelet_filterbank.c:52-302.  jpeg/proc: planes of stride `stride`, n = transform size.
 * final_level != 0 <=> last_stage == wvlts_order-1 (no LL copy-back).  keep (may be NULL): receives the
 * first 256 rows x 512 of the transposed pass-1 plane (q>=22, level 0; wavelet_filterbank.c:107-112). */
void nhwo_analysis(int16_t *jpeg, int16_t *proc, int stride, int n, int final_level, int16_t *keep);
/* a7: wavelet_filterbank.c:305-496. */
void nhwo_synthesis(int16_t *jpeg, int16_t *proc, int stride, int n);

/* ---- decoder (BASELINE config 5; nhwo_dec.c) ---- */
/* .nhw bytes -> 786432 bytes in the order the reference's nhw-dec writes them behind its 54-byte BMP header */
int nhwo_decode(const uint8_t *nhw, size_t len, uint8_t *bgr, int *quality);
/* checkpoint before the colour matrix: planes = Y, U, V, 262144 bytes each (decode_image's im_bufferY/U/V) */
int nhwo_decode_planes(const uint8_t *nhw, size_t len, uint8_t *planes, int *quality);
void nhwo_dec_color(const uint8_t *y, const uint8_t *u, const uint8_t *v, int q, uint8_t *out);
void nhwo_dec_bmp_header(uint8_t h[54]);

#ifdef __cplusplus
}
#endif
#endif
