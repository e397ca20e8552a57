/*
 * nhw_hip.h -- C ABI of libnhwhip.so: the MI355X (gfx950) NHW encoder hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI; its de-facto C API
 * is encoder/codec.h:184-219 (`read_image_bmp` / `encode_image` / `write_compressed_file`, one
 * 512x512 image per call, errors by exit()).  The entry points below replace that trio for whole
 * batches of images:
 *
 *   reference                                            this library
 *   ---------------------------------------------------  ------------------------------------------
 *   im.setup->quality_setting (nhw_encoder_cli.c:175)     `quality` argument (1..23: every setting the reference has tables for)
 *   read_image_bmp  -> im_buffer4 (nhw_encoder.c:3047)    caller passes n x 786432 BGR24 bytes in BMP
 *                                                         file order (what fread at :3086 delivers)
 *   downsample_YUV420 + encode_image (codec.h:184,189)    nhw_enc_batch / nhw_enc_batch_device
 *   write_compressed_file (nhw_encoder.c:3100)            the .nhw bytes land in the output arena
 *   exit(-1) on code-book overflow (compress_pixel.c:234) per-image status NHW_E_CODEBOOK
 *
 * Plain C types only; device pointers are passed as void*; `stream` is a hipStream_t passed as
 * void* (NULL = the library's own stream).  One nhw_enc handle drives one GPU and is not
 * re-entrant; use one handle per host thread / process (one process per GPU under torchrun).
 */
#ifndef NHW_HIP_H
#define NHW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NHW_IMG_BYTES   786432u      /* 512*512*3, encoder/codec.h:58-61 */
#define NHW_OUT_STRIDE  (512u << 10) /* bytes reserved per image in the device output arena */

enum {
	NHW_OK = 0,
	NHW_E_QUALITY = -1,   /* quality outside the supported set */
	NHW_E_CODEBOOK = -2,  /* reference would exit(-1): compress_pixel.c:234,270,271 */
	NHW_E_SPACE = -3,     /* output arena too small */
	NHW_E_ARG = -4,
	NHW_E_HIP = -5,       /* a HIP call failed; see nhw_last_error() */
	NHW_E_FORMAT = -6     /* decoder: not a well-formed .nhw file (the reference prints "Not an .nhw file" and exits, or reads out of bounds) */
};

typedef struct nhw_enc nhw_enc;

/* stage timings of the last nhw_enc_batch_device call, measured with hipEvents on the launch stream */
typedef struct {
	float total_ms;
	float front_ms;       /* colour + pre-filter + level-1 analysis (the HBM-roofline kernels) */
	float color_dwt_ms;   /* quality 1..16: the colour + 4:2:0 kernel, the first of the front group; 0 for 17..23 (the conversion is inside the fused front kernel) */
	float luma_ms;        /* luma tail up to and including the luma quantiser and Y31; the chroma sequence runs next to it on a stream of its own and -- the default since round 5 -- is joined IN FRONT of the luma quantiser, so this figure includes any wait for it (NHW_CHROMA_FORK=0: the chroma sequence follows behind) */
	float chroma_ms;      /* what is left of the chroma sequence once the luma tail is done: about 0 with the default join in front of the quantiser (NHW_QUANT_JOIN=0: the join in front of the packetiser, where this is the exposed rest; NHW_CHROMA_FORK=0: its full time) */
	float entropy_ms;
	int parts;            /* sub-batches the stages behind the front ran as (each on a stream of its own); with more than one, luma/chroma/entropy_ms are those of the first */
	int front_images;     /* images covered by front_ms */
	float prefilter_ms;   /* quality 1..16: the rationed luma pre-filter (k_low_pre, k_low_mapfix, k_low_chain, k_low_apply, k_low_markrows, k_low_marks), second of the front group; 0 for 17..23 */
} nhw_timing;

/* lifecycle: replaces `im.setup=malloc(..)` + per-image mallocs of encode_image (nhw_encoder.c:108-...).  Everything a batch of up to
 * max_batch images needs on the device is allocated here and nowhere else: the workspace (7.0 MB per image) and the staging buffers of the
 * host path nhw_enc_batch / nhw_enc_synth_batch (1.8 MB per image) -- no call behind it allocates, so the first batch costs what the others do */
int  nhw_enc_create(int device, int max_batch, nhw_enc **out);
/* The same with flags.  NHW_CREATE_DEVICE_ONLY: for callers of nhw_enc_batch_device only -- the host path's staging buffers (1.8 MB per image:
 * 7.3 GB at max_batch 4096) are not allocated; a later nhw_enc_batch / nhw_enc_synth_batch on such a handle still works and allocates them
 * then (that one call pays for it).  Unknown flag bits are NHW_E_ARG. */
#define NHW_CREATE_DEVICE_ONLY 1u
int  nhw_enc_create_ex(int device, int max_batch, unsigned flags, nhw_enc **out);
void nhw_enc_destroy(nhw_enc *e);
const char *nhw_last_error(void);
int  nhw_quality_supported(int quality);

/* What the reference's out-of-bounds reads return.  NHW_COMPAT_CANONICAL (default, normative): zeros -- the output equals the reference
 * sources built with a zero-filling guard allocator.  NHW_COMPAT_GLIBC_ONESHOT: the heap neighbours of the stock `gcc -O3` nhw-enc run on
 * one image per process -- the output equals that binary's, except for the bytes it leaves un-initialised itself (the
 * last byte of the res1/res5/res6 word sections and of the two select-word and code-book sections, the last two of res3's). */
enum { NHW_COMPAT_CANONICAL = 0, NHW_COMPAT_GLIBC_ONESHOT = 1 };
int  nhw_enc_set_compat(nhw_enc *e, int mode);

/* Encode n images already resident in HBM.  d_bgr: n*NHW_IMG_BYTES.  d_out: n*NHW_OUT_STRIDE, image i's
 * .nhw starts at i*NHW_OUT_STRIDE.  d_sizes[i] = byte length, d_status[i] = NHW_OK / NHW_E_CODEBOOK.
 * Asynchronous on `stream`. */
int nhw_enc_batch_device(nhw_enc *e, const void *d_bgr, int n, int quality, void *d_out, uint32_t *d_sizes,
                         int32_t *d_status, void *stream);

/* Host convenience: H2D, encode, compact, D2H.  out_off has n+1 entries; image i is
 * out_arena[out_off[i] .. out_off[i+1]).  status has n entries.  Synchronous. */
int nhw_enc_batch(nhw_enc *e, const uint8_t *bgr, int n, int quality, uint8_t *out_arena, size_t arena_cap,
                  uint64_t *out_off, int32_t *status);

/* SURVEY.md section 8d synthetic inputs generated on the device (image i gets seed seed_base+i). */
int nhw_synth_batch_device(nhw_enc *e, void *d_bgr, int n, uint32_t seed_base, void *stream);
/* the same images generated on the device, encoded, and the .nhw files brought to the host like nhw_enc_batch does (`nhw-enc --synthetic`) */
int nhw_enc_synth_batch(nhw_enc *e, int n, uint32_t seed_base, int quality, uint8_t *out_arena, size_t arena_cap, uint64_t *out_off, int32_t *status);

/* page-locked host memory for nhw_enc_batch's input (DMA at PCIe speed, overlapped with the encode of the chunk before), and the number
 * of GPUs this process sees (one encoder handle per device, e.g. one host thread each: tools/nhw_enc.c --gpus) */
void *nhw_host_alloc(size_t bytes);
void  nhw_host_free(void *p);
int   nhw_device_count(void);

int nhw_enc_last_timing(nhw_enc *e, nhw_timing *t);

/* ---- stage-level entry points (kernel parity tests; same stream rules) ----
 * colour + 4:2:0 (colorspace.c:55-260), any quality 1..23: d_y n*262144 int16, d_u/d_v n*65536 uint8 */
int nhw_stage_color(nhw_enc *e, const void *d_bgr, int n, int quality, void *d_y, void *d_u, void *d_v, void *stream);
/* luma pre-filter (image_processing.c:558-2426) as a stage of its own: quality 1..16 (k_low_pre .. k_low_marks, the kernels the encoder runs, the whole batch in line);
 * for 17..21 it is a step inside the fused front kernel (NHW_E_QUALITY here).  In place on d_y */
int nhw_stage_prefilter(nhw_enc *e, void *d_y, int n, int quality, void *stream);
/* one analysis level (wavelet_filterbank.c:52-302) on planes of `stride` shorts per row, n_img images
 * spaced plane_stride shorts apart; size = transform size; final_level as in the oracle.
 * Size 256 / 128: any `short` input (blocks whose second pass would leave 16 bits take a 32-bit path that follows the reference's `int`
 * accumulators).  Size 512 is the encoder's level-1 kernel, built for luma, and has a DOMAIN: with U = the largest sample of the planes
 * (0 if none is positive) and L = minus the smallest (0 if none is negative),
 *       104 U + 40 L <= NHW_ANA512_BOUND   and   104 L + 40 U <= NHW_ANA512_BOUND
 * (the 2-D low-pass is the outer product of [-1 2 6 2 -1]: positive weights 104, negative 40; 47 below 32767 for the error-diffusion
 * carry and the rounding offset of filters.c:246-276; proof: nhwcodec_amd/csrc/nhw_front_image.h).  E.g. L = 4 allows U <= 313, L = 0
 * U <= 314; the encoder's luma is 0..255 plus at most +-4 of its pre-filters.  Planes outside the domain are refused with NHW_E_ARG
 * (a synchronising check on the stream): the entry point never returns a plane that differs from the reference's wavelet_analysis(). */
#define NHW_ANA512_BOUND 32720
int nhw_stage_analysis(nhw_enc *e, void *d_jpeg, void *d_proc, int n_img, size_t plane_stride, int stride, int size,
                       int final_level, void *stream);
int nhw_stage_synthesis(nhw_enc *e, void *d_jpeg, void *d_proc, int n_img, size_t plane_stride, int stride, int size,
                        void *stream);
/* the two chroma level-1 analyses (nhw_encoder.c:2265, 2576) as the encoder launches them for quality >= 15, on the 4:2:0 planes of the handle's
 * last batch (a measurement hook: bench.py adds their time to the fused front kernel's).  The launches are ordered behind that batch; NHW_E_ARG
 * when the handle's last whole batch had fewer than n images or a quality below 15 (nothing it could work on); it overwrites the chroma
 * work planes, so it belongs between batches */
int nhw_stage_chroma_l1(nhw_enc *e, int n, void *stream);

/* ---- decoder (BASELINE config 5): replaces decode_image + write_image_bmp (decoder/codec.h:184-186,
 * decoder/nhw_decoder.c:54, decoder/nhw_decoder_cli.c:108), one launch sequence per batch of files ----
 * d_nhw: an arena in HBM holding the .nhw files, file i at d_off[i] with d_len[i] bytes (device arrays; the encoder's output
 * arena is such an arena with d_off[i] = i*NHW_OUT_STRIDE and d_len = d_sizes).  d_bgr: n*NHW_IMG_BYTES,
 * the pixel bytes in the order nhw-dec writes them behind its 54-byte header (nhw_dec_bmp_header).  d_status[i] =
 * NHW_OK / NHW_E_FORMAT, d_quality[i] (may be NULL) = the quality setting stored in file i.  Asynchronous on `stream`. */
typedef struct nhw_dec nhw_dec;
int  nhw_dec_create(int device, int max_batch, nhw_dec **out);
void nhw_dec_destroy(nhw_dec *d);
const char *nhw_dec_last_error(void);
int nhw_dec_batch_device(nhw_dec *d, const void *d_nhw, const uint64_t *d_off, const uint32_t *d_len, int n, void *d_bgr, int32_t *d_status,
                         int32_t *d_quality, void *stream);
/* host convenience: H2D of the files (nhw[off[i]..off[i+1])), decode, D2H.  Synchronous. */
int nhw_dec_batch(nhw_dec *d, const uint8_t *nhw, const uint64_t *off, int n, uint8_t *bgr, int32_t *status, int32_t *quality);
void nhw_dec_bmp_header(uint8_t h[54]);
/* hipEvent timings of the last nhw_dec_batch_device call (events on its launch stream): the whole sequence, the entropy stages
 * (parse, prefix-code walk, un-zig-zag), the two level-1 luma synthesis passes and the colour kernel -- the last three are the kernels
 * SURVEY.md 8(d) prices against the HBM roofline for the decode path */
typedef struct { float total_ms, entropy_ms, recon_ms; } nhw_dec_timing;   /* recon_ms: the final reconstruction kernel (level-1 synthesis both ways + colour) */
int nhw_dec_last_timing(nhw_dec *d, nhw_dec_timing *t);

#ifdef __cplusplus
}
#endif
#endif
