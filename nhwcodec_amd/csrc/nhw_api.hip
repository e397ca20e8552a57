/*
 * nhw_api.hip -- the C ABI of libnhwhip.so (include/nhw_hip.h): workspace, batch driver, host
 * convenience path, stage entry points, hipEvent timing.  gfx950 / ROCm only.
 *
 * Batch driver = encode_image (rcanut/nhwcodec encoder/nhw_encoder.c:103-2878) re-cut as a sequence of
 * batch-wide kernel launches: every launch processes the same stage of all n images.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "../../include/nhw_hip.h"
#include "nhw_ws.h"

/* launchers (nhw_front.hip, nhw_tail.hip) */
void nhw_launch_color(const uint8_t *bgr, int n, int q, int16_t *y, size_t y_stride, uint8_t *u, uint8_t *v, size_t c_stride, hipStream_t s);
void nhw_launch_analysis(int16_t *jpeg, int16_t *proc, int n, size_t plane_stride, int stride, int size, int final_level, hipStream_t s,
                         int16_t *save = nullptr, size_t save_plane = 0, int save_row = 0, int save_kind = 0, const uint8_t *src8 = nullptr, size_t src8_plane = 0, int drop_t = 0,
                         const int16_t *alt = nullptr, size_t alt_plane = 0, int alt_stride = 0);
void nhw_launch_synthesis(int16_t *jpeg, int16_t *proc, int n, size_t plane_stride, int stride, int size, hipStream_t s, int drop_nat = 0,
                          const uint16_t *verb_list = nullptr, size_t verb_list_stride = 0, const int *verb_len = nullptr, size_t verb_len_stride = 0);
void nhw_launch_l2_recon(int16_t *jpeg, const int16_t *proc, size_t plane_stride, int16_t *ll1, size_t ll1_stride, int n, hipStream_t s);
void nhw_launch_synth(uint8_t *bgr, int n, uint32_t seed_base, hipStream_t s);
void nhw_launch_front_fused(const uint8_t *bgr, int q, uint8_t *pu, uint8_t *pv, size_t c_stride, const int16_t *y, size_t y_stride, int with_prefilter,
                            uint8_t *st, size_t s_stride, int16_t *proc, int16_t *jpeg, size_t plane_stride, int16_t *ll1, size_t ll1_stride,
                            int16_t *keep, size_t keep_stride, int n, hipStream_t s, int switches);
void nhw_launch_phase(int ph, const NhwWs &ws, int comp, uint8_t *out, uint32_t *sizes, int32_t *status, hipStream_t s);
void nhw_launch_front_stale(const int16_t *y, size_t y_stride, const uint8_t *st, size_t s_stride, int16_t *stale, size_t stale_stride, int n, hipStream_t s);
void nhw_launch_low_stale(const int16_t *km, size_t km_stride, int16_t *stale, size_t stale_stride, int n, hipStream_t s);
void nhw_launch_wave(int ph, const NhwWs &ws, hipStream_t s);
enum { WV_DQ1, WV_DQ0, WV_EMIT, WV_QUANT };
void nhw_launch_copy_block(const int16_t *src, size_t src_plane, int src_row, int16_t *dst, size_t dst_plane, int dst_row, int rows, int cols, int n, hipStream_t s);
enum { PH_L1, PH_L2, PH_L3, PH_L4A, PH_C0, PH_C2, PH_C3, PH_C4, PH_C5, PH_FINAL, PH_L4B, PH_L4C, PH_L4D, PH_LLC, PH_L4C2 };
/* quality 1..16 only (nhw_low.hip) */
int nhw_launch_low_prefilter(const int16_t *src, size_t src_stride, int16_t *y, size_t y_stride, int16_t *km, size_t km_stride, uint8_t *so, size_t so_stride, uint8_t *chain, size_t chain_stride,
                             uint16_t *tab, size_t tab_stride, int q, int n, hipStream_t s, int force = 0, int parts = 1, hipStream_t *aux = nullptr, hipEvent_t *ev = nullptr);
void nhw_launch_low_prefilter_chroma(const uint8_t *src, size_t src_stride, int16_t *dst, size_t dst_stride, int q, int n, hipStream_t s);
void nhw_launch_low_chroma_thin(int16_t *plane, size_t plane_stride, int n, hipStream_t s);
void nhw_launch_low_ll2(int16_t *proc, size_t plane_stride, int q, int n, hipStream_t s);

int nhw_front_set_attrs(const char **where);   /* nhw_front.hip, nhw_tail.hip: dynamic-LDS opt-ins of the device the handle lives on */
int nhw_tail_set_attrs(const char **where);
static thread_local std::string g_err;
extern "C" const char *nhw_last_error(void) { return g_err.c_str(); }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { char b_[256]; snprintf(b_, sizeof b_, "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); g_err = b_; return NHW_E_HIP; } } while (0)

struct nhw_enc {
	int device, max_batch;
	NhwWs ws;
	size_t slab_bytes;
	hipStream_t own_stream;
	hipStream_t part_stream[4];   /* a large batch runs as up to four sub-batches on streams of their own (see nhw_enc_batch_device) */
	hipEvent_t part_ev[5];
	hipStream_t low_stream[4];    /* quality 1..16: the pre-filter's sub-batches (nhw_launch_low_prefilter) */
	hipEvent_t low_ev[13];
	int low_parts;                /* how many (NHW_LOW_PARTS; 1 = the whole batch in line) */
	int low_parts_used;           /* ... in the batch that is being queued */
	int low_chroma;               /* quality 1..16: where the chroma sequence starts (NHW_LOW_CHROMA: 0 behind the front group, 1 behind the colour kernel, 2 behind the last sub-batch's pass A) */
	hipStream_t ll_stream;        /* the LL2 coder (Y16) beside the second dequantiser simulation */
	hipEvent_t ll_ev[2];
	int ll_fork;
	int quant_join;               /* the side streams join in front of the luma quantiser (q <= 21) */
	int parts;
	hipEvent_t ev[7];
	bool timed;
	int timed_parts, timed_front_images;
	/* host convenience path */
	uint8_t *d_in, *d_out, *d_compact;
	uint32_t *d_sizes; int32_t *d_status; uint64_t *d_offs;
	int conv_cap;
	int chroma_fork;  /* the chroma sequence on a stream of its own next to the luma tail (NHW_CHROMA_FORK=0 turns it off) */
	int lists_fork;   /* the position lists (Y24/Y25) on a third stream (NHW_LISTS_FORK=0 turns it off: +0.75 ms per q20 batch) */
	int front_fallback; /* debug: every row / segment of the pre-filter carry takes its exact fallback path (tests) */
	int stop_after;   /* debug: leave the batch driver after this many stages (0 = run everything) */
	int last_n, last_q; /* images and quality of the last whole batch (nhw_stage_chroma_l1 works on what it left in the 4:2:0 planes) */
};

static const size_t k_buf_bytes[B_COUNT] = {
	/* JPEG   */ 8 * Q, /* PROC */ 8 * Q, /* PU */ Q, /* PV */ Q, /* CJPEG */ 2 * Q, /* CPROC */ 2 * Q,
	/* LL1    */ 2 * Q, /* L2SAVE */ 2 * Q, /* CLL1 */ Q / 2, /* CL2SAVE */ Q / 2, /* KEEP */ 4 * Q, /* FIRST */ 2 * Q,
	/* BAND   */ 2 * Q, /* HS */ 4 * Q + 256, /* KMAP */ 8 * Q, /* ROWMAP (unused) */ 16, /* ROWSTATE */ 512, /* SCAN */ 6 * Q,
	/* LLBYTES*/ 24832, /* LLFULL */ 16384, /* EXW */ 16384 + 256, /* LLCOMP */ 32768, /* LLWORD */ 16384, /* LLMEM */ 32768,
	/* RES4   */ 8192, /* RAW */ 2 * Q + 1024, /* PAY */ 2 * Q + 256, /* CC */ 2 * Q + 1024, /* HALF */ 2 * Q + 1024, /* TMP16 */ Q / 2,
	/* R1     */ Q + 64, 8192 + 64, 16384 + 64, /* R3 */ Q + 64, 8192 + 64, 16384 + 64, /* R5 */ Q + 64, 8192 + 64, 16384 + 64,
	/* R6     */ 2 * Q + 1024, 16384 + 64, 16384 + 64, /* CHARRES */ 2048 + 64, /* QSET3 */ 8 * Q + 64,
	/* RESU64 */ 512, /* RESV64 */ 512, /* PACKET */ 320000, /* BOOK1 */ 768, /* BOOK2 */ 768, /* SEL1 */ 16384 + 64, /* SEL2 */ 16384 + 64,
	/* S1     */ 131072, /* S2 */ 131072, /* HIST */ 5632, /* META */ 256, /* PROF */ 512, /* ROWFLAG (unused) */ 16, /* SEGMAP (unused) */ 16, /* STALE */ (8 + 9 * 512) * 2,
	/* NZQ (32 x 128 words of 64 bits + 33 flush bases) */ Q / 2 + 256, /* NZS */ Q / 2, /* VOFF */ Q / 4, /* VALS (every symbol non-zero: 4 Q) */ 4 * Q,
	/* CNZQ (16 flushes x 64 lanes x 2 words of 64 bits + 17 flush bases) */ Q / 4 + 256, /* CVALS */ 2 * Q,
	/* CJPEG_V */ 2 * Q, /* CPROC_V */ 2 * Q, /* CLL1_V */ Q / 2, /* CL2SAVE_V */ Q / 2, /* UBYTES */ Q,
	/* LOWTAB (quality 1..16: pass A's five candidate masks, 64 bytes each, for every row) */ 320 * 512
};

static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int nhw_quality_supported(int quality) { return quality >= 1 && quality <= 23; }

extern "C" int nhw_enc_set_compat(nhw_enc *e, int mode)
{
	if (!e || (mode != NHW_COMPAT_CANONICAL && mode != NHW_COMPAT_GLIBC_ONESHOT)) return NHW_E_ARG;
	if (e->ws.compat != mode && mode == NHW_COMPAT_CANONICAL) {      /* the compatibility mode writes behind ll1 and the level-2 copy: give the guards their zeros back */
		HIPCHK(hipSetDevice(e->device));
		HIPCHK(hipDeviceSynchronize());
		HIPCHK(hipMemset2D(e->ws.base + e->ws.off[B_LL1] + 2 * Q, e->ws.stride[B_LL1], 0, 1024, (size_t)e->max_batch));
		HIPCHK(hipMemset2D(e->ws.base + e->ws.off[B_L2SAVE] + 2 * Q, e->ws.stride[B_L2SAVE], 0, 256, (size_t)e->max_batch));
	}
	e->ws.compat = mode;
	return NHW_OK;
}

extern "C" void nhw_enc_destroy(nhw_enc *e);
static int host_buffers(nhw_enc *e, int n);
/* device bytes per image of the host path's staging (nhw_enc_batch / nhw_enc_synth_batch): input slot, output slot, compacted output */
#define HOST_PATH_BYTES ((size_t)NHW_IMG_BYTES + 2 * (size_t)NHW_OUT_STRIDE + 24)
extern "C" int nhw_enc_create_ex(int device, int max_batch, unsigned flags, nhw_enc **out)
{
	if (!out || max_batch < 1 || max_batch > 65535 || (flags & ~(unsigned)NHW_CREATE_DEVICE_ONLY)) { g_err = "bad argument"; return NHW_E_ARG; }
	const bool host_staging = !(flags & NHW_CREATE_DEVICE_ONLY);
	HIPCHK(hipSetDevice(device));
	nhw_enc *e = new nhw_enc();
	memset(e, 0, sizeof *e);
	e->device = device; e->max_batch = max_batch;
	size_t total = 0;
	for (int b = 0; b < B_COUNT; b++) {
		e->ws.stride[b] = round_up(k_buf_bytes[b] + GUARD, 256);
		e->ws.off[b] = total + GUARD;
		total += GUARD + e->ws.stride[b] * (size_t)max_batch;
	}
	e->slab_bytes = total;
	const int rc = [&]() -> int {                                  /* a failure half-way leaves nothing behind: the handle is destroyed below */
		size_t free_b = 0, total_b = 0;
		HIPCHK(hipMemGetInfo(&free_b, &total_b));
		const size_t need = total + (host_staging ? HOST_PATH_BYTES * (size_t)max_batch : 0);
		if (need > free_b) {                                       /* 7.0 MB of workspace (+ 1.8 MB of host-path staging) per image: say so instead of failing inside hipMalloc */
			char b[200];
			snprintf(b, sizeof b, "encoder workspace for max_batch %d needs %zu MiB (%.1f MiB per image), %zu MiB of HBM are free", max_batch, need >> 20, (double)need / max_batch / 1048576.0, free_b >> 20);
			g_err = b;
			return NHW_E_ARG;
		}
		{ const char *where = ""; int rc_ = nhw_front_set_attrs(&where); if (!rc_) rc_ = nhw_tail_set_attrs(&where);   /* before anything is launched on this device */
		  if (rc_) { g_err = std::string(where) + " -> " + hipGetErrorString((hipError_t)rc_); return NHW_E_HIP; } }
		HIPCHK(hipMalloc((void **)&e->ws.base, total));
		HIPCHK(hipMemset(e->ws.base, 0, total));       /* guards must be zero; they are never written afterwards */
		HIPCHK(hipStreamCreate(&e->own_stream));
		for (int i = 0; i < 7; i++) HIPCHK(hipEventCreate(&e->ev[i]));
		for (int i = 0; i < 4; i++) HIPCHK(hipStreamCreateWithFlags(&e->part_stream[i], hipStreamNonBlocking));
		for (int i = 0; i < 5; i++) HIPCHK(hipEventCreateWithFlags(&e->part_ev[i], hipEventDisableTiming));
		for (int i = 0; i < 4; i++) HIPCHK(hipStreamCreateWithFlags(&e->low_stream[i], hipStreamNonBlocking));
		for (int i = 0; i < 13; i++) HIPCHK(hipEventCreateWithFlags(&e->low_ev[i], hipEventDisableTiming));
		{ const char *lp = getenv("NHW_LOW_PARTS"); e->low_parts = lp ? atoi(lp) : 2; if (e->low_parts < 1 || e->low_parts > 4) e->low_parts = 1; }
		{ const char *lc = getenv("NHW_LOW_CHROMA"); e->low_chroma = lc ? atoi(lc) : 2; }
		HIPCHK(hipStreamCreateWithFlags(&e->ll_stream, hipStreamNonBlocking));
		for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&e->ll_ev[i], hipEventDisableTiming));
		/* the host path's staging buffers, for the whole of max_batch, now: allocated on the first nhw_enc_batch they made that call twice as
		 * slow as the ones behind it (gigabytes of hipMalloc inside the timed region of whoever measured it).  A caller that only ever hands over
		 * device buffers says NHW_CREATE_DEVICE_ONLY and does not pay for them; should it call the host path after all, that call allocates. */
		return host_staging ? host_buffers(e, max_batch) : NHW_OK;
	}();
	if (rc != NHW_OK) { nhw_enc_destroy(e); return rc; }
	e->parts = 1;   /* sub-batches on streams of their own (NHW_PARTS=2..4) bought 4 % while the tail kernels were latency-bound; they no longer do */
	e->chroma_fork = 1;
	if (const char *p = getenv("NHW_CHROMA_FORK")) e->chroma_fork = atoi(p) != 0;
	e->lists_fork = 1;
	e->ll_fork = 1;
	e->quant_join = 1;
	if (const char *p = getenv("NHW_QUANT_JOIN")) e->quant_join = atoi(p) != 0;
	if (const char *p = getenv("NHW_LL_FORK")) e->ll_fork = atoi(p) != 0;
	if (const char *p = getenv("NHW_LISTS_FORK")) e->lists_fork = atoi(p) != 0;
	if (const char *p = getenv("NHW_PARTS")) { const int k = atoi(p); if (k >= 1 && k <= 4) e->parts = k; }
	*out = e;
	return NHW_OK;
}

extern "C" int nhw_enc_create(int device, int max_batch, nhw_enc **out) { return nhw_enc_create_ex(device, max_batch, 0u, out); }

extern "C" void nhw_enc_destroy(nhw_enc *e)
{
	if (!e) return;
	(void)hipSetDevice(e->device);
	(void)hipDeviceSynchronize();
	if (e->ws.base) (void)hipFree(e->ws.base);
	if (e->d_in) (void)hipFree(e->d_in);
	if (e->d_out) (void)hipFree(e->d_out);
	if (e->d_compact) (void)hipFree(e->d_compact);
	if (e->d_sizes) (void)hipFree(e->d_sizes);
	if (e->d_status) (void)hipFree(e->d_status);
	if (e->d_offs) (void)hipFree(e->d_offs);
	for (int i = 0; i < 7; i++) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
	if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
	for (int i = 0; i < 4; i++) if (e->part_stream[i]) (void)hipStreamDestroy(e->part_stream[i]);
	for (int i = 0; i < 5; i++) if (e->part_ev[i]) (void)hipEventDestroy(e->part_ev[i]);
	for (int i = 0; i < 4; i++) if (e->low_stream[i]) (void)hipStreamDestroy(e->low_stream[i]);
	for (int i = 0; i < 13; i++) if (e->low_ev[i]) (void)hipEventDestroy(e->low_ev[i]);
	if (e->ll_stream) (void)hipStreamDestroy(e->ll_stream);
	for (int i = 0; i < 2; i++) if (e->ll_ev[i]) (void)hipEventDestroy(e->ll_ev[i]);
	delete e;
}

static inline int16_t *plane16(const NhwWs &ws, int b) { return (int16_t *)(ws.base + ws.off[b]); }
static inline uint8_t *plane8(const NhwWs &ws, int b) { return ws.base + ws.off[b]; }

/* the whole launch sequence for the images of one workspace view on one stream; `timed`: record the stage events of nhw_timing */
static int run_batch(nhw_enc *e, const NhwWs &ws_in, const void *d_bgr, int n, int quality, void *d_out, uint32_t *d_sizes, int32_t *d_status, hipStream_t s,
                     int timed /* 0: no events, 1: ev[0..4] (whole batch), 2: ev[2], ev[3] (tail of the first sub-batch; the caller closes with ev[4]) */,
                     int what = 3 /* bit 0: the front launch group (colour, pre-filter, level-1 analysis), bit 1: everything behind it */)
{
	NhwWs ws = ws_in;
	const int q = quality;
	const bool low = q <= 16;      /* integer colour, the rationed pre-filter of image_processing.c:838-2423 and the other quality 1..16 forms (nhw_low.hip) */
	int16_t *jpeg = plane16(ws, B_JPEG), *proc = plane16(ws, B_PROC);
	int16_t *cjpeg = plane16(ws, B_CJPEG), *cproc = plane16(ws, B_CPROC);
	const size_t ps = ws.stride[B_JPEG] / 2, cps = ws.stride[B_CJPEG] / 2;
	uint8_t *out = (uint8_t *)d_out;

	int stage = 0;
#define STAGE_DONE() do { if (e->stop_after && ++stage == e->stop_after) { HIPCHK(hipGetLastError()); return NHW_OK; } } while (0)   
	(void)n;
	if (what & 1) {
	if (timed == 1) HIPCHK(hipEventRecord(e->ev[0], s));
	/* a1 + a2 + Y2 + Y3: colour + 4:2:0, pre-filter (q<=21, nhw_encoder.c:116-119), level-1 analysis (:125), LL1 copy (:127-135): ONE kernel
	 * for quality 17..23 (k_front_image with the pre-filter, k_front_plain without: a workgroup walks an image top to bottom).  The luma plane
	 * never reaches HBM.  Quality 1..16: colour kernel -> luma plane, the rationed pre-filter (nhw_low.hip) -> k_front_plain's input plane. */
	int16_t *yin = plane16(ws, B_KMAP);
	const size_t yin_stride = ws.stride[B_KMAP];
	if (low) {
		nhw_launch_color((const uint8_t *)d_bgr, n, q, jpeg, ws.stride[B_JPEG], plane8(ws, B_PU), plane8(ws, B_PV), ws.stride[B_PU], s);
		HIPCHK(hipEventRecord(e->ev[5], s));                      /* with the front group, whoever brackets it: nhw_timing.color_dwt_ms / prefilter_ms */
		STAGE_DONE();
		{ const int lparts = (timed == 1 && what == 3 && !e->stop_after && n >= 1024) ? e->low_parts : 1;   /* (the stage checks and small batches: in line) */
		  e->low_parts_used = lparts;
		  HIPCHK((hipError_t)nhw_launch_low_prefilter(jpeg, ws.stride[B_JPEG] / 2, yin, yin_stride / 2, proc, ps, plane8(ws, B_SCAN), ws.stride[B_SCAN], plane8(ws, B_KEEP), ws.stride[B_KEEP], (uint16_t *)plane8(ws, B_LOWTAB), ws.stride[B_LOWTAB], q, n, s, (e->front_fallback & 1) ? 32 : 0,
		                                              lparts, e->low_stream, e->low_ev)); }   /* contrast map -> proc plane, flags -> scan buffer: both free until the band kernel / the quantiser; the pair machine's answers -> the q >= 22 plane */
		HIPCHK(hipEventRecord(e->ev[6], s));
		STAGE_DONE();
		if (ws.compat) nhw_launch_low_stale(proc, ps, plane16(ws, B_STALE), ws.stride[B_STALE], n, s);   /* compatibility mode only: the map cells the stock binary's heap re-uses */
		nhw_launch_front_fused(nullptr, q, nullptr, nullptr, 0, yin, yin_stride, 0, plane8(ws, B_ROWSTATE), ws.stride[B_ROWSTATE],
		                       proc, jpeg, ps, plane16(ws, B_LL1), ws.stride[B_LL1] / 2, nullptr, 0, n, s, ws.dbg ? 2 : 0);
	} else {
		HIPCHK(hipEventRecord(e->ev[5], s)); HIPCHK(hipEventRecord(e->ev[6], s));   /* no kernels of their own for colour and pre-filter: both times 0 */
		STAGE_DONE();
		if (q < 22) STAGE_DONE();
		nhw_launch_front_fused((const uint8_t *)d_bgr, q, plane8(ws, B_PU), plane8(ws, B_PV), ws.stride[B_PU], yin, yin_stride /* developer builds only: a plane for a dump */, q < 22,
		                       plane8(ws, B_ROWSTATE), ws.stride[B_ROWSTATE], proc, jpeg, ps, plane16(ws, B_LL1), ws.stride[B_LL1] / 2,
		                       q > 21 ? plane16(ws, B_KEEP) : nullptr, ws.stride[B_KEEP] / 2, n, s, (e->front_fallback & 1) | (ws.dbg ? 2 : 0));
		if (ws.compat && q < 22) {   /* compatibility mode only: the kernel-map cells the stock binary's heap re-uses are replayed from a luma plane */
			nhw_launch_color((const uint8_t *)d_bgr, n, q, yin, yin_stride, plane8(ws, B_PU), plane8(ws, B_PV), ws.stride[B_PU], s);
			nhw_launch_front_stale(yin, yin_stride, plane8(ws, B_ROWSTATE), ws.stride[B_ROWSTATE], plane16(ws, B_STALE), ws.stride[B_STALE], n, s);
		}
	}
	STAGE_DONE();
	STAGE_DONE();
	if (timed == 1) HIPCHK(hipEventRecord(e->ev[1], s));
	if (!(what & 2)) { HIPCHK(hipGetLastError()); return NHW_OK; }
	}
	/* The chroma sequence needs nothing of the luma tail except the length of the exception list (its own entries go behind the
	 * luma ones, Y15): it runs on a stream of its own next to the luma tail and fills the issue slots the latency-bound luma kernels
	 * leave.  (Since round 5 V works in planes of its own and U's symbols are parked in B_UBYTES until V's quantiser merges them: the
	 * sequence no longer waits for the band plane Y29 is done with, nor V's head for U's quantiser.) */
	const bool fork = timed == 1 && what == 3 && !e->stop_after && e->chroma_fork;
	const bool fork_ll = fork && q > 13 && !ws.compat && e->ll_fork;
	ws.defer_verbatim = fork_ll;
	hipStream_t cs = fork ? e->part_stream[0] : s;
	ws.split_chroma = fork;                                          /* (the stage checks and the in-line order keep the reference's one set of planes) */
	auto chroma_head = [&](int comp) -> int {                        /* everything up to the second dequantiser simulation */
		const bool vp = comp && ws.split_chroma;
		int16_t *cjpeg = plane16(ws, vp ? B_CJPEG_V : B_CJPEG), *cproc = plane16(ws, vp ? B_CPROC_V : B_CPROC);
		int16_t *cll1 = plane16(ws, vp ? B_CLL1_V : B_CLL1), *cl2save = plane16(ws, vp ? B_CL2SAVE_V : B_CL2SAVE);
		const bool widen_in_analysis = q > 14 && !ws.dbg;              /* the analysis reads the byte plane itself (the stage checks keep the copy as a stage of its own) */
		if (q <= 14) nhw_launch_low_prefilter_chroma(comp ? plane8(ws, B_PV) : plane8(ws, B_PU), ws.stride[B_PU], cjpeg, cps, q, n, cs);   /* :2263 / :2579 */
		else if (!widen_in_analysis) nhw_launch_phase(PH_C0, ws, comp, out, d_sizes, d_status, cs);
		STAGE_DONE();
		nhw_launch_analysis(cjpeg, cproc, n, cps, H, H, 0, cs, cll1, ws.stride[B_CLL1] / 2, H / 2, 2,   /* + the copy of LL1 */
		                    widen_in_analysis ? (comp ? plane8(ws, B_PV) : plane8(ws, B_PU)) : nullptr, ws.stride[B_PU], ws.dbg ? 0 : 2);   /* 2: nor the LL quadrant back into the work plane -- the level-2 analysis below reads its copy */
		if (low) nhw_launch_low_chroma_thin(cproc, cps, n, cs);      /* :2277-2308 / :2590-2621 */
		STAGE_DONE();
		STAGE_DONE();
		if (ws.dbg) nhw_launch_analysis(cjpeg, cproc, n, cps, H, H / 2, 1, cs, nullptr, 0, 0, 0, nullptr, 0, 0);
		else nhw_launch_analysis(cjpeg, cproc, n, cps, H, H / 2, 1, cs, nullptr, 0, 0, 0, nullptr, 0, 1, cll1, ws.stride[B_CLL1] / 2, H / 2);   /* from the copy of LL1 */
		STAGE_DONE();
		nhw_launch_phase(PH_C2, ws, comp, out, d_sizes, d_status, cs);
		STAGE_DONE();
		nhw_launch_synthesis(cjpeg, cproc, n, cps, H, H / 2, cs, !ws.dbg);
		STAGE_DONE();
		nhw_launch_phase(PH_C3, ws, comp, out, d_sizes, d_status, cs);
		STAGE_DONE();
		nhw_launch_analysis(cjpeg, cproc, n, cps, H, H / 2, 1, cs, cl2save, ws.stride[B_CL2SAVE] / 2, H / 2, 1, nullptr, 0, !ws.dbg);   /* + the copy of the level-2 block */
		STAGE_DONE();
		STAGE_DONE();
		nhw_launch_phase(PH_C4, ws, comp, out, d_sizes, d_status, cs);
		STAGE_DONE();
		nhw_launch_synthesis(cjpeg, cproc, n, cps, H, H / 2, cs, !ws.dbg);
		STAGE_DONE();
		return 1;
	};
	auto chroma_tail = [&](int comp) -> int {                        /* marks, LL2 emission (appends to the exception list), quantiser, stream bytes */
		nhw_launch_phase(PH_C5, ws, comp, out, d_sizes, d_status, cs);
		STAGE_DONE();
		return 1;
	};
#define CHROMA(call) do { const int rc_ = (call); if (rc_ != 1) return rc_; } while (0)   /* 1 = carry on; NHW_OK (debug stop) or an error leaves */
	if (fork) {
		HIPCHK(hipStreamWaitEvent(cs, !low || e->low_chroma == 0 ? e->ev[1] : (e->low_chroma == 2 && e->low_parts_used > 1) ? e->low_ev[e->low_parts_used] : e->ev[5], 0));   /* behind the front launch group: that one is bound by vector issue and has nothing to give (and its time is the roofline figure).  Quality 1..16: behind the colour kernel already -- the rationed pre-filter's chain (k_low_chain) is one wavefront a picture on the scalar unit and leaves the vector units and the memory system idle for milliseconds */
		CHROMA(chroma_head(0));
		CHROMA(chroma_head(1));                                      /* V's head in planes of its own, right behind U's: U's quantiser waits for the luma tail, and this stream stood idle until then (2 ms of a q20 step).  (Measured and not taken: V's head on a stream of its own beside U's, +0.3 ms; V's head held back until the second dequantiser simulation is through, +0.4 ms.) */
	}
	/* Y4: level-2 analysis (:139) */
	/* the LL rows come from ll1 (the front's copy of them in natural orientation, res256): the front does not write them into the work plane as well
	 * outside the stage checks, and this analysis fills that quadrant of the work plane itself (its transposed first-direction plane) */
	nhw_launch_analysis(jpeg, proc, n, ps, W, H, 1, s, nullptr, 0, 0, 0, nullptr, 0, 0, plane16(ws, B_LL1), ws.stride[B_LL1] / 2, H);
	STAGE_DONE();
	if (q > 6) {                                                     /* first closed loop (:141-283) */
	nhw_launch_phase(PH_L1, ws, 0, out, d_sizes, d_status, s);
	nhw_launch_wave(WV_DQ1, ws, s);          /* every quality (1..16: rationed low bits, no marking passes) */
	STAGE_DONE();
	if (ws.dbg) {
	nhw_launch_synthesis(jpeg, proc, n, ps, W, H, s);
	STAGE_DONE();
	nhw_launch_phase(PH_L2, ws, 0, out, d_sizes, d_status, s);
	STAGE_DONE();
	} else nhw_launch_l2_recon(jpeg, proc, ps, plane16(ws, B_LL1), ws.stride[B_LL1] / 2, n, s);   /* synthesis + Y8 + Y9 on one residency of the block; the stage checks take the three kernels */
	if (q > 12) nhw_launch_analysis(jpeg, proc, n, ps, W, H, 1, s, plane16(ws, B_L2SAVE), ws.stride[B_L2SAVE] / 2, H, 1);   /* + Y13 (:623-631): copy of the coefficient block */
	else nhw_launch_analysis(jpeg, proc, n, ps, W, H, 1, s);
	STAGE_DONE();
	}
	if (q <= 12) {                                                   /* Y11 (q <= 11), Y12, then Y13 */
		nhw_launch_low_ll2(proc, ps, q, n, s);
		nhw_launch_copy_block(proc, ps, W, plane16(ws, B_L2SAVE), ws.stride[B_L2SAVE] / 2, H, H, H, n, s);
	}
	STAGE_DONE();
	nhw_launch_wave(WV_EMIT, ws, s);                                 /* Y14, Y15 */
	/* Y16, the LL2 coder, is a latency-bound parse (0.6 ms at 0.3 TB/s) in front of the vector-bound dequantiser simulation, which only wants its
	 * list of verbatim samples -- at its very end, to put them back into the block.  Production: the coder runs beside the simulation on a
	 * stream of its own and the synthesis behind both does the putting back (ws.defer_verbatim).  Not in the compatibility mode and not
	 * below q14, where the coder's launch also lays out heap neighbours that the passes behind it read (luma_p3_par). */
	if (fork_ll) {
		HIPCHK(hipEventRecord(e->ll_ev[0], s));
		HIPCHK(hipStreamWaitEvent(e->ll_stream, e->ll_ev[0], 0));
		nhw_launch_phase(PH_L3, ws, 0, out, d_sizes, d_status, e->ll_stream);
		HIPCHK(hipEventRecord(e->ll_ev[1], e->ll_stream));
		HIPCHK(hipEventRecord(e->part_ev[0], e->ll_stream));   /* exception list of the luma plane complete, and the coder through with the bytes behind the luma samples: the chroma emission writes its own there */
	} else {
	nhw_launch_phase(PH_L3, ws, 0, out, d_sizes, d_status, s);
	if (fork) HIPCHK(hipEventRecord(e->part_ev[0], s));              /* exception list of the luma plane complete */
	}
	if (q > 12) {                                                    /* second closed loop (:759-779) */
	nhw_launch_wave(WV_DQ0, ws, s);
	STAGE_DONE();
	if (fork_ll) {
		HIPCHK(hipStreamWaitEvent(s, e->ll_ev[1], 0));               /* (the coder is long done: the simulation takes twice its time) */
		nhw_launch_synthesis(jpeg, proc, n, ps, W, H, s, q <= 21 && !ws.dbg, ws.buf<uint16_t>(B_LLMEM, 0), ws.stride[B_LLMEM], &ws.buf<NhwMeta>(B_META, 0)->ll_mem_len, ws.stride[B_META]);
	} else
	nhw_launch_synthesis(jpeg, proc, n, ps, W, H, s, q <= 21 && !ws.dbg);   /* its copy in natural orientation is only read by Y19 (q > 21, :766-777) */
	STAGE_DONE();
	}
	nhw_launch_phase(PH_L4A, ws, 0, out, d_sizes, d_status, s);      /* Y19-Y23 */
	/* Y24, Y25: the position lists are read by nothing before the packetiser, and what the pass leaves in the residual-code plane by nobody
	 * at all; below q21 it shares no scratch with the passes behind it either (from q21 on its third list and Y27's snapshot both live in
	 * the hs plane, and Y29 needs Y24), so there it runs beside them on a stream of its own */
	const bool fork_lists = fork && q <= 20 && q > 12 && e->lists_fork;
	if (fork_lists) {
		HIPCHK(hipEventRecord(e->part_ev[2], s));
		HIPCHK(hipStreamWaitEvent(e->part_stream[1], e->part_ev[2], 0));
		nhw_launch_phase(PH_L4B, ws, 0, out, d_sizes, d_status, e->part_stream[1]);
		HIPCHK(hipEventRecord(e->part_ev[3], e->part_stream[1]));
	} else if (q > 12)
		nhw_launch_phase(PH_L4B, ws, 0, out, d_sizes, d_status, s);  /* Y24, Y25 (:1498) */
	nhw_launch_phase(PH_L4C, ws, 0, out, d_sizes, d_status, s);      /* Y26, Y27 */
	const bool early_join = fork && e->quant_join;
	auto chroma_rest = [&]() -> int {
		HIPCHK(hipStreamWaitEvent(cs, e->part_ev[0], 0));
		CHROMA(chroma_tail(0));
		CHROMA(chroma_tail(1));
		nhw_launch_phase(PH_LLC, ws, 0, out, d_sizes, d_status, cs);   /* Z1: the chroma LL2 coder appends to the luma one's output (Y16, long done) */
		HIPCHK(hipEventRecord(e->part_ev[1], cs));
		return 1;
	};
	/* The quantiser is a wavefront an image at 115 registers: four wavefronts fill a SIMD's register file, and it is as fast as its slowest
	 * wavefront is late.  A side stream's workgroup that sits on a CU when it starts keeps four of its images waiting for a second round (the
	 * kernel took 3.0 ms beside the chroma sequence, 1.9 alone).  So the side streams are let finish first (they have had the
	 * multi-round kernels Y19-Y27 to hide behind), and Y31 and the packetiser then run alone as well. */
	if (early_join) {
		CHROMA(chroma_rest());
		if (fork_lists) HIPCHK(hipStreamWaitEvent(s, e->part_ev[3], 0));
		HIPCHK(hipStreamWaitEvent(s, e->part_ev[1], 0));
	}
	nhw_launch_wave(WV_QUANT, ws, s);                                /* Y28 (+ Y30: the symbols leave in stream order), every quality */
	if (q > 21) nhw_launch_phase(PH_L4C2, ws, 0, out, d_sizes, d_status, s);   /* Y29 */
	if (fork && !early_join) CHROMA(chroma_rest());                  /* queued here so that the wait finds its event recorded */
	nhw_launch_phase(PH_L4D, ws, 0, out, d_sizes, d_status, s);      /* Y31 (Y30, the stream order, is the quantisers' output order) */
	STAGE_DONE();
	if (timed) HIPCHK(hipEventRecord(e->ev[2], s));

	if (fork_lists && !early_join) HIPCHK(hipStreamWaitEvent(s, e->part_ev[3], 0));
	if (fork) { if (!early_join) HIPCHK(hipStreamWaitEvent(s, e->part_ev[1], 0)); }
	else
		for (int comp = 0; comp < 2; comp++) {       /* U then V (:2255-2570, :2572-2868) */
			CHROMA(chroma_head(comp));
			CHROMA(chroma_tail(comp));
		}
#undef CHROMA
	if (timed) HIPCHK(hipEventRecord(e->ev[3], s));
	if (!fork) nhw_launch_phase(PH_LLC, ws, 0, out, d_sizes, d_status, s);     /* Z1 */
	nhw_launch_phase(PH_FINAL, ws, 0, out, d_sizes, d_status, s);   /* Z2, container */
	if (timed == 1) HIPCHK(hipEventRecord(e->ev[4], s));
	HIPCHK(hipGetLastError());
	return NHW_OK;
}

/* Most kernels of the sequence are bound by latency at the occupancy their LDS footprint allows, not by HBM or the ALUs, so a
 * large batch is cut into sub-batches whose sequences run on streams of their own: kernels of different stages overlap on the
 * CUs.  Images are independent and the workspace is indexed per image, so a sub-batch is just a shifted view of it. */
extern "C" int nhw_enc_batch_device(nhw_enc *e, const void *d_bgr, int n, int quality, void *d_out, uint32_t *d_sizes,
                                    int32_t *d_status, void *stream)
{
	if (!e || !d_bgr || !d_out || !d_sizes || !d_status || n < 1 || n > e->max_batch) { g_err = "bad argument"; return NHW_E_ARG; }
	if (!nhw_quality_supported(quality)) { g_err = "quality outside 1..23"; return NHW_E_QUALITY; }
	HIPCHK(hipSetDevice(e->device));
	hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
	NhwWs ws = e->ws;
	ws.n = n; ws.q = quality; ws.dbg = e->stop_after != 0;
	e->timed = false;                                              /* set again only when a whole, un-stopped batch has recorded every event of nhw_timing */
	const int parts = (e->stop_after || n < 512) ? 1 : e->parts;
	if (parts == 1) {
		const int rc = run_batch(e, ws, d_bgr, n, quality, d_out, d_sizes, d_status, s, 1);
		if (rc == NHW_OK && !e->stop_after) { e->timed = true; e->timed_parts = 1; e->timed_front_images = n; e->last_n = n; e->last_q = quality; }
		return rc;
	}
	/* the front launch group is the part that is bound by the memory system and the ALUs: it runs once for the whole batch */
	HIPCHK(hipEventRecord(e->ev[0], s));
	{ const int rc = run_batch(e, ws, d_bgr, n, quality, d_out, d_sizes, d_status, s, 0, 1); if (rc != NHW_OK) return rc; }
	HIPCHK(hipEventRecord(e->ev[1], s));
	HIPCHK(hipEventRecord(e->part_ev[4], s));
	e->timed_front_images = n;
	for (int k = 0; k < parts; k++) {
		const int i0 = (int)((long long)n * k / parts), i1 = (int)((long long)n * (k + 1) / parts);
		NhwWs view = ws;
		view.n = i1 - i0;
		for (int b = 0; b < B_COUNT; b++) view.off[b] += (size_t)i0 * ws.stride[b];
		hipStream_t ps_ = e->part_stream[k];
		HIPCHK(hipStreamWaitEvent(ps_, e->part_ev[4], 0));
		const int rc = run_batch(e, view, (const uint8_t *)d_bgr + (size_t)i0 * (W * W * 3), i1 - i0, quality, (uint8_t *)d_out + (size_t)i0 * NHW_OUT_STRIDE,
		                         d_sizes + i0, d_status + i0, ps_, k == 0 ? 2 : 0, 2);
		if (rc != NHW_OK) return rc;
		HIPCHK(hipEventRecord(e->part_ev[k], ps_));
		HIPCHK(hipStreamWaitEvent(s, e->part_ev[k], 0));
	}
	HIPCHK(hipEventRecord(e->ev[4], s));
	e->timed = true; e->timed_parts = parts; e->last_n = n; e->last_q = quality;
	return NHW_OK;
}

extern "C" int nhw_enc_last_timing(nhw_enc *e, nhw_timing *t)
{
	if (!e || !t || !e->timed) { g_err = "no timed batch"; return NHW_E_ARG; }
	HIPCHK(hipEventSynchronize(e->ev[4]));
	memset(t, 0, sizeof *t);
	HIPCHK(hipEventElapsedTime(&t->total_ms, e->ev[0], e->ev[4]));
	HIPCHK(hipEventElapsedTime(&t->front_ms, e->ev[0], e->ev[1]));   /* several parts: the later stage times are those of the first sub-batch, on its stream */
	HIPCHK(hipEventElapsedTime(&t->luma_ms, e->ev[1], e->ev[2]));
	HIPCHK(hipEventElapsedTime(&t->chroma_ms, e->ev[2], e->ev[3]));
	HIPCHK(hipEventElapsedTime(&t->entropy_ms, e->ev[3], e->ev[4]));
	HIPCHK(hipEventElapsedTime(&t->color_dwt_ms, e->ev[0], e->ev[5]));
	HIPCHK(hipEventElapsedTime(&t->prefilter_ms, e->ev[5], e->ev[6]));
	t->parts = e->timed_parts; t->front_images = e->timed_front_images;
	return NHW_OK;
}

extern "C" int nhw_synth_batch_device(nhw_enc *e, void *d_bgr, int n, uint32_t seed_base, void *stream)
{
	if (!e || !d_bgr || n < 1) { g_err = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(e->device));
	nhw_launch_synth((uint8_t *)d_bgr, n, seed_base, stream ? (hipStream_t)stream : e->own_stream);
	HIPCHK(hipGetLastError());
	return NHW_OK;
}

/* ------------------------------------------------------------------------------------------------ host path */
__global__ void k_offsets(const uint32_t *sizes, uint64_t *offs, int n)
{
	if (blockIdx.x || threadIdx.x) return;
	uint64_t acc = 0;
	for (int i = 0; i < n; i++) { offs[i] = acc; acc += sizes[i]; }
	offs[n] = acc;
}
__global__ __launch_bounds__(256) void k_compact(const uint8_t *out, const uint32_t *sizes, const uint64_t *offs, uint8_t *dst)
{
	const int img = blockIdx.x;
	const uint8_t *s = out + (size_t)img * NHW_OUT_STRIDE;
	uint8_t *d = dst + offs[img];
	for (uint32_t i = threadIdx.x; i < sizes[img]; i += 256) d[i] = s[i];
}

/* buffers of the host path for up to n images; on a failed allocation nothing dangles and the capacity stays what really exists */
static int host_buffers(nhw_enc *e, int n)
{
	if (e->conv_cap >= n) return NHW_OK;
	void **ptrs[6] = { (void **)&e->d_in, (void **)&e->d_out, (void **)&e->d_compact, (void **)&e->d_sizes, (void **)&e->d_status, (void **)&e->d_offs };
	for (auto pp : ptrs) { if (*pp) (void)hipFree(*pp); *pp = nullptr; }
	e->conv_cap = 0;
	HIPCHK(hipMalloc((void **)&e->d_in, (size_t)n * NHW_IMG_BYTES));
	HIPCHK(hipMalloc((void **)&e->d_out, (size_t)n * NHW_OUT_STRIDE));
	HIPCHK(hipMalloc((void **)&e->d_compact, (size_t)n * NHW_OUT_STRIDE));
	HIPCHK(hipMalloc((void **)&e->d_sizes, sizeof(uint32_t) * n));
	HIPCHK(hipMalloc((void **)&e->d_status, sizeof(int32_t) * n));
	HIPCHK(hipMalloc((void **)&e->d_offs, sizeof(uint64_t) * (n + 1)));
	e->conv_cap = n;
	return NHW_OK;
}

/* compact the per-image output slots and bring them to the host */
static int host_download(nhw_enc *e, int n, uint8_t *out_arena, size_t arena_cap, uint64_t *out_off, int32_t *status)
{
	hipStream_t s = e->own_stream;
	k_offsets<<<1, 1, 0, s>>>(e->d_sizes, e->d_offs, n);
	k_compact<<<n, 256, 0, s>>>(e->d_out, e->d_sizes, e->d_offs, e->d_compact);
	HIPCHK(hipMemcpyAsync(out_off, e->d_offs, sizeof(uint64_t) * (n + 1), hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(status, e->d_status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	if (out_off[n] > arena_cap) { g_err = "output arena too small"; return NHW_E_SPACE; }
	HIPCHK(hipMemcpy(out_arena, e->d_compact, out_off[n], hipMemcpyDeviceToHost));
	return NHW_OK;
}

/* page-locked host memory: buffers handed to nhw_enc_batch that come from here travel by DMA at PCIe speed while the previous chunk
 * is being encoded (pageable memory is staged by the runtime and moves at about half of that) */
extern "C" void *nhw_host_alloc(size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
extern "C" void nhw_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int nhw_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

extern "C" int nhw_enc_batch(nhw_enc *e, const uint8_t *bgr, int n, int quality, uint8_t *out_arena, size_t arena_cap,
                             uint64_t *out_off, int32_t *status)
{
	if (!e || !bgr || !out_arena || !out_off || !status || n < 1 || n > e->max_batch) { g_err = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(e->device));
	{ const int rc = host_buffers(e, n); if (rc) return rc; }
	hipStream_t s = e->own_stream, cs = e->part_stream[3];
	/* chunks of 1024 images: the upload of a chunk (its own stream) overlaps the encode of the one before; every chunk is
	 * encoded in the first workspace slots, one after the other on `s` */
	const int chunk = 1024;
	for (int i0 = 0; i0 < n; i0 += chunk) {
		const int m = n - i0 < chunk ? n - i0 : chunk;
		hipEvent_t up;
		HIPCHK(hipEventCreateWithFlags(&up, hipEventDisableTiming));
		HIPCHK(hipMemcpyAsync(e->d_in + (size_t)i0 * NHW_IMG_BYTES, bgr + (size_t)i0 * NHW_IMG_BYTES, (size_t)m * NHW_IMG_BYTES, hipMemcpyHostToDevice, cs));
		HIPCHK(hipEventRecord(up, cs));
		HIPCHK(hipStreamWaitEvent(s, up, 0));
		HIPCHK(hipEventDestroy(up));
		const int rc = nhw_enc_batch_device(e, e->d_in + (size_t)i0 * NHW_IMG_BYTES, m, quality, e->d_out + (size_t)i0 * NHW_OUT_STRIDE, e->d_sizes + i0, e->d_status + i0, s);
		if (rc) return rc;
	}
	return host_download(e, n, out_arena, arena_cap, out_off, status);
}

/* SURVEY.md 8(d) synthetic images seed_base .. seed_base+n-1, generated on the device, encoded, and the files brought to the host:
 * `nhw-enc --synthetic` (tools/nhw_enc.c) */
extern "C" int nhw_enc_synth_batch(nhw_enc *e, int n, uint32_t seed_base, int quality, uint8_t *out_arena, size_t arena_cap, uint64_t *out_off, int32_t *status)
{
	if (!e || !out_arena || !out_off || !status || n < 1 || n > e->max_batch) { g_err = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(e->device));
	{ const int rc = host_buffers(e, n); if (rc) return rc; }
	nhw_launch_synth(e->d_in, n, seed_base, e->own_stream);
	HIPCHK(hipGetLastError());
	{ const int rc = nhw_enc_batch_device(e, e->d_in, n, quality, e->d_out, e->d_sizes, e->d_status, e->own_stream); if (rc) return rc; }
	return host_download(e, n, out_arena, arena_cap, out_off, status);
}

/* ------------------------------------------------------------------------------------------------ stage entry points */
extern "C" int nhw_stage_color(nhw_enc *e, const void *d_bgr, int n, int quality, void *d_y, void *d_u, void *d_v, void *stream)
{
	if (!e || n < 1) return NHW_E_ARG;
	if (quality < 1 || quality > 23) return NHW_E_QUALITY;          /* this stage covers every quality (the whole encoder: 17..23) */
	HIPCHK(hipSetDevice(e->device));
	nhw_launch_color((const uint8_t *)d_bgr, n, quality, (int16_t *)d_y, 8 * Q, (uint8_t *)d_u, (uint8_t *)d_v, Q, stream ? (hipStream_t)stream : e->own_stream);
	HIPCHK(hipGetLastError());
	return NHW_OK;
}

/* the luma pre-filter as a stage exists for quality 1..16 only (k_low_prefilter, the kernel the encoder runs); for 17..21 it is a step
 * inside the fused front kernel and has no output of its own: nhw_debug_stop_after + nhw_debug_read see the planes behind it */
extern "C" int nhw_stage_prefilter(nhw_enc *e, void *d_y, int n, int quality, void *stream)
{
	if (!e || !d_y || n < 1 || n > e->max_batch) { g_err = "bad argument"; return NHW_E_ARG; }
	if (quality < 1 || quality > 16) { g_err = "the pre-filter is a stage of its own only for quality 1..16"; return NHW_E_QUALITY; }
	HIPCHK(hipSetDevice(e->device));
	const NhwWs &ws = e->ws;
	hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
	/* in place for the caller: filter into the workspace plane the encoder uses, copy back */
	HIPCHK((hipError_t)nhw_launch_low_prefilter((const int16_t *)d_y, 4 * Q, plane16(ws, B_KMAP), ws.stride[B_KMAP] / 2, plane16(ws, B_PROC), ws.stride[B_PROC] / 2, plane8(ws, B_SCAN), ws.stride[B_SCAN], plane8(ws, B_KEEP), ws.stride[B_KEEP], (uint16_t *)plane8(ws, B_LOWTAB), ws.stride[B_LOWTAB], quality, n, s));
	HIPCHK(hipMemcpy2DAsync(d_y, 8 * Q, plane16(ws, B_KMAP), ws.stride[B_KMAP], 8 * Q, (size_t)n, hipMemcpyDeviceToDevice, s));
	HIPCHK(hipGetLastError());
	return NHW_OK;
}

/* smallest and largest sample of n planes of W x W shorts (the domain check of the size-512 analysis stage) */
__global__ __launch_bounds__(256) void k_plane_range(const int16_t *__restrict__ base, size_t plane_stride, int *__restrict__ mnmx)
{
	const uint4 *p = reinterpret_cast<const uint4 *>(base + (size_t)blockIdx.y * plane_stride);
	int mn = 32767, mx = -32768;
	for (int i = blockIdx.x * 256 + threadIdx.x; i < W * W / 8; i += gridDim.x * 256) {
		const uint4 v = p[i];
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
		for (int k = 0; k < 4; k++) {
			const int a = (int16_t)(w[k] & 0xFFFF), b = (int16_t)(w[k] >> 16);
			mn = a < mn ? a : mn; mn = b < mn ? b : mn; mx = a > mx ? a : mx; mx = b > mx ? b : mx;
		}
	}
	for (int o = 32; o; o >>= 1) { const int a = __shfl_xor(mn, o), b = __shfl_xor(mx, o); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
	if ((threadIdx.x & 63) == 0) { atomicMin(&mnmx[0], mn); atomicMax(&mnmx[1], mx); }
}

/* one analysis level with the kernels the encoder runs: size 512 = the band kernel on a luma plane (its LL copy goes to the jpeg plane,
 * the second copy it makes to the workspace's ll1), 256 / 128 = the whole-block kernels.
 * Size 512 has a DOMAIN (include/nhw_hip.h, proof in nhw_front_image.h): the level-1 kernel runs both filter passes in packed 16-bit
 * arithmetic, which equals the reference's `int` accumulators (filters.c:203-287, 346-386) only while the second pass stays inside 16 bits.
 * Planes outside it are refused (NHW_E_ARG) -- never answered with a plane that differs from wavelet_analysis(). */
extern "C" int nhw_stage_analysis(nhw_enc *e, void *d_jpeg, void *d_proc, int n_img, size_t plane_stride, int stride, int size,
                                  int final_level, void *stream)
{
	if (!e || n_img < 1) { g_err = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(e->device));
	hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
	if (size == 512) {
		if (stride != W || final_level || n_img > e->max_batch) { g_err = "size 512: stride 512, not the final level, n <= max_batch"; return NHW_E_ARG; }
		if (((uintptr_t)d_jpeg & 15) || ((uintptr_t)d_proc & 15) || (plane_stride & 7) || plane_stride < (size_t)W * W) {   /* the range check and the kernel read 16 bytes at a time */
			g_err = "size 512: planes must be 16-byte aligned and plane_stride (in samples) a multiple of 8, at least 512 x 512"; return NHW_E_ARG;
		}
		const NhwWs &ws = e->ws;
		{                                                          /* the domain check: U = largest sample (or 0), L = -smallest (or 0); 104 U + 40 L and 104 L + 40 U at most NHW_ANA512_BOUND */
			int *d_mm = reinterpret_cast<int *>(plane8(ws, B_ROWSTATE)), mm[2] = { 32767, -32768 };   /* (the front kernel's row-state bytes: free until it runs) */
			HIPCHK(hipMemcpyAsync(d_mm, mm, sizeof mm, hipMemcpyHostToDevice, s));
			k_plane_range<<<dim3(32, n_img), 256, 0, s>>>((const int16_t *)d_jpeg, plane_stride, d_mm);
			HIPCHK(hipMemcpyAsync(mm, d_mm, sizeof mm, hipMemcpyDeviceToHost, s));
			HIPCHK(hipStreamSynchronize(s));
			const long U = mm[1] > 0 ? mm[1] : 0, L = mm[0] < 0 ? -(long)mm[0] : 0;
			if (104 * U + 40 * L > NHW_ANA512_BOUND || 104 * L + 40 * U > NHW_ANA512_BOUND) {
				g_err = "size 512: samples outside the level-1 kernel's 16-bit domain (104 U + 40 L <= 32720, see nhw_hip.h)";
				return NHW_E_ARG;
			}
		}
		/* the level-1 kernel's input is a plane of its own (the caller's jpeg plane receives the LL rows) */
		HIPCHK(hipMemcpy2DAsync(plane16(ws, B_KMAP), ws.stride[B_KMAP], d_jpeg, plane_stride * 2, 8 * Q, (size_t)n_img, hipMemcpyDeviceToDevice, s));
		nhw_launch_front_fused(nullptr, 20, nullptr, nullptr, 0, plane16(ws, B_KMAP), ws.stride[B_KMAP], 0, plane8(ws, B_ROWSTATE), ws.stride[B_ROWSTATE],
		                       (int16_t *)d_proc, (int16_t *)d_jpeg, plane_stride, plane16(ws, B_LL1), ws.stride[B_LL1] / 2, nullptr, 0, n_img, s, 2);
	} else if (size == 256 || size == 128)
		nhw_launch_analysis((int16_t *)d_jpeg, (int16_t *)d_proc, n_img, plane_stride, stride, size, final_level, s);
	else { g_err = "transform size must be 512, 256 or 128"; return NHW_E_ARG; }
	HIPCHK(hipGetLastError());
	return NHW_OK;
}

/* the two chroma level-1 analyses (wavelet_analysis(256, 0, 0) of U and of V, nhw_encoder.c:2265, 2576) exactly as the encoder launches them for quality >= 15:
 * from the 4:2:0 byte planes the front left in the workspace, without the store nothing reads.  A measurement hook: bench.py brackets it with
 * events to add these launches' time to the fused front kernel's (SURVEY 8(d) counts their output among that kernel's bytes). */
extern "C" int nhw_stage_chroma_l1(nhw_enc *e, int n, void *stream)
{
	if (!e || n < 1 || n > e->max_batch) { g_err = "bad argument"; return NHW_E_ARG; }
	if (!e->timed || n > e->last_n || e->last_q < 15) {             /* the byte planes must be those of a whole batch at a quality that launches this form (q >= 15: the analysis widens the bytes itself) */
		g_err = "nhw_stage_chroma_l1: the handle's last batch does not cover the request (needs a completed batch of >= n images at quality >= 15)";
		return NHW_E_ARG;
	}
	HIPCHK(hipSetDevice(e->device));
	const NhwWs &ws = e->ws;
	hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
	HIPCHK(hipStreamWaitEvent(s, e->ev[4], 0));                     /* behind that batch, whatever stream it ran on: its chroma sequence (a stream of the handle) works in the planes written here */
	for (int comp = 0; comp < 2; comp++)
		nhw_launch_analysis(plane16(ws, B_CJPEG), plane16(ws, B_CPROC), n, ws.stride[B_CJPEG] / 2, H, H, 0, s, plane16(ws, B_CLL1), ws.stride[B_CLL1] / 2, H / 2, 2,
		                    comp ? plane8(ws, B_PV) : plane8(ws, B_PU), ws.stride[B_PU], 2);
	HIPCHK(hipGetLastError());
	return NHW_OK;
}

extern "C" int nhw_stage_synthesis(nhw_enc *e, void *d_jpeg, void *d_proc, int n_img, size_t plane_stride, int stride, int size, void *stream)
{
	if (!e || n_img < 1) return NHW_E_ARG;
	if (size != 256 && size != 128) { g_err = "synthesis: transform size must be 256 or 128 (the encoder has no synthesis of size 512)"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(e->device));
	nhw_launch_synthesis((int16_t *)d_jpeg, (int16_t *)d_proc, n_img, plane_stride, stride, size, stream ? (hipStream_t)stream : e->own_stream);
	HIPCHK(hipGetLastError());
	return NHW_OK;
}

/* ------------------------------------------------------------------------------------------------ debug hooks (tests only) */
extern "C" int nhw_debug_front_fallback(nhw_enc *e, int on) { if (!e) return NHW_E_ARG; e->front_fallback = on; return NHW_OK; }
extern "C" int nhw_debug_stop_after(nhw_enc *e, int stage) { if (!e) return NHW_E_ARG; e->stop_after = stage; return NHW_OK; }
/* developer hook: order-independent 64-bit digest of the first `bytes` bytes of workspace buffer `buf`, one per image, into device memory */
__global__ __launch_bounds__(256) void k_debug_hash(const uint8_t *base, size_t stride, size_t words, unsigned long long *out)
{
	const uint32_t *p = reinterpret_cast<const uint32_t *>(base + (size_t)blockIdx.x * stride);
	unsigned long long h = 0;
	for (size_t i = threadIdx.x; i < words; i += 256) {
		unsigned long long x = ((unsigned long long)p[i] + 0x9E3779B97F4A7C15ull) * (2 * i + 1);
		x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
		h += x;
	}
	if (threadIdx.x == 0) out[blockIdx.x] = 0;
	__syncthreads();
	atomicAdd(&out[blockIdx.x], h);
}
extern "C" int nhw_debug_hash(nhw_enc *e, int buf, size_t bytes, int n, void *d_out, void *stream)
{
	if (!e || buf < 0 || buf >= B_COUNT || n < 1 || n > e->max_batch || bytes > e->ws.stride[buf]) return NHW_E_ARG;
	HIPCHK(hipSetDevice(e->device));
	k_debug_hash<<<n, 256, 0, stream ? (hipStream_t)stream : e->own_stream>>>(e->ws.base + e->ws.off[buf], e->ws.stride[buf], bytes / 4, (unsigned long long *)d_out);
	HIPCHK(hipGetLastError());
	return NHW_OK;
}
extern "C" int nhw_debug_fill(nhw_enc *e, int buf, int byte, size_t bytes, int n)
{
	if (!e || buf < 0 || buf >= B_COUNT || n < 1 || n > e->max_batch || bytes + GUARD > e->ws.stride[buf]) return NHW_E_ARG;
	HIPCHK(hipSetDevice(e->device));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemset2D(e->ws.base + e->ws.off[buf], e->ws.stride[buf], byte, bytes, (size_t)n));
	HIPCHK(hipDeviceSynchronize());
	return NHW_OK;
}
extern "C" int nhw_debug_read(nhw_enc *e, int buf, int img, void *dst, size_t bytes)
{
	if (!e || buf < 0 || buf >= B_COUNT || img < 0 || img >= e->max_batch || bytes > e->ws.stride[buf]) return NHW_E_ARG;
	HIPCHK(hipSetDevice(e->device));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(dst, e->ws.base + e->ws.off[buf] + (size_t)img * e->ws.stride[buf], bytes, hipMemcpyDeviceToHost));
	return NHW_OK;
}
