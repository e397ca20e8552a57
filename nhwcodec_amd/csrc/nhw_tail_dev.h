/*
 * nhw_tail_dev.h -- device code of the order-dependent part of the NHW encode path (gfx950).
 *
 * Everything between the filterbank passes of encode_image (rcanut/nhwcodec encoder/nhw_encoder.c:103-2878)
 * is a chain of in-place, raster-order coefficient heuristics, followed by serial byte/bit coders
 * (encoder/image_processing.c:108-521, 2600-3353; encoder/compress_pixel.c:53-1022).  Images are
 * independent, the passes inside one image are not: in this revision every image is owned by one
 * wavefront and the passes run in the reference's order on that wavefront (lane 0 walks the serial
 * chains).  Pass ids (Y5..Y31) are those of SURVEY.md Appendix A; file:line citations are into the
 * reference encoder.  Quality 17..23.
 */
#ifndef NHW_TAIL_DEV_H
#define NHW_TAIL_DEV_H

#include "nhw_ws.h"

#define DEV __device__ static
#define NHW_OK 0
#define NHW_E_CODEBOOK (-2)
#define NHW_E_SPACE (-3)
#define S_CAP 131072   /* capacity of the sign-bit scratch lists */
#ifdef NHW_PROFILE
#define PROF_BEGIN() unsigned long long t0_ = wall_clock64()
#define PROF(c, slot) do { unsigned long long t1_ = wall_clock64(); ((unsigned long long *)((c)->prof))[slot] += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define PROF_BEGIN() do {} while (0)
#define PROF(c, slot) do {} while (0)
#endif

namespace nhw {

DEV inline int iabs(int v) { return v < 0 ? -v : v; }

struct PosList { uint8_t *list, *bits, *word; NhwPosLens *len; };

/* per-image view of the workspace; the scalar encoder state (reference encode_state, codec.h:125-181) lives in
 * NhwMeta in global memory so that every thread of the workgroup sees one copy */
struct Ctx {
	int q;
	int compat;                    /* NhwWs::compat */
	const int16_t *stale;          /* compat mode: kernel-map cells (k_front_stale) */
	int16_t *jpeg, *proc, *cjpeg, *cproc, *ll1, *l2save, *cll1, *cl2save, *keep, *first_order, *band, *hs, *tmp16;
	uint8_t *pu, *pv, *scan, *ll_bytes, *ll_full, *exw, *res4, *ll_comp, *ll_word, *ch_res, *res_u64, *res_v64;
	uint8_t *sel_word1, *sel_word2, *book1, *book2, *raw, *pay, *cc, *half, *s1, *s2;
	uint16_t *ll_mem, *char_res1;
	uint32_t *qsetting3, *packet;
	int *hist;
	void *prof;
	NhwMeta *m;
	PosList res1, res3, res5, res6;
};

DEV void poslist_finish(Ctx *c, PosList *pl, uint8_t *raw, int raw_len, const uint8_t *payload, int payload_len, int word_mode);

DEV void ctx_load(Ctx *c, const NhwWs &ws, int img)
{
	NhwMeta *m = ws.buf<NhwMeta>(B_META, img);
	c->m = m;
	c->q = ws.q;
	c->compat = ws.compat; c->stale = ws.buf<int16_t>(B_STALE, img);
	c->jpeg = ws.buf<int16_t>(B_JPEG, img); c->proc = ws.buf<int16_t>(B_PROC, img);
	c->cjpeg = ws.buf<int16_t>(B_CJPEG, img); c->cproc = ws.buf<int16_t>(B_CPROC, img);
	c->ll1 = ws.buf<int16_t>(B_LL1, img); c->l2save = ws.buf<int16_t>(B_L2SAVE, img);
	c->cll1 = ws.buf<int16_t>(B_CLL1, img); c->cl2save = ws.buf<int16_t>(B_CL2SAVE, img);
	c->keep = ws.buf<int16_t>(B_KEEP, img); c->first_order = ws.buf<int16_t>(B_FIRST, img);
	c->band = ws.buf<int16_t>(B_BAND, img); c->hs = ws.buf<int16_t>(B_HS, img); c->tmp16 = ws.buf<int16_t>(B_TMP16, img);
	c->pu = ws.buf<uint8_t>(B_PU, img); c->pv = ws.buf<uint8_t>(B_PV, img); c->scan = ws.buf<uint8_t>(B_SCAN, img);
	c->ll_bytes = ws.buf<uint8_t>(B_LLBYTES, img); c->ll_full = ws.buf<uint8_t>(B_LLFULL, img);
	c->exw = ws.buf<uint8_t>(B_EXW, img); c->res4 = ws.buf<uint8_t>(B_RES4, img);
	c->ll_comp = ws.buf<uint8_t>(B_LLCOMP, img); c->ll_word = ws.buf<uint8_t>(B_LLWORD, img);
	c->ch_res = c->ll_comp;
	c->res_u64 = ws.buf<uint8_t>(B_RESU64, img); c->res_v64 = ws.buf<uint8_t>(B_RESV64, img);
	c->sel_word1 = ws.buf<uint8_t>(B_SEL1, img); c->sel_word2 = ws.buf<uint8_t>(B_SEL2, img);
	c->book1 = ws.buf<uint8_t>(B_BOOK1, img); c->book2 = ws.buf<uint8_t>(B_BOOK2, img);
	c->raw = ws.buf<uint8_t>(B_RAW, img); c->pay = ws.buf<uint8_t>(B_PAY, img);
	c->cc = ws.buf<uint8_t>(B_CC, img); c->half = ws.buf<uint8_t>(B_HALF, img);
	c->s1 = ws.buf<uint8_t>(B_S1, img); c->s2 = ws.buf<uint8_t>(B_S2, img);
	c->ll_mem = ws.buf<uint16_t>(B_LLMEM, img); c->char_res1 = ws.buf<uint16_t>(B_CHARRES, img);
	c->qsetting3 = ws.buf<uint32_t>(B_QSET3, img); c->packet = ws.buf<uint32_t>(B_PACKET, img);
	c->hist = ws.buf<int>(B_HIST, img);
	c->prof = ws.buf<uint8_t>(B_PROF, img);
	c->res1.list = ws.buf<uint8_t>(B_R1LIST, img); c->res1.bits = ws.buf<uint8_t>(B_R1BITS, img); c->res1.word = ws.buf<uint8_t>(B_R1WORD, img); c->res1.len = &m->r1;
	c->res3.list = ws.buf<uint8_t>(B_R3LIST, img); c->res3.bits = ws.buf<uint8_t>(B_R3BITS, img); c->res3.word = ws.buf<uint8_t>(B_R3WORD, img); c->res3.len = &m->r3;
	c->res5.list = ws.buf<uint8_t>(B_R5LIST, img); c->res5.bits = ws.buf<uint8_t>(B_R5BITS, img); c->res5.word = ws.buf<uint8_t>(B_R5WORD, img); c->res5.len = &m->r5;
	c->res6.list = ws.buf<uint8_t>(B_R6LIST, img); c->res6.bits = ws.buf<uint8_t>(B_R6BITS, img); c->res6.word = ws.buf<uint8_t>(B_R6WORD, img); c->res6.len = &m->r6;
}

__device__ static const uint8_t k_big_pos[19] = { 10, 12, 14, 18, 20, 22, 26, 28, 30, 34, 36, 38, 42, 44, 46, 50, 52, 54, 58 };
__device__ static const uint8_t k_big_neg[19] = { 60, 62, 66, 68, 70, 74, 76, 78, 82, 84, 86, 90, 92, 94, 98, 100, 102, 106, 108 };

DEV int odd(int v) { return (v & 1) == 1; }
DEV int in_4_7(int v) { return v > 3 && v <= 7; }
DEV int in_m7_m4(int v) { return v < -3 && v >= -7; }
DEV int is_567(int v) { return v == 5 || v == 6 || v == 7; }
DEV int is_m567(int v) { return v == -5 || v == -6 || v == -7; }
DEV int16_t clear_bit0(int v) { return (int16_t)(v & 0xFFFE); }

/* shared tail of the dequantiser: dead zone, bias by 128, floor the magnitude to a multiple of 8,
 * then the decoder's reconstruction offsets (image_processing.c:3003-3015) */
DEV int dequant_value(int a)
{
	if (a < DEADZONE && a > -DEADZONE) return 0;
	a += 128;
	if (a < 0) a = -((-a) & 0xFFF8); else a &= 0xFFF8;
	return a > 128 ? a - 125 : a - 131;
}

/* triple / vertical-pair pattern marking shared by rows<128 (cols 129..254) and rows 128..254
 * (cols 1..254): image_processing.c:2759-2853 */
DEV void mark_small_runs(int16_t *p, int16_t *jp, int row0, int row1, int col0)
{
	int r, j;
	for (r = row0; r < row1; r++)
		for (j = col0; j < H - 1; j++) {
			const int a = r * W + j;
			if (p[a] > 3 && p[a] < 8) {
				if (in_4_7(p[a - 1])) {
					if (in_4_7(p[a + 1])) { p[a - 1] = 15300; p[a] = 0; jp[a] = 5; jp[a + 1] = 5; j++; }
					else if (in_4_7(p[a + W - 1]) && in_4_7(p[a + W])) {
						p[a - 1] = 15500; jp[a] = 5; p[a + W - 1] = 15500; jp[a + W] = 5; p[a + W] = 0; j++;
					}
				}
			} else if (p[a] < -3 && p[a] > -8) {
				if (in_m7_m4(p[a - 1])) {
					if (in_m7_m4(p[a + 1])) { p[a - 1] = 15400; p[a] = 0; jp[a] = -6; jp[a + 1] = -5; j++; }
					else if (in_m7_m4(p[a + W - 1]) && in_m7_m4(p[a + W])) {
						p[a - 1] = 15600; jp[a] = -5; p[a + W - 1] = 15600; jp[a + W] = -5; p[a + W] = 0; j++;
					}
				}
			}
		}
}

/* equal-sign 5..7 pairs: image_processing.c:2857-2905 */
DEV void mark_pairs(int16_t *p, int row0, int row1, int col0)
{
	int r, j;
	for (r = row0; r < row1; r++)
		for (j = col0; j < H - 1; j++) {
			const int a = r * W + j;
			if (is_567(p[a])) { if (is_567(p[a + 1])) { p[a] = 15700; j++; } }
			else if (is_m567(p[a])) { if (is_m567(p[a + 1])) { p[a] = 15800; j++; } }
		}
}

/* per-row dequantisation of detail bands: image_processing.c:2909-3015 and 3018-3124 */
DEV void dequant_rows(int16_t *p, int16_t *jp, int row0, int row1, int col0, int part)
{
	int r, j;
	for (r = row0; r < row1; r++)
		for (j = col0; j < H; j++) {
			const int at = r * W + j;
			int a = p[at];
			if (a > 15000) {
				if (a == 15300) { jp[at] = 5; j += 2; }
				else if (a == 15400) { jp[at] = -5; j += 2; }
				else if (a == 15500) { jp[at] = 5; j++; }
				else if (a == 15600) { jp[at] = -5; j++; }
				else if (a == 15700) { jp[at] = 6; jp[at + 1] = 6; j++; }
				else if (a == 15800) { jp[at] = -6; jp[at + 1] = -6; j++; }
				continue;
			}
			if (a < -12 && ((-a) & 7) == 6) { if (j < H - 1 && p[at + 1] == -7) p[at + 1] = -8; }
			if (a < 0) {
				if (a == -7 && j < H - 1 && p[at + 1] == 8) { p[at] = -8; a = -8; }
				a = -a;
				if ((a & 7) < 7) a &= 0xFFF8;
				a = -a;
			}
			else if (a == 8 && j < H - 1 && p[at + 1] == -7) p[at + 1] = -8;
			else if (a > 12 && !part && (a & 7) >= 6) { if (j < H - 1 && p[at + 1] == 7) p[at + 1] = 8; }
			jp[at] = (int16_t)dequant_value(a);
		}
}

/* a8: offsetY_recons256, image_processing.c:2600-3190.  `part` 1 = first closed loop, 0 = second. */
DEV void dequant_sim_luma(Ctx *c, int part)
{
	int16_t *p = c->proc, *jp = c->jpeg;
	const int q = c->q;
	int r, j;

	if (q > 17) {                                    /* :2609-2640, four odd LL2 samples in a row */
		for (r = 0; r < H / 2; r++)
			for (j = 0; j < H / 2 - 3; j++) {
				const int a = r * W + j;
				if (odd(p[a]) && odd(p[a + 1]) && odd(p[a + 2]) && odd(p[a + 3]) && iabs(p[a] - p[a + 3]) > 1) {
					if (!part) { p[a] += 16000; p[a + 1] += 16000; p[a + 2] += 16000; p[a + 3] += 16000; }
					else { p[a] += 16000; p[a + 2] += 16000; }
					j += 3;
				}
			}
	}

	for (r = 0; r < H / 2; r++)                      /* :2642-2695 */
		for (j = 0; j < H / 2; j++) {
			int a = r * W + j;
			if (p[a] > 10000) {
				if (!part) jp[a] = p[a];
				else {
					p[a] -= 16000; jp[a] = p[a];
					jp[a + 1] = (p[a + 1] > 0 && p[a + 1] < 256) ? clear_bit0(p[a + 1]) : p[a + 1];
					j++;
				}
				continue;
			}
			else if (odd(p[a]) && j > 0 && odd(p[a + 1])) {
				if (j < H / 2 - 2 && odd(p[a + 2])) { if (iabs(p[a] - p[a + 2]) > 1 && q > 17) p[a + 1]++; }
				else if (r * W < Q - W - 2 && odd(p[a + W]) && odd(p[a + W + 1]) && !(p[a + W + 2] & 1)) {
					if (p[a + W] < 10000 && q > 17) p[a + W]++;
				}
			}
			else if (odd(p[a]) && r >= 1 && r * W < Q - 3 * W) {
				if (odd(p[a + W]) && odd(p[a + W + 1]) && odd(p[a + 2 * W]) && !(p[a + 3 * W] & 1)) {
					if (p[a + W] < 10000 && q > 17) p[a + W]++;
				}
			}
			if (part) jp[a] = (p[a] > 0 && p[a] < 256) ? clear_bit0(p[a]) : p[a];
		}

	if (!part) {                                     /* :2697-2735 */
		int16_t *tmp = c->tmp16;
		int t = 0, i;
		for (r = 0; r < H / 2; r++)
			for (j = 0; j < H / 2; j++) {
				const int a = r * W + j;
				if (p[a] < 10000) {
					tmp[t++] = p[a];
					jp[a] = (p[a] >= 0 && p[a] < 256) ? clear_bit0(p[a]) : p[a];
				} else {
					p[a] -= 16000; tmp[t++] = p[a]; jp[a] = p[a];
				}
			}
		/* samples the LL coder sent verbatim keep their exact value (q>15) */
		for (i = 0; i < c->m->ll_mem_len; i++) {
			const int idx = c->ll_mem[i];
			jp[((idx >> 7) << 9) + (idx & 127)] = tmp[idx];
		}
	}

	/* q>16 from here (:2757) */
	mark_small_runs(p, jp, 0, H / 2, H / 2 + 1);
	mark_small_runs(p, jp, H / 2, H - 1, 1);
	if (!part) {
		mark_pairs(p, 0, H / 2, H / 2);
		mark_pairs(p, H / 2, H, 0);
	}
	dequant_rows(p, jp, 0, H / 2, H / 2, part);
	dequant_rows(p, jp, H / 2, H, 0, part);

	if (!part) {                                     /* :3154-3188 isolated coefficient shrink (q>16 form) */
		for (r = 1; r < H - 1; r++)
			for (j = 1; j < H - 1; j++) {
				const int e = r * W + j;
				if (iabs(jp[e]) >= 8) {
					if (iabs(jp[e - W - 1]) >= 8 || iabs(jp[e - W]) >= 8 || iabs(jp[e - W + 1]) >= 8 ||
					    iabs(jp[e - 1]) >= 8 || iabs(jp[e + 1]) >= 8 ||
					    iabs(jp[e + W - 1]) >= 8 || iabs(jp[e + W]) >= 8 || iabs(jp[e + W + 1]) >= 8) continue;
					if (r >= H / 2 || j >= H / 2) { if (jp[e] > 0) jp[e]--; else jp[e]++; }
				}
			}
	}
}

/* offsetUV_recons256, image_processing.c:3192-3353 (q>15 form of the LL part) */
DEV void dequant_rows_chroma(int16_t *p, int16_t *jp, int row0, int row1, int col0, int comp)
{
	int r, j;
	for (r = row0; r < row1; r++)
		for (j = col0; j < H / 2; j++) {
			const int at = r * H + j;
			int a = p[at];
			if ((a == -7 || a == -8) && !comp) {
				if (j < H / 2 - 1 && (p[at + 1] == -7 || p[at + 1] == -8)) { jp[at] = -11; jp[at + 1] = -11; j++; continue; }
			}
			if (a < 0) {
				a = -a;
				if (p[at + 1] < 0 && p[at + 1] > -8) { if ((a & 7) < 6) a &= 0xFFF8; }
				else { if ((a & 7) < 7) a &= 0xFFF8; }
				a = -a;
			}
			jp[at] = (int16_t)dequant_value(a);
		}
}

DEV void dequant_sim_chroma(Ctx *c, int comp)
{
	int16_t *p = c->cproc, *jp = c->cjpeg;
	int r, j;
	for (r = 0; r < H / 4; r++)
		for (j = 0; j < H / 4; j++) {
			const int i = r * H + j;
			if (comp) {                              /* :3198-3219 alternate which sample of a pair keeps bit 0 */
				if (r == 0) { jp[i] = p[i]; jp[i + 1] = clear_bit0(p[i + 1]); }
				else { jp[i] = clear_bit0(p[i]); jp[i + 1] = p[i + 1]; }
				j++;
			} else {                                 /* :3232-3242 */
				jp[i] = (p[i] > 0 && p[i] < 256) ? clear_bit0(p[i]) : p[i];
			}
		}
	dequant_rows_chroma(p, jp, 0, H / 4, H / 4, comp);
	dequant_rows_chroma(p, jp, H / 4, H / 2, 0, comp);
}

DEV int big_code(int a, const uint8_t *tab)
{
	int k = ((a & 0xFFF8) - 128) >> 3;
	return tab[k > 18 ? 18 : k];
}

/* a10: offsetY, image_processing.c:185-521 (q>16 branches) */
DEV void quantise_luma(Ctx *c)
{
	int16_t *p = c->proc;
	int i, r, j;

	for (i = 0; i < 4 * Q; i++) {                    /* :195-238 paired multiples of 8 in detail bands */
		const int col = i & (W - 1);
		if (!(i >= 2 * Q || col >= H)) continue;
		if (p[i] > 7 && p[i + 1] > 7 && col < W - 1) {
			const int a = p[i];
			if (!(a & 7) && !(p[i + 1] & 7)) {
				if (a > 15) {
					if (i > 0) {
						if (p[i - 1] <= 0) p[i]--;
						else if (p[i + 1] > 15) { if (col < W - 2 && p[i + 2] <= 0) p[i + 1]--; }
					}
				}
				else if (p[i + 1] > 15) { if (col < W - 2 && p[i + 2] <= 0) p[i + 1]--; }
			}
		}
	}

	for (r = 0; r < H; r++)                          /* :241-284 */
		for (j = 1; j < H - 1; j++) {
			const int a = r * W + j;
			if (p[a] > 3 && p[a] < 8) {
				if (in_4_7(p[a - 1])) {
					if (in_4_7(p[a + 1])) { p[a] = 12700; p[a - 1] = 10100; j++; }
					else if (in_4_7(p[a + W - 1]) && in_4_7(p[a + W])) {
						p[a - 1] = 12100; p[a] = 10100; p[a + W - 1] = 10100; p[a + W] = 10100; j++;
					}
				}
			} else if (p[a] < -3 && p[a] > -8) {
				if (in_m7_m4(p[a - 1])) {
					if (in_m7_m4(p[a + 1])) { p[a] = 12900; p[a - 1] = 10100; j++; }
					else if (in_m7_m4(p[a + W - 1]) && in_m7_m4(p[a + W])) {
						p[a - 1] = 12200; p[a] = 10100; p[a + W - 1] = 10100; p[a + W] = 10100; j++;
					}
				}
			}
		}
	for (r = 0; r < H; r++)                          /* :286-311 */
		for (j = 0; j < H - 1; j++) {
			const int a = r * W + j;
			if (is_567(p[a])) { if (is_567(p[a + 1])) { p[a] = 10300; j++; } }
			else if (is_m567(p[a])) { if (is_m567(p[a + 1])) { p[a] = 10204; j++; } }
		}

	for (i = 0; i < 4 * Q; i++) {                    /* :314-519 */
		const int col = i & (W - 1);
		int a = p[i];
		if (a > 10000) {
			if (a == 10100) { p[i] = 128; continue; }
			else if (a == 12700) { p[i] = 127; continue; }
			else if (a == 12900) { p[i] = 129; continue; }
			else if (a == 10204) { p[i] = 125; continue; }
			else if (a == 10300) { p[i] = 126; continue; }
			else if (a == 12100) { p[i] = 121; continue; }
			else if (a == 12200) { p[i] = 122; continue; }
		}
		if (a > 127) { p[i] = (int16_t)big_code(a, k_big_pos); continue; }
		else if (a < -127) { p[i] = (int16_t)big_code(-a, k_big_neg); continue; }

		if (a < -12 && ((-a) & 7) == 6) { if (col < W - 1 && p[i + 1] == -7) p[i + 1] = -9; }
		if (a < 0) {
			if (a == -7 && p[i + 1] == 8 && col < W - 1) { p[i] = -8; a = -8; }
			a = -a;
			if (a > 14 && (a & 7) == 7 && p[i + 1] > 0 && p[i + 1] < 8) a -= 2;
			if ((a & 7) < 7) a &= 504;
			a = -a;
		}
		else if (a == 8 && p[i + 1] == -7 && col < W - 1) p[i + 1] = -8;
		else if (a > 12 && (a & 7) >= 6) { if (col < W - 1 && p[i + 1] == 7) p[i + 1] = 9; }

		if (a < DEADZONE && a > -DEADZONE) p[i] = 128;
		else p[i] = (int16_t)((a + 128) & 248);
	}
}

/* offsetUV, image_processing.c:108-183 */
DEV void quantise_chroma(Ctx *c)
{
	int16_t *p = c->cproc;
	int i;
	for (i = 0; i < Q; i++) {
		int a = p[i];
		if (a > 10000) {
			if (a == 12400) { p[i] = 124; continue; }
			else if (a == 12600) { p[i] = 126; continue; }
			else if (a == 12900) { p[i] = 122; continue; }
			else if (a == 13000) { p[i] = 130; continue; }
		}
		if (a > 127) { p[i] = (int16_t)big_code(a, k_big_pos); continue; }
		else if (a < -127) { p[i] = (int16_t)big_code(-a, k_big_neg); continue; }

		if ((a == -7 || a == -8) && (i & 255) < H - 1 && (p[i + 1] == -7 || p[i + 1] == -8)) {
			p[i] = 120; p[i + 1] = 120; i++; continue;
		}
		if (a < 0) {
			a = -a;
			if (p[i + 1] < 0 && p[i + 1] > -8) { if ((a & 7) < 6) a &= 504; }
			else { if ((a & 7) < 7) a &= 504; }
			a = -a;
		}
		else if (a > 6 && (a & 7) >= 6) { if ((i & 255) < H - 1 && p[i + 1] == 7) p[i + 1] = 8; }

		if (a < DEADZONE && a > -DEADZONE) p[i] = 128;
		else p[i] = (int16_t)((a + 128) & 248);
	}
}





DEV int mult8_or_7(int m) { return !(m & 7) || (m & 7) == 7; } /* on a magnitude */

/* Y5: mark L2 detail coefficients whose quantisation error sign is predictable (nhw_encoder.c:144-177) */
DEV void tag_l2_details(Ctx *c)
{
	const int16_t *p = c->proc;
	int r, j;
	for (r = 0; r < H; r++)
		for (j = 0; j < H; j++) {
			const int at = r * W + j, s = p[at];
			int16_t *cell = c->ll1 + r * H + j;
			if (r < H / 2 && j < H / 2) continue;
			if (s < -7) { if (mult8_or_7(-s)) *cell += 16000; }
			else if (s < -4) *cell += 12000;
			else if (s >= 0) {
				if (s >= 2 && s < 5) {
					if (at >= W + 1 && at < 2 * Q - W - 1 && (p[at - (W + 1)] != 0 || p[at + (W + 1)] != 0)) *cell += 12000;
				}
				else if (!(s & 7)) *cell += 12000;
				else if ((s & 7) == 1) *cell += 12000;
				else if (s > 4 && s <= 7) *cell += 16000;
			}
		}
}

/* Y8: nudge the reconstructed LL1 sample that sits under each tagged coefficient (:183-216) */
DEV void apply_tags(Ctx *c)
{
	int16_t *p = c->proc;
	int r, j;
	for (r = 0; r < H; r++)
		for (j = 0; j < H; j++) {
			int16_t *cell = c->ll1 + r * H + j;
			int step;
			if (*cell > 14000) { *cell -= 16000; step = 1; }
			else if (*cell > 10000) { *cell -= 12000; step = -1; }
			else continue;
			if (r < H / 2 && j >= H / 2) p[(2 * (j - H / 2) + 1) * W + 2 * r] += step;
			else if (r >= H / 2 && j < H / 2) p[2 * j * W + 2 * (r - H / 2) + 1] += step;
			else if (r >= H / 2 && j >= H / 2) p[(2 * (j - H / 2) + 1) * W + 2 * (r - H / 2) + 1] += step;
		}
}

DEV int big_step(int d) /* correction for a large closed-loop error (:225-232) */
{
	if (d > 11) return -7; if (d > 7) return -4; if (d > 5) return -2; if (d > 4) return -1;
	if (d < -11) return 7; if (d < -7) return 4; if (d < -5) return 2; if (d < -4) return 1;
	return 0;
}

/* Y9: LL1 pre-compensation, strictly left to right (:218-279) */
DEV void precompensate_ll1(Ctx *c)
{
	int16_t *p = c->proc, *jp = c->jpeg;
	const int16_t *o = c->ll1;
	int r, j;
	for (r = 0; r < H; r++)
		for (j = 0; j < H; j++) {
			const int e = r * W + j, k = r * H + j, d = p[e] - o[k];
			int step = big_step(d);
			if (!step && iabs(d) > 1) {
				int a = p[e + 1] - o[k + 1];
				if (iabs(a) > 4) a += big_step(a) ? big_step(a) : (a > 0 ? -1 : 1);
				a += p[e - 1] - o[k - 1];
				if (d >= 4 && a >= 1) step = -1;
				else if (d <= -4 && a <= -1) step = 1;
				else if (d == 3 && a >= 0) step = -1;
				else if (d == -3 && a <= 0) step = 1;
				else if (iabs(a) >= 3) {
					if (d > 0 && a > 0) step = -1;
					else if (d < 0 && a < 0) step = 1;
					else if (a >= 5) step = -2;
					else if (a <= -5) step = 2;
					else if (a >= 4) step = -1;
					else if (a <= -4) step = 1;
				}
			}
			jp[e] = (int16_t)(o[k] + step);
			p[e] = (int16_t)(p[e] + step);
		}
}

/* Y14: four odd LL2 samples in a row -> nhw_res4 (:636-657) */
DEV void tag_res4(Ctx *c)
{
	int16_t *p = c->proc;
	int r, j, n = 0;
	for (r = 0; r < H / 2; r++) {
		int hit = 0;
		for (j = 0; j < H / 2 - 3; j++) {
			const int a = r * W + j;
			if (odd(p[a]) && odd(p[a + 1]) && odd(p[a + 2]) && odd(p[a + 3]) && iabs(p[a] - p[a + 3]) > 1) {
				p[a] += 24000; p[a + 1] += 16000; p[a + 2] += 16000; p[a + 3] += 16000;
				n++; hit++; j += 3;
			}
		}
		if (!hit) n++;
	}
	c->m->res4_len = n;
}

/* Y15: LL2 emission (:661-741) */
DEV void emit_ll2(Ctx *c)
{
	int16_t *p = c->proc;
	const int q = c->q;
	int r, j, a = 0, e = 0, n4 = 0;
	for (r = 0; r < H / 2; r++) {
		int hit = 0;
		for (j = 0; j < H / 2; j++) {
			const int at = r * W + j;
			int s = p[at];
			if (q > 17 && s > 10000) {
				if (s > 20000) { s -= 24000; c->res4[n4++] = (uint8_t)(j + 1); hit++; }
				else s -= 16000;
			}
			else if (odd(s) && j > 0 && odd(p[at + 1])) {
				if (j < H / 2 - 2 && odd(p[at + 2])) { if (iabs(s - p[at + 2]) > 1 && q > 17) p[at + 1]++; }
				else if (r * W < Q - W - 2 && odd(p[at + W]) && odd(p[at + W + 1]) && !(p[at + W + 2] & 1)) {
					if (p[at + W] < 10000 && q > 17) p[at + W]++;
				}
			}
			else if (odd(s) && r >= 1 && r * W < Q - 3 * W) {
				if (odd(p[at + W]) && odd(p[at + W + 1]) && odd(p[at + 2 * W]) && !(p[at + 3 * W] & 1)) {
					if (p[at + W] < 10000 && q > 17) p[at + W]++;
				}
			}

			if ((s > 255 || s < 0) && (j > 0 || r > 0)) {   /* out of byte range: escape triple, repeat previous */
				int mag;
				c->exw[e++] = (uint8_t)r;
				if (s > 255) { c->exw[e++] = (uint8_t)(j + 128); mag = s - 255; }
				else { c->exw[e++] = (uint8_t)j; mag = -s; }
				c->exw[e++] = (uint8_t)(mag > 255 ? 255 : mag);
				c->ll_bytes[a] = c->ll_bytes[a - 1]; c->ll_full[a] = c->ll_bytes[a - 1]; a++;
			} else {
				if (s > 255) s = 255; else if (s < 0) s = 0;
				c->ll_full[a] = (uint8_t)s; c->ll_bytes[a++] = (uint8_t)(s & 254);
			}
			p[at] = 0;
		}
		if (q > 17) {
			if (!hit) c->res4[n4++] = 128; else c->res4[n4 - 1] += 128;
		}
	}
	c->m->exw_len = e;
}

/* Y21: +-5..7 run tagging (:970-1073) */
DEV void tag_small_runs(Ctx *c)
{
	int16_t *p = c->proc;
	int r, j;
	for (r = 1; r < H - 1; r++)
		for (j = H + 1; j < W - 1; j++) {
			int16_t *v = p + r * W + j;
			if (v[0] > 4 && v[0] < 8) { if (in_4_7(v[-1]) && in_4_7(v[1])) { v[0] = 12700; v[-1] = 10100; v[1] = 10100; } }
			else if (v[0] < -4 && v[0] > -8) { if (in_m7_m4(v[-1]) && in_m7_m4(v[1])) { v[0] = 12900; v[-1] = 10100; v[1] = 10100; } }
			else if (v[0] == 8) {
				if ((v[-1] & 0xFFFE) == 6 || (v[1] & 0xFFFE) == 6) v[0] = 10;
				else if (v[1] == 8) { v[0] = 9; v[1] = 9; }
			}
			else if (v[0] == -8) {
				if (((-v[-1]) & 0xFFFE) == 6 || ((-v[1]) & 0xFFFE) == 6) v[0] = -9;
				else if (v[1] == -8) { v[0] = -9; v[1] = -9; }
			}
			/* the reference's (-7,-6/-7) and (7,7) branches (:995-1002) are unreachable: 5..7 and
			 * -7..-5 are consumed by the two tests above */
		}
	for (r = H + 1; r < W - 1; r++)
		for (j = 1; j < H - 1; j++) {
			int16_t *v = p + r * W + j;
			if (v[0] > 4 && v[0] < 8) { if (in_4_7(v[-1]) && in_4_7(v[1])) { v[0] = 12700; v[-1] = 10100; v[1] = 10100; } }
			else if (v[0] < -4 && v[0] > -8) { if (in_m7_m4(v[-1]) && in_m7_m4(v[1])) { v[0] = 12900; v[-1] = 10100; v[1] = 10100; } }
			else if (v[0] == 8) { if ((v[-1] & 0xFFFE) == 6 || (v[1] & 0xFFFE) == 6) v[0] = 10; }
			else if (v[0] == -8) { if (((-v[-1]) & 0xFFFE) == 6 || ((-v[1]) & 0xFFFE) == 6) v[0] = -9; }
			/* :1040-1064 (-6/-7 and 7 with a vertical partner) unreachable for the same reason */
		}
}

/* Y22/Y23 nudges of the LH1 coefficient paired with an LL1 sample */
DEV void nudge_up_small(int16_t *v)                       /* L_W1 (:1251-1262) */
{
	if (v[0] == 7) { if (v[-1] >= 0 && v[-1] < 8) v[0] += 2; }
	else if (v[0] == 8) { if (v[-1] >= -2 && v[-1] < 8) v[0] += 2; }
}
DEV void nudge_m2(int16_t *v)                             /* L_W2 (:1264-1275) */
{
	if (v[0] < -14) { if (mult8_or_7(-v[0])) v[0]++; }
	else if (v[0] == 7 || (v[0] & 0xFFFE) == 8) { if (v[-1] >= -2) v[0] += 3; }
}
DEV void nudge_m3(Ctx *c, int16_t *v, int16_t *cell)  /* L_W3 (:1277-1294) */
{
	if (c->q >= 21) *cell = 14500;
	else if (v[0] < -14) { if (mult8_or_7(-v[0])) v[0]++; }
	else if (v[0] >= 0 && ((v[0] + 2) & 0xFFFC) == 8) { if (v[-1] >= -2) v[0] = 10; }
	else if (v[0] > 14 && (v[0] & 7) == 7) v[0]++;
}
DEV void mark_m_large(Ctx *c, int16_t *v, int16_t *cell, int res) /* L_W5 (:1296-1325) */
{
	*cell = 14000;
	if (res == -4) { if (v[0] == -7 || v[0] == -8) { if (v[-1] < 2 && v[-1] > -8) v[0] = -9; } }
	else if (res < -6) {
		if (res < -7 && c->q >= 21) *cell = 14900;
		else if (v[0] < -14) { if (mult8_or_7(-v[0])) v[0]++; }
		else if (v[0] == 7 || v[0] == 8) { if (v[-1] >= -1 && v[-1] < 8) v[0] += 3; }
	}
}

/* Y22: residual classification, column by column (:1084-1325) */
DEV void classify_residuals(Ctx *c, int res_setting)
{
	int16_t *p = c->proc, *o = c->ll1;
	const int q = c->q;
	int j, r;
	for (j = 0; j < H; j++)
		for (r = 0; r < H - 1; r++) {
			const int s = r * W + j, k = r * H + j;
			int16_t *cell = o + k;
			int16_t *lh = p + j * W + H + r;            /* (j<<9)+(i>>9)+IM_DIM */
			const int res = p[s] - o[k], a = p[s + W] - o[k + H];
			const int d2 = p[s + 2 * W] - o[k + 2 * H];  /* two rows down: reads past ll1 on the last rows */
#define MARK(code, step) do { *cell = (code); p[s + W] += (step); p[s + 2 * W] += (step); } while (0)
#define SNAP(code) do { *cell = (code); p[s + W] = o[k + H]; } while (0)
			if (res == 2 && a == 2 && d2 >= 2) { if (d2 < 5 || d2 > 6) MARK(12400, -2); }
			else if (((res == 2 && a == 3) || (res == 3 && a == 2)) && d2 > 1 && d2 < 6) MARK(12400, -2);
			else if (res == 3 && a == 3) {
				if (d2 > 0 && d2 < 6) MARK(12400, -2);
				else if (q >= 19) SNAP(12100);
			}
			else if (a == -4 && (res == 2 || res == 3) && (d2 == 2 || d2 == 3)) {
				if (res == 2 && d2 == 2) p[s + W]++; else MARK(12400, -2);
			}
			else if (res == 1 && a == 3 && d2 == 2) {
				if (r > 0 && (p[s - W] - o[k - H]) >= 0) MARK(12400, -2);
			}
			else if ((res == 3 || res == 4 || res == 5 || res > 6) && (a == 3 || (a & 0xFFFE) == 4)) {
				if (res > 6) SNAP(12500);
				else if (q >= 19) SNAP(12100);
				else if (q == 18) {
					if (res < 5 && a == 5) o[k + H] = 14100;
					else if (res >= 5) *cell = 14100;
					else if (res == 3 && a >= 4) o[k + H] = 14100;
					p[s + W] = o[k + H];
				}
			}
			else if ((res == 2 || res == 3) && (a == 2 || a == 3)) {
				if (d2 == 0 || d2 == 1) {
					const int x0 = p[s + 1] - o[k + 1], x1 = p[s + W + 1] - o[k + H + 1];
					if ((x0 == 2 || x0 == 3) && (x1 == 2 || x1 == 3) && (p[s + 2 * W + 1] - o[k + 2 * H + 1]) > 0) MARK(12400, -2);
				}
			}
			else if (a == 4 && (res == -2 || res == -3) && (d2 == -2 || d2 == -3)) {
				if (res == -2 && d2 == -2) p[s + W]--; else MARK(12300, 2);
			}
			else if ((res == -3 || res == -4 || res == -5 || res < -7) && (a == -3 || a == -4 || a == -5)) {
				if (res < -7) SNAP(12600);
				else if (q >= 19) SNAP(12200);
				else if (q == 18) {
					if (res > -5 && a == -5) o[k + H] = 14000;
					else if (res <= -5) *cell = 14000;
					else if (res == -3 && a <= -4) o[k + H] = 14000;
					p[s + W] = o[k + H];
				}
			}
			else if (a == -2 || a == -3) {
				if (res == -2 || res == -3) {
					if (d2 < 0) MARK(12300, 2);
					else if (res == -3 && q >= 21) *cell = 14500;
					else if (d2 == 0) {
						const int x0 = p[s + 1] - o[k + 1], x1 = p[s + W + 1] - o[k + H + 1];
						if ((x0 == -2 || x0 == -3) && (x1 == -2 || x1 == -3) && (p[s + 2 * W + 1] - o[k + 2 * H + 1]) < 0) MARK(12300, 2);
					}
					else if (res == -2) nudge_m2(lh);
					else nudge_m3(c, lh, cell);
				}
				else if (res == -1 && a == -3 && d2 == -2) {
					if (r > 0 && (p[s - W] - o[k - H]) <= 0) MARK(12300, 2);
				}
				else if (res == -1) { if (d2 == -3) MARK(12300, 2); else nudge_up_small(lh); }
				else if (res == -4) { if (d2 < -1 && d2 > -4) MARK(12300, 2); else mark_m_large(c, lh, cell, res); }
			}
			else if (!res || res == -1) nudge_up_small(lh);
			else if (res == -2) nudge_m2(lh);
			else if (res == -3) nudge_m3(c, lh, cell);
			else if (res < -res_setting) mark_m_large(c, lh, cell, res);
#undef MARK
#undef SNAP
		}
}

/* Y23: remaining samples -> small codes, plus LH1 nudges (:1329-1420) */
DEV void code_residuals(Ctx *c, int res_setting)
{
	int16_t *p = c->proc, *o = c->ll1;
	const int q = c->q;
	int r, j;
	for (r = 0; r < H; r++)
		for (j = 0; j < H; j++) {
			int16_t *cell = o + r * H + j;
			int16_t *v = p + j * W + H + r;
			if (*cell < 12000) {
				const int res = p[r * W + j] - *cell;
				*cell = 0;
				if (!res || res == 1) { if (v[0] == -7 || v[0] == -8) { if (v[-1] < 2 && v[-1] > -8) v[0] = -9; } }
				else if (res == 2) {
					if (v[0] > 15 && !(v[0] & 7)) v[0]--;
					else if (v[0] == -7 || v[0] == -8) { if (v[-1] <= 1) v[0] = -9; }
					else if (v[0] == -6) { if (v[-1] <= -1 && v[-1] > -8) v[0] = -9; }
				}
				else if (res == 3) {
					if (q >= 21) *cell = 144;
					else if (v[0] > 15 && !(v[0] & 7)) v[0]--;
					else if (v[0] <= 0 && (((-v[0]) + 2) & 0xFFFC) == 8) { if (v[-1] <= 2) v[0] = -10; }
				}
				else if (res > res_setting) {
					*cell = 141;
					if (res == 4) { if (v[0] == 7 || (v[0] & 0xFFFE) == 8) { if (v[-1] >= 0 && v[-1] < 8) v[0] += 2; } }
					else if (res > 6) {
						if (res > 7 && q >= 21) *cell = 148;
						else if (v[0] > 15 && !(v[0] & 7)) v[0]--;
						else if (v[0] == -6 || v[0] == -7 || v[0] == -8) { if (v[-1] < 0 && v[-1] > -8) v[0] = -9; }
					}
				}
			} else {
				switch (*cell) {
				case 14000: *cell = 140; break; case 14500: *cell = 145; break;
				case 12200: *cell = 122; break; case 12100: *cell = 121; break;
				case 12300: *cell = 123; break; case 12400: *cell = 124; break;
				case 14100: *cell = 141; break; case 12500: *cell = 125; break;
				case 12600: *cell = 126; break; case 14900: *cell = 149; break;
				default: break;
				}
			}
		}
}

/* Y24: feed the residual codes back into the kept first-order plane, q>=22 (:1426-1496) */
DEV void adjust_first_order(Ctx *c)
{
	int16_t *f = c->first_order;
	int r, j;
	for (r = 0; r < H; r++)
		for (j = 0; j < H - 2; j++) {
			const int code = c->ll1[r * H + j];
			int16_t *t = f + j * H + r;
			switch (code) {
			case 141: t[0] -= 5; break;           case 140: t[0] += 5; break;
			case 144: t[0] -= 3; break;           case 145: t[0] += 3; break;
			case 121: t[0] -= 4; t[1] -= 3; break; case 122: t[0] += 4; t[1] += 3; break;
			case 123: t[0] += 2; t[1] += 2; t[2] += 2; break;
			case 124: t[0] -= 2; t[1] -= 2; t[2] -= 2; break;
			case 126: t[0] += 9; t[1] += 3; break; case 125: t[0] -= 9; t[1] -= 3; break;
			case 148: t[0] -= 8; break;           case 149: t[0] += 8; break;
			default: break;
			}
		}
}

/* Y25: compaction of the code plane into the three position lists (:1498-1887) */
DEV void build_poslists(Ctx *c)
{
	int16_t *o = c->ll1;
	uint8_t *raw = c->raw;
	uint8_t *pay = c->pay;
	int pass;
	for (pass = 0; pass < 3; pass++) {
		int r, j, n = 0, e = 0;
		if (pass == 1 && c->q < 19) continue;
		if (pass == 2 && c->q < 21) continue;
		for (r = 0; r < H; r++)
			for (j = 0; j < H; j++) {
				int16_t *cell = o + r * H + j;
				if (j == H - 2) { cell[0] = 0; cell[1] = 0; raw[n++] = H - 2; j++; continue; }
				if (pass == 0) {
					switch (*cell) {
					case 141: raw[n++] = (uint8_t)j; *cell = 0;   pay[e++] = 1; break;
					case 140: raw[n++] = (uint8_t)j; *cell = 0;   pay[e++] = 0; break;
					case 126: raw[n++] = (uint8_t)j; *cell = 122; pay[e++] = 0; break;
					case 125: raw[n++] = (uint8_t)j; *cell = 121; pay[e++] = 1; break;
					case 148: raw[n++] = (uint8_t)j; *cell = 144; pay[e++] = 1; break;
					case 149: raw[n++] = (uint8_t)j; *cell = 145; pay[e++] = 0; break;
					default: break;
					}
				} else if (pass == 1) {
					switch (*cell) {
					case 121: raw[n++] = (uint8_t)j; *cell = 0; pay[e++] = 1; break;
					case 122: raw[n++] = (uint8_t)j; *cell = 0; pay[e++] = 0; break;
					case 123: raw[n++] = (uint8_t)j; *cell = 0; pay[e++] = 2; break;
					case 124: raw[n++] = (uint8_t)j; *cell = 0; pay[e++] = 3; break;
					default: break;
					}
				} else {
					if (*cell == 144) { raw[n++] = (uint8_t)j; *cell = 0; pay[e++] = 1; }
					else if (*cell == 145) { raw[n++] = (uint8_t)j; *cell = 0; pay[e++] = 0; }
				}
			}
		poslist_finish(c, pass == 0 ? &c->res1 : pass == 1 ? &c->res3 : &c->res5, raw, n, pay, e, pass == 1 ? 2 : 1);
	}
}

/* the "ripple" adjustment that follows each of the three detail clean-ups (:1957-1976 etc.) */
DEV void ripple(int16_t *v, int may_look_two_ahead)
{
	const int e = v[0];
	if (iabs(e) <= 6) return;
	if (e >= 8 && (e & 7) < 2) { if (v[1] > 7 && v[1] < 10000) v[1]--; }
	else if (e == -7 && v[1] == 8) v[0] = -8;
	else if (e == 8 && v[1] == -7) v[1] = -8;
	else if (e < -7 && ((-e) & 7) < 2) {
		if (v[1] < -14) {
			if (((-v[1]) & 7) == 7) v[1]++;
			else if (((-v[1]) & 7) < 2 && may_look_two_ahead && v[2] <= 0) v[1]++;
		}
	}
}
DEV int loud_neighbours(const int16_t *v)
{
	return (iabs(v[-1]) + 2 >= 8) + (iabs(v[1]) + 2 >= 8) + (iabs(v[-W]) + 2 >= 8) + (iabs(v[W]) + 2 >= 8);
}

/* Y27: three detail-band clean-ups (:1912-2098) */
DEV void clean_details(Ctx *c)
{
	int16_t *p = c->proc;
	const int q = c->q;
	int r, j, lim, lim2;

	lim = q > 22 ? 8 : 9; lim2 = q > 22 ? 4 : 9;            /* LH1: rows 1..254, cols 257..510 */
	for (r = 1; r < H - 1; r++)
		for (j = H + 1; j < W - 1; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(v[0]) >= DEADZONE - 2) {
				if (iabs(v[0]) < lim2) {
					if (loud_neighbours(v) < 3 && v[0] < lim && v[0] > -lim) { if (v[0] < -6) v[0] = -7; else if (v[0] > 6) v[0] = 7; }
				}
			} else v[0] = 0;
			ripple(v, j < W - 2);
		}

	lim = q > 17 ? 8 : 9; lim2 = q > 22 ? 4 : 9;            /* HL1: rows 256..510, cols 1..255 */
	for (r = H; r < W - 1; r++)
		for (j = 1; j < H; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(v[0]) >= DEADZONE - 2) {
				if (iabs(v[0]) < lim2) {
					const int n = loud_neighbours(v);
					if ((n < 3 && v[0] < lim && v[0] > -lim) || !n) v[0] = (int16_t)(v[0] < 0 ? -7 : 7);
				}
			} else v[0] = 0;
			ripple(v, j < H - 2);
		}

	lim = q > 22 ? 8 : 11;                                   /* HH1: rows 256..510, cols 257..510 */
	for (r = H; r < W - 1; r++)
		for (j = H + 1; j < W - 1; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(v[0]) >= DEADZONE - 1) {
				if (iabs(v[0]) < lim) { if (loud_neighbours(v) < 3) v[0] = (int16_t)(v[0] < 0 ? -7 : 7); }
			} else v[0] = 0;
			ripple(v, j < W - 2);
		}
}

/* Y30 + Y31: serpentine gather into the symbol stream, then the symbol rewrites (:2108-2252) */
DEV void scan_and_rewrite(Ctx *c)
{
	const int16_t *p = c->proc;
	uint8_t *s = c->scan;
	const int n = 4 * Q;
	int strip, r, t, i, run;

	for (strip = 0, t = 0; strip < W / 4; strip++)        /* 128 strips of 4 columns */
		for (r = 0; r < W; r++) {
			const int16_t *row = p + r * W + 4 * strip;
			if (!(r & 1)) { s[t] = (uint8_t)row[0]; s[t + 1] = (uint8_t)row[1]; s[t + 2] = (uint8_t)row[2]; s[t + 3] = (uint8_t)row[3]; }
			else { s[t] = (uint8_t)row[3]; s[t + 1] = (uint8_t)row[2]; s[t + 2] = (uint8_t)row[1]; s[t + 3] = (uint8_t)row[0]; }
			t += 4;
		}

	for (i = 0; i < n - 4; i++) {                         /* :2136-2161 (+-8, 0,0,0, +-8) */
		if (s[i] != 128 && s[i + 1] == 128) {
			if (s[i + 2] == 128) {
				if (s[i + 3] == 128) {
					if (s[i] == 136 && s[i + 4] == 136) { s[i] = 132; s[i + 4] = 201; i += 4; }
					else if (s[i] == 136 && s[i + 4] == 120) { s[i] = 133; s[i + 4] = 201; i += 4; }
					else if (s[i] == 120 && s[i + 4] == 136) { s[i] = 134; s[i + 4] = 201; i += 4; }
					else if (s[i] == 120 && s[i + 4] == 120) { s[i] = 135; s[i + 4] = 201; i += 4; }
					else i += 3;
				} else i += 2;
			} else i++;
		}
	}

	for (i = 0; i < 4; i++) { s[i] = 128; s[n - 4 + i] = 128; }
	c->m->select1 = 0; c->m->select2 = 0;
	for (i = 4; i < n - 4; i++) {                         /* :2166-2219 */
		if (s[i] == 136 || s[i] == 120) {
			const int before4 = s[i - 1] == 128 && s[i - 2] == 128 && s[i - 3] == 128 && s[i - 4] == 128;
			const int pair = (s[i + 1] == 120 || s[i + 1] == 136);
			if (s[i + 2] == 128 && pair && before4) { s[i + 1] = (uint8_t)(s[i + 1] == 120 ? 157 : 159); c->m->select2++; }
			else if (s[i - 1] == 128 && pair && s[i + 2] == 128 && s[i + 3] == 128 && s[i + 4] == 128 && s[i + 5] == 128) {
				s[i + 1] = (uint8_t)(s[i + 1] == 120 ? 157 : 159); c->m->select2++;
			}
			else if (before4 && s[i + 1] == 128) { s[i] = (uint8_t)(s[i] == 136 ? 153 : 155); c->m->select1++; }
			else if (s[i - 1] == 128 && s[i + 1] == 128 && s[i + 2] == 128 && s[i + 3] == 128 && s[i + 4] == 128) {
				s[i] = (uint8_t)(s[i] == 136 ? 153 : 155); c->m->select1++;
			}
		}
	}

	for (i = 0, run = 0; i < n; i++) {                    /* :2222-2252 keep 153/155 off run boundaries */
		while (s[i] == 128 && s[i + 1] == 128) {
			run++;
			if (run > 255) {
				for (t = 0; t < 4; t++) { if (s[i + t] == 153) s[i + t] = 124; else if (s[i + t] == 155) s[i + t] = 123; }
				i--; run = 0;
			} else i++;
		}
		if (run >= 252) { if (s[i + 1] == 153) s[i + 1] = 124; else if (s[i + 1] == 155) s[i + 1] = 123; }
		run = 0;
	}
}





/* ------------------------------------------------------------------------------------------
 * LL2 luma coder (Y_highres_compression)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	Ctx *c;
	const uint8_t *s;  /* LL2 samples (even values), followed by zeros */
	const uint8_t *full; /* the same samples with bit 0 */
	uint8_t *o;        /* staging output */
	int j, mem;
} llc;

/* sample sent outside the DPCM stream: marker + two halved samples; the exact first sample goes
 * to the verbatim list (q>15: compress_pixel.c:564-572) */
DEV int ll_verbatim(llc *k, int i)
{
	k->o[k->j++] = 128;
	k->o[k->j++] = (uint8_t)(128 + (k->s[i] >> 1));
	k->o[k->j++] = (uint8_t)(128 + (k->s[i + 1] >> 1));
	k->c->ll_word[k->mem++] = k->full[i];
	k->c->ll_mem[k->c->m->ll_mem_len++] = (uint16_t)i;
	return i + 1;
}

/* three deltas in three bytes, or verbatim when a delta sits on the range edge (COMP3/COMP4) */
DEV int ll_triple(llc *k, int i, int d0, int d1, int d2)
{
	if (d0 == 64 || d1 == 32 || d2 == 64) return ll_verbatim(k, i);
	d1 >>= 1;
	k->o[k->j++] = 64;
	k->o[k->j++] = (uint8_t)(64 + d0 + (d1 >> 3));
	k->o[k->j++] = (uint8_t)(((d1 & 7) << 5) + (d2 >> 1));
	return i + 2;
}

/* s, full, o may live in LDS (the coder is one serial walk; staging its 16 KiB input and its output in LDS takes it off
 * the global-memory latency chain) */
DEV void ll_code_luma(Ctx *c, const uint8_t *s, const uint8_t *full, uint8_t *o)
{
	const int n = Q >> 2;
	llc k;
	int i, e, runs8 = 0, runs16 = 0, mode;

	/* statistics: runs of >=8 and of 16 equal samples (compress_pixel.c:482-497).  The inner walk
	 * does not test the upper bound (it reads the zeros behind the LL2 samples). */
	for (i = 1, e = 0; i < n; i++) {
		while (s[i] == s[i - 1]) {
			e++;
			if (e < 16) { if (e == 8) runs8++; i++; }
			else { runs16++; break; }
		}
		e = 0;
	}
	runs8 += runs16;
	mode = runs16 > 299 ? 2 : (runs8 > 179 ? 1 : 0);   /* :506-508 */
	c->m->res_low = mode;
	c->m->ll_mem_len = 0;

	k.c = c; k.s = s; k.full = full; k.o = o; k.j = 1; k.mem = 0;
	o[0] = s[0];

	for (i = 1; i < n; i++) {
		int d0 = s[i] - s[i - 1], d1 = s[i + 1] - s[i];
		const int d2ok = iabs(s[i + 2] - s[i + 1]) <= 32 && i < n - 2;
		if (d0 == 0 && d1 == 0) {
			int a = 0, d;
			if (mode == 0) {                                    /* :515-553 */
				if (s[i + 2] == s[i + 1]) a = 1;
				i += a + 2;
				o[k.j] = (uint8_t)(a << 3);
				d = s[i] - s[i - 1];
				if (d == 2) {
					const int f = s[i + 1] - s[i];
					if (f == -2) { o[k.j] += 2; i++; } else if (f == 0) { o[k.j] += 3; i++; } else o[k.j] += 1;
				} else if (d == -2) {
					const int f = s[i + 1] - s[i];
					if (f == 2) { o[k.j] += 4; i++; } else if (f == 0) { o[k.j] += 5; i++; } else o[k.j] += 6;
				} else if (d == 4) o[k.j] += 7;
				else i--;
				k.j++;
			} else if (mode == 1) {                             /* :652-673 */
				while (a < 7 && s[i + a + 2] == s[i + a + 1]) a++;
				i += a + 2;
				o[k.j] = (uint8_t)(a << 2);
				d = s[i] - s[i - 1];
				if (d == 2) o[k.j] += 1; else if (d == -2) o[k.j] += 2; else if (d == 0) o[k.j] += 3; else i--;
				k.j++;
			} else {                                            /* :762-775 */
				while (a < 63 && s[i + a + 2] == s[i + a + 1]) a++;
				i += a + 1;
				o[k.j++] = (uint8_t)a;
			}
		}
		else if (mode == 0 && iabs(d0) <= 6 && iabs(d1) <= 8) { /* :554-599 */
			d0 += 6; d1 += 8;
			if (d0 == 12 || d1 == 16) {
				if (d2ok) i = ll_triple(&k, i, d0 + 26, d1 + 8, s[i + 2] - s[i + 1] + 32);
				else i = ll_verbatim(&k, i);
			} else {
				if (d0 < 8) o[k.j++] = (uint8_t)(32 + (d0 << 2) + (d1 >> 1));
				else if (d0 == 8) o[k.j++] = (uint8_t)(16 + (d1 >> 1));
				else o[k.j++] = (uint8_t)(24 + (d1 >> 1));
				i++;
			}
		}
		else if (mode == 1 && iabs(d0) <= 4 && iabs(d1) <= 8) { /* :674-706 */
			d0 += 4; d1 += 8;
			if (d0 == 8 || d1 == 16) {
				if (d2ok) i = ll_triple(&k, i, d0 + 28, d1 + 8, s[i + 2] - s[i + 1] + 32);
				else i = ll_verbatim(&k, i);
			} else { o[k.j++] = (uint8_t)(32 + (d0 << 2) + (d1 >> 1)); i++; }
		}
		else if (iabs(d0) <= 32 && iabs(d1) <= 16 && d2ok)      /* :600-630 */
			i = ll_triple(&k, i, d0 + 32, d1 + 16, s[i + 2] - s[i + 1] + 32);
		else
			i = ll_verbatim(&k, i);
	}

	/* strip the 64 / 128 markers (and the first halved sample of a verbatim record): :828-866 */
	{
		const int j = k.j;
		uint8_t *tmp = o;     /* in place: the write index never passes the read index; the bytes behind j must read 0 */
		int w = 1;
		for (i = j; i < j + 8; i++) tmp[i] = 0;
		for (i = 1; i < j - 1; i++) {
			if (tmp[i] == 64) { o[w++] = tmp[i + 1]; o[w++] = tmp[i + 2]; i += 2; }
			else if (tmp[i] == 128) { o[w++] = tmp[i + 2]; i += 2; }
			else o[w++] = tmp[i];
		}
		if (i < j) o[w++] = tmp[j - 1];
		c->m->ll_comp_y_len = w;
	}
	c->m->ll_word_len = k.mem;
}

/* ------------------------------------------------------------------------------------------
 * LL2 chroma coder (highres_compression), appended behind the luma stream
 * ------------------------------------------------------------------------------------------ */
DEV void ll_code_chroma(Ctx *c)
{
	uint8_t *s = c->ll_bytes;
	uint8_t *o = c->ll_comp;
	const int lo = Q >> 2, hi = (Q >> 2) + (Q >> 3);
	int i, j, a = 0, wide = 0;

	for (i = lo; i < hi; i++) s[i] &= 252;           /* compress_pixel.c:886 */
	c->m->res_high = c->m->res_low;                         /* :887 */
	j = c->m->ll_comp_y_len;
	o[j++] = s[lo];

	for (i = lo + 1; i < hi; i++) {
		int d0 = s[i] - s[i - 1], d1 = s[i + 1] - s[i];
		if (d0 == 0 && d1 == 0) {                     /* :898-945 run of equal samples, up to 14 */
			while (s[i + a + 2] == s[i + a + 1]) {
				a++;
				if (a < 7) continue;
				wide = 1;
				if (a >= 14) break;
			}
			i += a + 1;
			if (wide) o[j] = (uint8_t)(64 + (7 << 3) + a - 7);
			else {
				int d;
				i++;
				o[j] = (uint8_t)(64 + (a << 3));
				d = s[i] - s[i - 1];
				if (d == 4) {
					if (s[i + 1] - s[i] == -4) {
						if (s[i + 2] - s[i + 1] == 0) { o[j] += 3; i += 2; } else { o[j] += 2; i++; }
					} else o[j] += 1;
				} else if (d == -4) {
					if (s[i + 1] - s[i] == 4) {
						if (s[i + 2] - s[i + 1] == 0) { o[j] += 4; i += 2; } else { o[j] += 5; i++; }
					} else o[j] += 6;
				} else if (d == 8) o[j] += 7;
				else i--;
			}
			a = 0; wide = 0;
			j++;
		}
		else if (iabs(d0) <= 4 && iabs(d1) <= 4) {    /* :946-984 steps of 0/+-4 */
			int code = 0, d2;
			if (!d0 && d1 == 4) code = 0; else if (!d0 && d1 == -4) code = 1;
			else if (d0 == 4 && !d1) code = 2; else if (d0 == -4 && !d1) code = 3;
			else if (d0 == 4 && d1 == 4) code = 4; else if (d0 == 4 && d1 == -4) code = 5;
			else if (d0 == -4 && d1 == 4) code = 6; else if (d0 == -4 && d1 == -4) code = 7;
			d2 = s[i + 2] - s[i + 1];
			if (d2 == 0) { o[j++] = (uint8_t)(128 + 64 + (code << 2)); i += 2; }
			else if (d2 == 4) { o[j++] = (uint8_t)(128 + 64 + (code << 2) + 1); i += 2; }
			else if (d2 == -4) { o[j++] = (uint8_t)(128 + 64 + (code << 2) + 2); i += 2; }
			else if (d2 == 8) { o[j++] = (uint8_t)(128 + 64 + (code << 2) + 3); i += 2; }
			else { o[j++] = (uint8_t)(((d0 + 16) << 1) + ((d1 + 16) >> 2)); i++; }
		}
		else if (iabs(d0) <= 16 && iabs(d1) <= 16) {  /* :985-1003 */
			d0 += 16; d1 += 16;
			if (d0 == 32 || d1 == 32) o[j++] = (uint8_t)(128 + (s[i] >> 2));
			else { o[j++] = (uint8_t)((d0 << 1) + (d1 >> 2)); i++; }
		}
		else o[j++] = (uint8_t)(128 + (s[i] >> 2));    /* :1004-1010 */
	}
	c->m->ch_res_len = j;
}

/* ------------------------------------------------------------------------------------------
 * position-list side streams
 *   raw: column indices per row with a 254 marker closing every row; payload: one symbol per entry
 *   word_mode 1: one bit per payload symbol, res1 sizing (Y+1 bytes)
 *   word_mode 2: two bits per payload symbol
 * ------------------------------------------------------------------------------------------ */
DEV void poslist_finish(Ctx *c, PosList *pl, uint8_t *raw, int raw_len, const uint8_t *payload,
                         int payload_len, int word_mode)
{
	uint8_t *cc = c->cc;
	uint8_t *half;
	int i, kept = 1, n, packed = 1, nb, groups;

	/* drop a row marker when the column index falls across it (nhw_encoder.c:1546-1561) */
	memcpy(cc, raw, (size_t)raw_len);
	for (i = 1; i < raw_len - 1; i++) {
		if (cc[i] == H - 2 && cc[i - 1] != H - 2 && cc[i + 1] != H - 2) { if (cc[i - 1] <= cc[i + 1]) raw[kept++] = cc[i]; }
		else raw[kept++] = cc[i];
	}
	raw[kept++] = cc[raw_len - 1];
	n = kept;
	memcpy(cc, raw, (size_t)n);          /* cc = the pruned list (reference nhw_resN before packing) */

	/* halve, then fuse (small step, small step) into one byte (nhw_encoder.c:1569-1592) */
	half = c->half;
	for (i = 0; i < n; i++) half[i] = cc[i] >> 1;
	pl->list[0] = half[0];
	for (i = 1; i < n - 1; i++) {
		const int s0 = half[i] - half[i - 1];
		if (s0 >= 0 && s0 < 8) {
			const int s1 = half[i + 1] - half[i];
			if (s1 >= 0 && s1 < 16) { pl->list[packed++] = (uint8_t)(128 + (s0 << 4) + s1); i++; }
			else pl->list[packed++] = half[i];
		}
		else pl->list[packed++] = half[i];
	}
	pl->len->list_len = packed;

	/* plane of the dropped low bits, markers excluded (nhw_encoder.c:1594-1615) */
	for (i = 0, nb = 0; i < n; i++) if (cc[i] != H - 2) half[nb++] = cc[i];
	for (i = nb; i < nb + 8; i++) half[i] = 0;
	groups = (nb >> 3) + 1;
	for (i = 0; i < groups; i++) {
		int b, v = 0;
		for (b = 0; b < 8; b++) v = (v << 1) | (half[8 * i + b] & 1);
		pl->bits[i] = (uint8_t)v;
	}
	pl->len->bits_len = groups;

	/* payload symbols (nhw_encoder.c:1620-1631, 1751-1763); symbols behind payload_len read as 0 */
	groups = (payload_len >> 3) + 1;
	pl->len->word_len = 0;
	for (i = 0; i < groups; i++) {
		int b, sym[8];
		for (b = 0; b < 8; b++) sym[b] = (8 * i + b < payload_len) ? payload[8 * i + b] : 0;
		if (word_mode == 2) {
			pl->word[pl->len->word_len++] = (uint8_t)(((sym[0] & 3) << 6) | ((sym[1] & 3) << 4) | ((sym[2] & 3) << 2) | (sym[3] & 3));
			pl->word[pl->len->word_len++] = (uint8_t)(((sym[4] & 3) << 6) | ((sym[5] & 3) << 4) | ((sym[6] & 3) << 2) | (sym[7] & 3));
		} else {
			int v = 0;
			for (b = 0; b < 8; b++) v = (v << 1) | (sym[b] & 1);
			pl->word[pl->len->word_len++] = (uint8_t)v;
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * RLE + VLC packetiser (wavlts2packet)
 * ------------------------------------------------------------------------------------------ */

/* rank -> (length << 24) | code word of the NHW prefix code (format constant; reference encoder/tree.h:58-140) */
__device__ static const uint32_t k_vlc[290] = {
	0x02000000, 0x03000002, 0x03000004, 0x0400000a, 0x0400000b, 0x04000006, 0x04000007, 0x05000018,
	0x05000019, 0x0500001a, 0x06000036, 0x06000037, 0x07000070, 0x07000071, 0x080000e8, 0x080000e9,
	0x080000ea, 0x080000eb, 0x080000ec, 0x080000ed, 0x080000ee, 0x080000ef, 0x080000f0, 0x080000f1,
	0x080000f2, 0x080000f3, 0x090001c8, 0x090001c9, 0x090001ca, 0x090001cb, 0x090001cc, 0x090001cd,
	0x090001ce, 0x090001cf, 0x090001e8, 0x090001e9, 0x090001ea, 0x090001eb, 0x090001ec, 0x090001ed,
	0x090001ee, 0x090001ef, 0x0a0003e8, 0x0a0003e9, 0x0a0003ea, 0x0a0003eb, 0x0a0003ec, 0x0a0003ed,
	0x0a0003ee, 0x0a0003ef, 0x0a0003e4, 0x0a0003e5, 0x0a0003e6, 0x0a0003e7, 0x0b0007c0, 0x0b0007c1,
	0x0b0007e0, 0x0b0007e1, 0x0b0007f0, 0x0b0007f1, 0x0b0007f2, 0x0b0007f3, 0x0b0007f4, 0x0b0007f5,
	0x0b0007f6, 0x0b0007f7, 0x0b0007f8, 0x0b0007f9, 0x0b0007fa, 0x0b0007fb, 0x0b0007fc, 0x0b0007fd,
	0x0b0007fe, 0x0b0007ff, 0x0b0007e8, 0x0b0007e9, 0x0b0007ea, 0x0b0007eb, 0x0b0007ec, 0x0b0007ed,
	0x0b0007ee, 0x0b0007ef, 0x0c000f88, 0x0c000f89, 0x0c000f8a, 0x0c000f8b, 0x0c000f8c, 0x0c000f8d,
	0x0c000f8e, 0x0c000f8f, 0x0c000fc8, 0x0c000fc9, 0x0c000fca, 0x0c000fcb, 0x0c000fcc, 0x0c000fcd,
	0x0c000fce, 0x0c000fcf, 0x0d001f08, 0x0d001f09, 0x0d001f0a, 0x0d001f0b, 0x0e003f10, 0x0e003f11,
	0x0e003f12, 0x0e003f13, 0x0e003f14, 0x0e003f15, 0x0e003f16, 0x0e003f17, 0x1101f0c0, 0x1101f0c1,
	0x1101f0c2, 0x1101f0c3, 0x1101f0c4, 0x1101f0c5, 0x1101f0c6, 0x1101f0c7, 0x1101f0c8, 0x1101f0c9,
	0x1101f0ca, 0x1101f0cb, 0x1101f0cc, 0x1101f0cd, 0x1101f0ce, 0x1101f0cf, 0x1101f0d0, 0x1101f0d1,
	0x1101f0d2, 0x1101f0d3, 0x1101f0d4, 0x1101f0d5, 0x1101f0d6, 0x1101f0d7, 0x1101f0d8, 0x1101f0d9,
	0x1101f0da, 0x1101f0db, 0x1101f0dc, 0x1101f0dd, 0x1101f0de, 0x1101f0df, 0x1101f0e0, 0x1101f0e1,
	0x1101f0e2, 0x1101f0e3, 0x1101f0e4, 0x1101f0e5, 0x1101f0e6, 0x1101f0e7, 0x1101f0e8, 0x1101f0e9,
	0x1101f0ea, 0x1101f0eb, 0x1101f0ec, 0x1101f0ed, 0x1101f0ee, 0x1101f0ef, 0x1101f0f0, 0x1101f0f1,
	0x1101f0f2, 0x1101f0f3, 0x1101f0f4, 0x1101f0f5, 0x1101f0f6, 0x1101f0f7, 0x1101f0f8, 0x1101f0f9,
	0x1101f0fa, 0x1101f0fb, 0x1101f0fc, 0x1101f0fd, 0x1101f0fe, 0x1101f0ff, 0x1101f8c0, 0x1101f8c1,
	0x1101f8c2, 0x1101f8c3, 0x1101f8c4, 0x1101f8c5, 0x1101f8c6, 0x1101f8c7, 0x1101f8c8, 0x1101f8c9,
	0x1101f8ca, 0x1101f8cb, 0x1101f8cc, 0x1101f8cd, 0x1101f8ce, 0x1101f8cf, 0x1101f8d0, 0x1101f8d1,
	0x1101f8d2, 0x1101f8d3, 0x1101f8d4, 0x1101f8d5, 0x1101f8d6, 0x1101f8d7, 0x1101f8d8, 0x1101f8d9,
	0x1101f8da, 0x1101f8db, 0x1101f8dc, 0x1101f8dd, 0x1101f8de, 0x1101f8df, 0x1101f8e0, 0x1101f8e1,
	0x1101f8e2, 0x1101f8e3, 0x1101f8e4, 0x1101f8e5, 0x1101f8e6, 0x1101f8e7, 0x1101f8e8, 0x1101f8e9,
	0x1101f8ea, 0x1101f8eb, 0x1101f8ec, 0x1101f8ed, 0x1203f1dc, 0x1203f1dd, 0x1203f1de, 0x1203f1df,
	0x1203f1e0, 0x1203f1e1, 0x1203f1e2, 0x1203f1e3, 0x1203f1e4, 0x1203f1e5, 0x1203f1e6, 0x1203f1e7,
	0x1307e3d0, 0x1307e3d1, 0x1307e3d2, 0x1307e3d3, 0x1307e3d4, 0x1307e3d5, 0x1307e3d6, 0x1307e3d7,
	0x1307e3d8, 0x1307e3d9, 0x1307e3da, 0x1307e3db, 0x1307e3dc, 0x1307e3dd, 0x1307e3de, 0x1307e3df,
	0x1307e3e0, 0x1307e3e1, 0x1307e3e2, 0x1307e3e3, 0x1307e3e4, 0x1307e3e5, 0x1307e3e6, 0x1307e3e7,
	0x1307e3e8, 0x1307e3e9, 0x1307e3ea, 0x1307e3eb, 0x1307e3ec, 0x1307e3ed, 0x1307e3ee, 0x1307e3ef,
	0x1307e3f0, 0x1307e3f1, 0x1307e3f2, 0x1307e3f3, 0x1307e3f4, 0x1307e3f5, 0x140fc7ec, 0x140fc7ed,
	0x140fc7ee, 0x140fc7ef, 0x140fc7f0, 0x140fc7f1, 0x140fc7f2, 0x140fc7f3, 0x140fc7f4, 0x140fc7f5,
	0x140fc7f6, 0x140fc7f7, 0x140fc7f8, 0x140fc7f9, 0x140fc7fa, 0x140fc7fb, 0x140fc7fc, 0x140fc7fd,
	0x140fc7fe, 0x140fc7ff,
};
typedef struct { uint32_t *w; uint32_t cur; int a, fill; } bitsink;
DEV void put_bits(bitsink *b, uint32_t code, int len)   /* MSB-first into 32-bit words (compress_pixel.c:334-345) */
{
	b->fill += len;
	if (b->fill <= 32) b->cur |= code << (32 - b->fill);
	else {
		const int spill = b->fill - 32;
		b->w[b->a++] = b->cur | (code >> spill);
		b->cur = (code & ((1u << spill) - 1)) << (32 - spill);
		b->fill = spill;
	}
}
DEV void flush_bits(bitsink *b) { b->w[b->a] = b->cur; }

DEV int book_symbol_ok(int v) /* which byte codes can enter the code book (:131-160) */
{
	if (v < 109) return !(v & 1);
	if (v == 112) return 1;
	if (v >= 120 && v < 141) return 1;
	if (v >= 144) return !(v & 3);
	return 0;
}

DEV int pack_part(Ctx *c, int part, bitsink *bs)
{
	uint8_t *d = c->scan;
	const int p1 = part ? 4 * Q : 0, p2 = part ? 6 * Q : 4 * Q;
	int *hist = c->hist, *runs = c->hist + 256;
	unsigned *weight = (unsigned *)(c->hist + 512);
	uint16_t *entry = (uint16_t *)(c->hist + 512 + 354);
	uint8_t *tmp_book = (uint8_t *)(entry + 580);
	int select = part ? 3 : 4, i, j, k, e, zone, top_is_zero;
	uint8_t *s1, *s2;
	int n1 = 0, n2 = 0;

	for (i = 0; i < 256; i++) { hist[i] = 0; runs[i] = 0; }

	/* histogram of symbols and of zero-run lengths (:81-107); a run is split at 254 */
	for (i = p1, e = 1; i < p2 - 1; i++) {
		int is_run = 0;
again:
		if (d[i] == 128) {
			while (i < p2 - 1 && d[i + 1] == 128) {
				e++; is_run = 1;
				if (e > 255) { runs[254]++; e = 1; is_run = 0; goto again; }
				else i++;
			}
		}
		if (is_run) runs[e]++; else hist[d[i]]++;
		e = 1;
	}

	for (;;) {                                               /* L_RATIO, :128-236 */
		unsigned zeros = hist[128] > 0 ? (unsigned)hist[128] : 0; /* isolated zeros (:147-153), then short runs */
		for (j = 2; j < 256; j++) if (runs[j] > 0) zeros += (unsigned)(j * runs[j]);
		for (j = 2; j < select; j++) runs[j] = 0;
		for (j = select; j < 256; j++) if (runs[j] > 0) zeros -= (unsigned)(j * runs[j]);
		hist[128] = (int)zeros;
		k = 0;
		for (j = select; j < 256; j++) if (runs[j] > 0) { entry[k] = (uint16_t)((j << 8) | 128); weight[k++] = (unsigned)runs[j]; }
		for (i = 0; i < 256; i++) if (book_symbol_ok(i) && hist[i] > 0) { entry[k] = (uint16_t)((1 << 8) | i); weight[k++] = (unsigned)hist[i]; }
		if (k <= 354) break;
		if (++select >= 100) return NHW_E_CODEBOOK;
	}

	/* stable descending sort by weight == the reference's adjacent-swap bubble sort (:238-252) */
	for (i = 1; i < k; i++) {
		const uint16_t en = entry[i]; const unsigned wt = weight[i];
		for (j = i; j > 0 && weight[j - 1] < wt; j--) { entry[j] = entry[j - 1]; weight[j] = weight[j - 1]; }
		entry[j] = en; weight[j] = wt;
	}

	for (i = 0; i < k; i++) {                                /* symbol -> rank (:261-266) */
		if ((entry[i] >> 8) == 1) hist[entry[i] & 0xFF] = i; else runs[entry[i] >> 8] = i;
	}
	top_is_zero = (entry[0] == ((1 << 8) | 128));
	if (part == 0 && !top_is_zero && k > 290) return NHW_E_CODEBOOK;   /* :269-271 */
	if (part == 1 && select != 4 && k > 290) return NHW_E_CODEBOOK;
	zone = (part == 0 && select == 4 && top_is_zero);

	s1 = c->s1;
	s2 = c->s2;

	{
		int tag = 0, pos;
		e = 1;
		for (i = p1; i < p2 - 1; i++) {                      /* :280-361 */
			const int px = d[i];
			int have_pos = 0;
			if (px == 153) { if (n1 < S_CAP) s1[n1] = 0; n1++; continue; }
			if (px == 155) { if (n1 < S_CAP) s1[n1] = 1; n1++; continue; }
			if (px == 157) { if (n2 < S_CAP) s2[n2] = 0; n2++; continue; }
			if (px == 159) { if (n2 < S_CAP) s2[n2] = 1; n2++; continue; }
			if (px != 128 && px < 136 && px > 120) {
				pos = (uint16_t)hist[px];
				if (px > 131) i += 4;
				have_pos = 1;
			}
			else if (px == 128) {
				int split = 0;
				while (i < p2 - 1 && d[i + 1] == 128) {
					e++;
					if (e > 255) { e = 254; i--; split = 1; break; }
					else i++;
				}
				if (!split && e > 1 && e < select) { i -= (e - 1); tag = e; e = 1; }
			}
			for (;;) {                                       /* L_JUMP / L_ZE */
				if (!have_pos) pos = (uint16_t)((e == 1) ? hist[px] : runs[e]);
				have_pos = 0;
				if (pos >= 110 && pos < 174 && zone) put_bits(bs, (uint32_t)((1 << 6) | (pos - 110)), 15);
				else {
					if (pos >= 174 && zone) pos -= 64;
					put_bits(bs, k_vlc[pos] & 0xFFFFFF, (int)(k_vlc[pos] >> 24));
				}
				e = 1;
				if (tag > 0) { tag--; if (tag > 0) { i++; continue; } }
				break;
			}
		}
	}

	if (part == 0) {
		int b, w;
		c->m->size_data1 = bs->a + 1;
		c->m->wavelet_type = (select > 4 || !top_is_zero) ? 4 : 0;            /* :367-368 */
		/* sign bits of the isolated +-8 symbols and of the +-8 pairs (:370-398) */
		b = (n1 >> 3) + 1;
		for (i = 0; i < b; i++) { int t, v = 0; for (t = 0; t < 8; t++) v = (v << 1) | ((8 * i + t < n1 ? s1[8 * i + t] : 0) & 1); c->sel_word1[i] = (uint8_t)v; }
		c->m->select1 = b;
		b = (n2 >> 3) + 1;
		for (i = 0; i < b; i++) { int t, v = 0; for (t = 0; t < 8; t++) v = (v << 1) | ((8 * i + t < n2 ? s2[8 * i + t] : 0) & 1); c->sel_word2[i] = (uint8_t)v; }
		c->m->select2 = b;

		/* code book 1: symbols, a run entry is (3, length); de-interleave even/odd positions and
		 * collapse consecutive 3s into (3, count) (:400-424) */
		for (i = 0, e = 0; i < k; i++) {
			if ((entry[i] >> 8) == 1) c->book1[e++] = (uint8_t)(entry[i] & 0xFF);
			else { c->book1[e++] = 3; c->book1[e++] = (uint8_t)(entry[i] >> 8); }
		}
		for (i = 0, b = 0; i < e; i += 2) tmp_book[b++] = c->book1[i];
		for (i = 1; i < e; i += 2) tmp_book[b++] = c->book1[i];
		tmp_book[e] = 0;
		for (i = 0, w = 0, b = 0; i < e; i++) {
			while (tmp_book[i] == 3) { b++; i++; }
			if (b > 0) { c->book1[w++] = 3; c->book1[w++] = (uint8_t)b; b = 0; i--; }
			else c->book1[w++] = tmp_book[i];
		}
		c->m->size_book1 = w;
	} else {
		int b, w;
		c->m->size_data2 = bs->a + 1;
		for (i = 0, e = 0; i < k; i++) {                                   /* :431-459 */
			if ((entry[i] >> 8) == 1) c->book2[e++] = (uint8_t)((entry[i] & 0xFF) | 1);
			else { c->book2[e++] = (uint8_t)(entry[i] & 0xFF); c->book2[e++] = (uint8_t)(entry[i] >> 8); }
		}
		c->m->tree_end = e;
		for (i = 0, b = 0; i < e; i += 2) tmp_book[b++] = c->book2[i];
		for (i = 1; i < e; i += 2) tmp_book[b++] = c->book2[i];
		tmp_book[e] = 0;
		for (i = 0, w = 0, b = 0; i < e; i++) {
			while (tmp_book[i] == 128) { b++; i++; }
			if (b > 0) { c->book2[w++] = 128; c->book2[w++] = (uint8_t)b; b = 0; i--; }
			else c->book2[w++] = tmp_book[i];
		}
		c->m->size_book2 = w;
	}
	return NHW_OK;
}

DEV int packetise(Ctx *c)
{
	bitsink bs;
	uint8_t saved;
	int rc;
	bs.w = c->packet; bs.a = 0; bs.fill = 0; bs.cur = 0;
	saved = c->scan[4 * Q]; c->scan[4 * Q] = 3;               /* sentinel behind the luma part (compress_pixel.c:66) */
	rc = pack_part(c, 0, &bs);
	flush_bits(&bs);
	if (rc) return rc;
	bs.a++; bs.fill = 0; bs.cur = 0;                           /* :464 */
	c->scan[4 * Q] = saved;
	c->scan[6 * Q - 1] = c->scan[6 * Q - 2];                   /* :465 */
	rc = pack_part(c, 1, &bs);
	flush_bits(&bs);
	return rc;
}
/* ---------------------------------------------------------------- q>=22 extras */

/* decoder's view of the quantised LH1 band (rows<256, cols 256..511): image_processing.c:523-556.
 * code -> signed index of the |coef|>127 escape codes (tree.h:142-147): 10,12,14,18.. -> 1..19, 60.. -> -1..-19 */
DEV int big_index(int code)
{
	__device__ static const uint8_t pos[19] = { 10, 12, 14, 18, 20, 22, 26, 28, 30, 34, 36, 38, 42, 44, 46, 50, 52, 54, 58 };
	__device__ static const uint8_t neg[19] = { 60, 62, 66, 68, 70, 74, 76, 78, 82, 84, 86, 90, 92, 94, 98, 100, 102, 106, 108 };
	int k;
	for (k = 0; k < 19; k++) { if (code == pos[k]) return k + 1; if (code == neg[k]) return -(k + 1); }
	return 0;
}
DEV void band_recons(Ctx *c)
{
	const int16_t *p = c->proc;
	int16_t *b = c->band;
	int r, j, t = 0;
	memset(b, 0, sizeof(int16_t) * Q);
	for (r = 0; r < H; r++)
		for (j = 0; j < H; j++) {
			const int a = p[r * W + H + j];
			if (a == 128) { t++; continue; }
			else if (a == 127) { b[t - 1] = 5; b[t] = 6; b[t + 1] = 5; t += 2; j++; }
			else if (a == 129) { b[t - 1] = -5; b[t] = -7; b[t + 1] = -5; t += 2; j++; }
			else if ((a & 7) != 0) {
				/* extra_table is indexed 0..108; other odd codes (121..126 ...) read past it in the
				 * reference; those codes do not occur in this band at q>=22 */
				const int k = (a >= 0 && a < 109) ? big_index(a) : 0;
				b[t++] = (int16_t)(k > 0 ? 123 + (k << 3) : (k << 3) - 123);
			}
			else b[t++] = (int16_t)(a > 128 ? a - 125 : a - 131);
		}
}

/* half synthesis of the kept first-order LL + quantised LH vs the original pass-1 plane: res6,
 * char_res1, qsetting3 (wavelet_filterbank.c:498-707) */
DEV void hq_settings(Ctx *c)
{
	const int q = c->q;
	int16_t *hs = c->hs;
	uint8_t *raw = c->raw;
	uint8_t *pay = c->pay;
	const int thr = q > 22 ? 30 : 34;
	int i, r, j, n = 0, e = 0, nq = 0, nc = 0;
	for (r = 0; r < H; r++) {                                  /* upfilter53I + upfilter53III, :509-513 */
		const int16_t *lo = c->first_order + r * H, *hi = c->band + r * H;
		int16_t *out = hs + r * W;
		int k;
		for (k = 0; k < H; k++) {
			const int ln = k + 1 < H ? lo[k + 1] : lo[k];
			const int hp = k > 0 ? hi[k - 1] : hi[0], hn = k + 1 < H ? hi[k + 1] : hi[k];
			out[2 * k] = (int16_t)((int16_t)(lo[k] << 3) - ((hi[k] + hp) << 1));
			out[2 * k + 1] = (int16_t)((int16_t)((lo[k] + ln) << 2) + (6 * hi[k] - hp - hn));
		}
	}
	for (i = 0; i < 2 * Q; i++) {                              /* :518-541 */
		const int d = c->keep[i] - hs[i];
		if (iabs(d) > thr) {
			if (q > 22 && iabs(d) > 56) hs[i] = (int16_t)(d > 0 ? 32000 : 32500);
			else hs[i] = (int16_t)(d > 0 ? 30000 : 31000);
		}
	}
	if (q > 22) {                                              /* :547-564 */
		for (i = 0; i < 2 * Q; i++) {
			if (hs[i] == 32000) c->qsetting3[nq++] = (uint32_t)(i << 1);
			else if (hs[i] == 32500) c->qsetting3[nq++] = (uint32_t)(i << 1) + 1;
		}
	}
	c->m->qsetting3_len = nq;
	for (r = 0; r < H; r++)                                    /* :571-610 */
		for (j = 0; j < W; j++) {
			const int at = r * W + j;
			if (j == H - 2 || j == W - 2) {
				raw[n++] = H - 2;
				if (j == H - 2) {
					if (hs[at] == 30000) c->char_res1[nc++] = (uint16_t)(r * H);
					else if (hs[at] == 31000) c->char_res1[nc++] = (uint16_t)(r * H + 1);
					if (hs[at + 1] == 30000) c->char_res1[nc++] = (uint16_t)(r * H + 2);
					else if (hs[at + 1] == 31000) c->char_res1[nc++] = (uint16_t)(r * H + 3);
				}
				j++;
			}
			else if (hs[at] == 30000) { raw[n++] = (uint8_t)(j & 255); pay[e++] = 0; }
			else if (hs[at] == 31000) { raw[n++] = (uint8_t)(j & 255); pay[e++] = 1; }
		}
	c->m->char_res1_len = nc;
	poslist_finish(c, &c->res6, raw, n, pay, e, 1);
}

/* ---------------------------------------------------------------- chroma (nhw_encoder.c:2255-2868) */
DEV int mark_free_detail(int16_t *p, int at, int16_t code)
{
	/* first of HL2 / LH2 / HH2 co-located coefficients that is inside the dead zone carries the mark */
	if (iabs(p[at + H / 2]) < 8) { p[at + H / 2] = code; return 1; }
	if (iabs(p[at + Q / 2]) < 8) { p[at + Q / 2] = code; return 1; }
	if (iabs(p[at + Q / 2 + H / 2]) < 8) { p[at + Q / 2 + H / 2] = code; return 1; }
	return 0;
}

/* ---------------------------------------------------------------- container (nhw_encoder.c:3112-3218) */
typedef struct { uint8_t *p; size_t cap, n; int ovf; } sink;
DEV void put(sink *s, const void *d, size_t n) { if (s->n + n > s->cap) { s->ovf = 1; return; } memcpy(s->p + s->n, d, n); s->n += n; }
DEV void put16(sink *s, unsigned v) { uint8_t b[2] = { (uint8_t)v, (uint8_t)(v >> 8) }; put(s, b, 2); }
DEV void put32(sink *s, uint32_t v) { uint8_t b[4] = { (uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24) }; put(s, b, 4); }

DEV size_t container(Ctx *c, uint8_t *out, size_t cap)
{
	sink s = { out, cap, 0, 0 };
	const int q = c->q;
	uint8_t b;
	int i;
	b = (uint8_t)(c->m->res_high + c->m->wavelet_type); put(&s, &b, 1);
	b = (uint8_t)q; put(&s, &b, 1);
	put16(&s, (unsigned)c->m->size_book1); put16(&s, (unsigned)c->m->size_book2);
	put32(&s, (uint32_t)c->m->size_data1); put32(&s, (uint32_t)c->m->size_data2);
	put16(&s, (unsigned)c->m->tree_end); put16(&s, (unsigned)c->m->exw_len);
	if (q > 12) put16(&s, (unsigned)c->res1.len->list_len);
	if (q >= 19) { put16(&s, (unsigned)c->res3.len->list_len); put16(&s, (unsigned)c->res3.len->bits_len); }
	if (q > 17) put16(&s, (unsigned)c->m->res4_len);
	if (q > 12) put16(&s, (unsigned)c->res1.len->bits_len);
	if (q >= 21) { put16(&s, (unsigned)c->res5.len->list_len); put16(&s, (unsigned)c->res5.len->bits_len); }
	if (q > 21) { put32(&s, (uint32_t)c->res6.len->list_len); put16(&s, (unsigned)c->res6.len->bits_len); put16(&s, (unsigned)c->m->char_res1_len); }
	if (q > 22) put16(&s, (unsigned)c->m->qsetting3_len);
	put16(&s, (unsigned)c->m->select1); put16(&s, (unsigned)c->m->select2);
	if (q > 15) put16(&s, (unsigned)c->m->ll_word_len);
	put16(&s, (unsigned)c->m->ch_res_len);

	put(&s, c->book1, (size_t)c->m->size_book1); put(&s, c->book2, (size_t)c->m->size_book2);
	put(&s, c->exw, (size_t)c->m->exw_len);
	if (q > 12) { put(&s, c->res1.list, (size_t)c->res1.len->list_len); put(&s, c->res1.bits, (size_t)c->res1.len->bits_len); put(&s, c->res1.word, (size_t)c->res1.len->word_len); }
	if (q > 17) put(&s, c->res4, (size_t)c->m->res4_len);
	if (q >= 19) { put(&s, c->res3.list, (size_t)c->res3.len->list_len); put(&s, c->res3.bits, (size_t)c->res3.len->bits_len); put(&s, c->res3.word, (size_t)c->res3.len->word_len); }
	if (q >= 21) { put(&s, c->res5.list, (size_t)c->res5.len->list_len); put(&s, c->res5.bits, (size_t)c->res5.len->bits_len); put(&s, c->res5.word, (size_t)c->res5.len->word_len); }
	if (q > 21) {
		put(&s, c->res6.list, (size_t)c->res6.len->list_len); put(&s, c->res6.bits, (size_t)c->res6.len->bits_len); put(&s, c->res6.word, (size_t)c->res6.len->word_len);
		for (i = 0; i < c->m->char_res1_len; i++) put16(&s, c->char_res1[i]);
	}
	if (q > 22) for (i = 0; i < c->m->qsetting3_len; i++) put32(&s, c->qsetting3[i]);
	put(&s, c->sel_word1, (size_t)c->m->select1); put(&s, c->sel_word2, (size_t)c->m->select2);
	if (q > 15) { put(&s, c->res_u64, 2 * H); put(&s, c->res_v64, 2 * H); put(&s, c->ll_word, (size_t)c->m->ll_word_len); }
	put(&s, c->ch_res, (size_t)c->m->ch_res_len);
	for (i = 0; i < c->m->size_data2; i++) put32(&s, c->packet[i]);
	return s.ovf ? 0 : s.n;
}


/* ------------------------------------------------------------------------------------------------
 * phases: the stretches of encode_image between two filterbank passes (the filterbank itself runs as
 * separate data-parallel kernels, nhw_front.hip)
 * ------------------------------------------------------------------------------------------------ */

/* after the first L2 analysis: Y5 + Y6 (nhw_encoder.c:141-179) */
DEV void luma_p1(Ctx *c) { PROF_BEGIN(); tag_l2_details(c); PROF(c, 0); dequant_sim_luma(c, 1); PROF(c, 1); }

/* after the first L2 synthesis: Y8 + Y9 (:183-279) */
DEV void luma_p2(Ctx *c) { PROF_BEGIN(); apply_tags(c); PROF(c, 2); precompensate_ll1(c); PROF(c, 3); }

/* after the second L2 analysis (l2save already holds the L2 plane, Y13): Y14..Y18a (:636-762) */
DEV void luma_p3(Ctx *c)
{
	int r, i;
	for (i = Q >> 2; i < (Q >> 2) + (Q >> 3) + 64; i++) c->ll_bytes[i] = 0;  /* the LL coder reads zeros behind the luma samples */
	PROF_BEGIN();
	if (c->q > 17) tag_res4(c);
	emit_ll2(c);
	PROF(c, 4);
	ll_code_luma(c, c->ll_bytes, c->ll_full, c->ll_comp);
	PROF(c, 5);
	for (r = 0; r < H; r++) memcpy(c->proc + r * W, c->l2save + r * H, sizeof(int16_t) * H);   /* Y17 :749-755 */
	PROF(c, 6);
	dequant_sim_luma(c, 0);
	PROF(c, 7);
}

/* after the second L2 synthesis: Y19..Y31 (:766-2252) */
DEV void luma_p4(Ctx *c)
{
	const int q = c->q;
	int r, j, res_setting;
	PROF_BEGIN();
	if (q > 21) for (r = 0; r < H; r++) memcpy(c->first_order + r * H, c->jpeg + r * W, sizeof(int16_t) * H);   /* Y19 :766-777 */
	if (q < 20) {                                                                                               /* Y20 (:783-801) */
		int16_t *p = c->proc;
		for (r = H; r < W; r++) {
			for (j = 0; j < H; j++) { int16_t *v = p + r * W + j; if (iabs(*v) >= DEADZONE && iabs(*v) < 9) *v = (int16_t)(*v > 0 ? 7 : -7); }
			for (j = H; j < W; j++) { int16_t *v = p + r * W + j; if (iabs(*v) >= DEADZONE && iabs(*v) <= 14) *v = (int16_t)(*v > 0 ? 7 : -7); }
		}
	}
	PROF(c, 8);
	tag_small_runs(c);                                           /* Y21 */
	PROF(c, 9);
	res_setting = q >= 20 ? 3 : (q >= 18 ? 4 : 6);               /* :1075-1078 */
	classify_residuals(c, res_setting);                          /* Y22 */
	PROF(c, 10);
	code_residuals(c, res_setting);                              /* Y23 */
	PROF(c, 11);
	if (q > 21) adjust_first_order(c);                           /* Y24 */
	build_poslists(c);                                           /* Y25 */
	PROF(c, 12);
	{                                                            /* Y26 :1893-1910 */
		int16_t *p = c->proc;
		for (r = 0; r < H; r++)
			for (j = 0; j < H; j++) {
				const int16_t v = c->l2save[r * H + j];
				p[r * W + j] = (r < H / 2 && j < H / 2 && v <= 8000) ? 0 : v;
			}
	}
	PROF(c, 13);
	clean_details(c);                                            /* Y27 */
	PROF(c, 14);
	quantise_luma(c);                                            /* Y28 :2100 */
	PROF(c, 15);
	if (q > 21) { band_recons(c); hq_settings(c); }              /* Y29 :2102-2106 */
	PROF(c, 16);
	for (j = 0; j < 16; j++) c->scan[4 * Q + j] = 0;             /* im_nhw is calloc'ed: the rewrite pass peeks one byte past the luma part */
	scan_and_rewrite(c);                                         /* Y30, Y31 */
	PROF(c, 17);
}

/* chroma (nhw_encoder.c:2255-2868); comp 0 = U, 1 = V */
DEV void chroma_p0(Ctx *c, int comp)
{
	const uint8_t *src = comp ? c->pv : c->pu;
	for (int i = 0; i < Q; i++) c->cjpeg[i] = src[i];
}
DEV void chroma_p2(Ctx *c, int comp) { dequant_sim_chroma(c, 1); }
DEV void chroma_p3(Ctx *c, int comp)                              /* :2316-2336 (U), :2629-2648 (V) */
{
	int16_t *jp = c->cjpeg, *p = c->cproc, *o = c->cll1;
	for (int r = 0; r < H / 2; r++)
		for (int j = 0; j < H / 2; j++) {
			const int e = r * H + j, k = r * (H / 2) + j, d = p[e] - o[k];
			const int nx = p[e + 1] - o[k + 1];
			int step = 0;
			if (d > 10) step = -6; else if (d > 7) step = -3; else if (d > 4) step = -2; else if (d > 3) step = -1;
			else if (d > 2 && (comp ? nx > 0 : nx >= 0)) step = -1;
			else if (d < -10) step = 6; else if (d < -7) step = 3; else if (d < -4) step = 2; else if (d < -3) step = 1;
			else if (d < -2 && (comp ? nx < 0 : nx <= 0)) step = 1;
			jp[e] = (int16_t)(o[k] + step);
		}
}
DEV void chroma_p4(Ctx *c, int comp) { dequant_sim_chroma(c, 0); }
DEV void chroma_p5(Ctx *c, int comp)
{
	int16_t *p = c->cproc, *o = c->cll1;
	const int q = c->q;
	const int res_uv = q > 17 ? 4 : 5;                            /* :2370 */
	int r, j, i, a;
	PROF_BEGIN();
	if (q >= 18) {                                                /* :2372-2427; the reference's LL1 index runs on across rows */
		int k = 0;
		for (r = 0; r < H / 2; r++)
			for (j = 0; j < H / 2; j++, k++) {
				const int at = r * H + j, d = p[at] - o[k];
				if (d > 3 && d < 7) {
					const int d1 = p[at + 1] - o[k + 1];
					if (d1 > 2 && d1 < 7 && mark_free_detail(p, at, 12400)) { j++; k++; continue; }
				}
				else if (d < -3 && d > -7) {
					const int d1 = p[at + 1] - o[k + 1];
					if (d1 < -2 && d1 > -8 && mark_free_detail(p, at, 12600)) { j++; k++; continue; }
				}
				if (iabs(d) > res_uv) {
					if (d > 0) mark_free_detail(p, at, 12900);
					else if (d == -5) { if ((p[at + 1] - o[k + 1]) < 0) mark_free_detail(p, at, 13000); }
					else mark_free_detail(p, at, 13000);
				}
			}
	}
	for (r = 0; r < H / 2; r++) memcpy(p + r * H, c->cl2save + r * (H / 2), sizeof(int16_t) * (H / 2));  /* :2431-2439 */

	c->exw[c->m->exw_len++] = 0; c->exw[c->m->exw_len++] = 0;          /* :2489 (U), :2770 (V) */
	a = comp ? (Q >> 2) + (Q >> 4) : (Q >> 2);
	for (r = 0; r < H / 4; r++)                                   /* :2491-2525 LL2 emission */
		for (j = 0; j < H / 4; j++) {
			int s = p[r * H + j];
			if ((s > 255 || s < 0) && (j > 0 || r > 0)) {
				int mag;
				c->exw[c->m->exw_len++] = (uint8_t)r;
				if (s > 255) { c->exw[c->m->exw_len++] = (uint8_t)(j + 128); mag = s - 255; }
				else { c->exw[c->m->exw_len++] = (uint8_t)j; mag = -s; }
				c->exw[c->m->exw_len++] = (uint8_t)(mag > 255 ? 255 : mag);
				c->ll_bytes[a] = c->ll_bytes[a - 1]; a++;
			} else {
				if (s > 255) s = 255; else if (s < 0) s = 0;
				c->ll_bytes[a++] = (uint8_t)(s & 254);
			}
			p[r * H + j] = 0;
		}
	{                                                             /* bit 1 of every LL2 sample (:2527-2548) */
		uint8_t *dst = comp ? c->res_v64 : c->res_u64;
		const uint8_t *sb = c->ll_bytes + (comp ? 20480 : 16384);
		for (i = 0; i < 16 * H / 8; i++) {
			int b, v = 0;
			for (b = 0; b < 8; b++) v = (v << 1) | ((sb[8 * i + b] >> 1) & 1);
			dst[i] = (uint8_t)v;
		}
	}
	PROF(c, 21);
	quantise_chroma(c);
	PROF(c, 22);
	{                                                             /* serpentine, 32 strips of 8 columns, U even / V odd bytes (:2553-2570) */
		uint8_t *s = c->scan + 4 * Q + comp;
		int strip, t = 0;
		for (strip = 0; strip < H / 8; strip++)
			for (r = 0; r < H; r++) {
				const int16_t *row = p + r * H + 8 * strip;
				for (j = 0; j < 8; j++) s[2 * (t + j)] = (uint8_t)row[(r & 1) ? 7 - j : j];
				t += 8;
			}
	}
}

/* Z1, Z2 and the container (compress_pixel.c:878-1022, 53-469; nhw_encoder.c:3112-3218) */
DEV int final_phase(Ctx *c, uint8_t *out, size_t cap, uint32_t *size)
{
	int rc;
	size_t n;
	PROF_BEGIN();
	ll_code_chroma(c);
	PROF(c, 18);
	rc = packetise(c);
	PROF(c, 19);
	if (rc) { *size = 0; return rc; }
	n = container(c, out, cap);
	PROF(c, 20);
	*size = (uint32_t)n;
	return n ? NHW_OK : -3;
}

} // namespace nhw
#endif
