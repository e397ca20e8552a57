/*
 * nhwo_tail.c -- oracle: q>=22 extras, the chroma half of encode_image, the .nhw container and the
 * public entry point.  TEST INFRASTRUCTURE ONLY (see nhwo.h).
 * Reference: encoder/image_processing.c:523-556; encoder/wavelet_filterbank.c:498-707;
 *            encoder/nhw_encoder.c:2255-2878 (chroma), 3100-3218 (container).
 */
#include "nhwo_internal.h"

/* ---------------------------------------------------------------- trace */
typedef struct { char name[32]; uint32_t nblobs; uint32_t len[5]; } trace_hdr;
void nhwo_trace_put(nhwo_trace *t, const char *name, int nblobs, const void **blobs, const uint32_t *lens)
{
	trace_hdr h;
	size_t need = sizeof h;
	int i;
	if (!t || !t->buf) return;
	memset(&h, 0, sizeof h);
	strncpy(h.name, name, sizeof h.name - 1);
	h.nblobs = (uint32_t)nblobs;
	for (i = 0; i < nblobs; i++) { h.len[i] = lens[i]; need += lens[i]; }
	if (t->len + need > t->cap) return;
	memcpy(t->buf + t->len, &h, sizeof h); t->len += sizeof h;
	for (i = 0; i < nblobs; i++) { if (lens[i]) memcpy(t->buf + t->len, blobs[i], lens[i]); t->len += lens[i]; }
	t->count++;
}

/* ---------------------------------------------------------------- q>=22 extras */

/* decoder's view of the quantised LH1 band (rows<256, cols 256..511): image_processing.c:523-556.
 * code -> signed index of the |coef|>127 escape codes (tree.h:142-147): 10,12,14,18.. -> 1..19, 60.. -> -1..-19 */
static int big_index(int code)
{
	static const uint8_t pos[19] = { 10, 12, 14, 18, 20, 22, 26, 28, 30, 34, 36, 38, 42, 44, 46, 50, 52, 54, 58 };
	static const uint8_t neg[19] = { 60, 62, 66, 68, 70, 74, 76, 78, 82, 84, 86, 90, 92, 94, 98, 100, 102, 106, 108 };
	int k;
	for (k = 0; k < 19; k++) { if (code == pos[k]) return k + 1; if (code == neg[k]) return -(k + 1); }
	return 0;
}
void nhwo_band_recons(nhwo_ctx *c)
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
void nhwo_hq_settings(nhwo_ctx *c)
{
	const int q = c->q;
	int16_t *hs = (int16_t *)calloc(2 * Q + 64, sizeof(int16_t));
	uint8_t *raw = (uint8_t *)calloc(2 * Q + W + 64, 1);
	uint8_t *pay = (uint8_t *)calloc(2 * Q + 64, 1);
	const int thr = q > 22 ? 30 : 34;
	int i, r, j, n = 0, e = 0, nq = 0, nc = 0;

	if (c->trace) {
		const void *bl[2] = { c->first_order, c->keep };
		const uint32_t ln[2] = { 2 * Q, 4 * Q };
		nhwo_trace_put(c->trace, "first_order", 2, bl, ln);
	}
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
	c->qsetting3 = (uint32_t *)arena_get(&c->arena, sizeof(uint32_t) * (2 * Q + 8));
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
	c->qsetting3_len = nq;

	c->char_res1 = (uint16_t *)arena_get(&c->arena, sizeof(uint16_t) * (2 * W + 8));
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
	c->char_res1_len = nc;
	nhwo_poslist_finish(c, &c->res6, raw, n, pay, e, 1);
	if (c->trace) {
		const void *bl[5] = { c->res6.list, c->res6.bits, c->res6.word, c->char_res1, c->qsetting3 };
		const uint32_t ln[5] = { (uint32_t)c->res6.list_len, (uint32_t)c->res6.bits_len, (uint32_t)c->res6.word_len,
		                         (uint32_t)nc * 2, (uint32_t)nq * 4 };
		nhwo_trace_put(c->trace, "hq_settings", 5, bl, ln);
	}
	free(pay); free(raw); free(hs);
}

/* ---------------------------------------------------------------- chroma (nhw_encoder.c:2255-2868) */
static inline int mark_free_detail(int16_t *p, int at, int16_t code)
{
	/* first of HL2 / LH2 / HH2 co-located coefficients that is inside the dead zone carries the mark */
	if (iabs(p[at + H / 2]) < 8) { p[at + H / 2] = code; return 1; }
	if (iabs(p[at + Q / 2]) < 8) { p[at + Q / 2] = code; return 1; }
	if (iabs(p[at + Q / 2 + H / 2]) < 8) { p[at + Q / 2 + H / 2] = code; return 1; }
	return 0;
}

int nhwo_chroma(nhwo_ctx *c, int comp)
{
	int16_t *jp = c->cjpeg, *p = c->cproc, *o = c->cll1;
	const uint8_t *src = comp ? c->pv : c->pu;
	const int q = c->q;
	const int res_uv = q > 17 ? 4 : 5;                          /* :2370 */
	int r, j, i, a;

	if (nhwo_oob_mode == NHWO_OOB_GLIBC_ONESHOT)
		/* the chroma res256 (both planes) is carved out of the freed im_bufferU: the one short the passes below read behind it is that
		 * plane's bytes 32768, 32769 (SURVEY App. D method: malloc trace of the stock binary) */
		o[Q >> 2] = (int16_t)(c->pu[32768] | (c->pu[32769] << 8));
	for (i = 0; i < Q; i++) jp[i] = src[i];                     /* :2256 / :2573 */
	memset(p, 0, sizeof(int16_t) * Q);                          /* U: fresh zero plane; V re-uses it, every cell read later is rewritten first */

	if (q <= 14) { nhwo_prefilter_chroma(jp, q); trace_planes(c, "pre_processing_UV", jp, 2 * Q, NULL, 0); }   /* :2263 / :2579 */
	nhwo_analysis(jp, p, H, H, 0, NULL);
	trace_planes(c, "wavelet_analysis_256", jp, 2 * Q, p, 2 * Q);
	for (r = 0; r < H / 2; r++) memcpy(o + r * (H / 2), jp + r * H, sizeof(int16_t) * (H / 2));   /* :2271-2276 */
	if (q <= 16) {                                              /* level-1 chroma detail below 24 / 32 / 48 goes (:2277-2308, :2590-2621) */
		for (r = 0; r < H; r++)
			for (j = 0; j < H; j++) {
				int16_t *v = p + r * H + j;
				const int lim = r < H / 2 ? (j < H / 2 ? 0 : 24) : (j < H / 2 ? 32 : 48);
				if (iabs(*v) >= DEADZONE && iabs(*v) < lim) *v = 0;
			}
	}
	nhwo_analysis(jp, p, H, H / 2, 1, NULL);
	trace_planes(c, "wavelet_analysis_128", jp, 2 * Q, p, 2 * Q);
	nhwo_dequant_sim_chroma(c, 1);
	trace_planes(c, "offsetUV_recons256_c1", jp, 2 * Q, p, 2 * Q);
	nhwo_synthesis(jp, p, H, H / 2);
	trace_planes(c, "wavelet_synthesis_128", jp, 2 * Q, p, 2 * Q);

	for (r = 0; r < H / 2; r++)                                 /* :2316-2336 (U), :2629-2648 (V) */
		for (j = 0; j < H / 2; j++) {
			const int e = r * H + j, k = r * (H / 2) + j, d = p[e] - o[k];
			const int nx = p[e + 1] - o[k + 1];
			int step = 0;
			if (d > 10) step = -6; else if (d > 7) step = -3; else if (d > 4) step = -2; else if (d > 3) step = -1;
			else if (d > 2 && (comp ? nx > 0 : nx >= 0)) step = -1;
			else if (d < -10) step = 6; else if (d < -7) step = 3; else if (d < -4) step = 2; else if (d < -3) step = 1;
			else if (d < -2 && (comp ? nx < 0 : nx <= 0)) step = 1;
			jp[e] = (int16_t)(o[k] + step);
		}
	nhwo_analysis(jp, p, H, H / 2, 1, NULL);
	trace_planes(c, "wavelet_analysis_128", jp, 2 * Q, p, 2 * Q);

	for (r = 0; r < H / 2; r++) memcpy(c->cl2save + r * (H / 2), p + r * H, sizeof(int16_t) * (H / 2));  /* :2358-2366 */
	nhwo_dequant_sim_chroma(c, 0);
	trace_planes(c, "offsetUV_recons256_c0", jp, 2 * Q, p, 2 * Q);
	nhwo_synthesis(jp, p, H, H / 2);
	trace_planes(c, "wavelet_synthesis_128", jp, 2 * Q, p, 2 * Q);

	if (q >= 18) {                                               /* :2372-2427; the reference's LL1 index runs on across rows */
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

	if (q <= 11) {                                              /* chroma LL2 smoothing, in place in raster order (:2438-2478, :2739-2779) */
		int pass;
		for (pass = 0; pass < 2; pass++)
			for (r = 0; r < H / 4 - 2; r++)
				for (j = 0; j < H / 4 - 2; j++) {
					int16_t *v = p + r * H + j;
					if (!pass) {
						if (iabs(v[1] - v[2 * H + 1]) < 5 && iabs(v[H] - v[H + 2]) < 5 && iabs(v[H + 1] - v[H]) < 7 && iabs(v[1] - v[H + 1]) < 8)
							v[H + 1] = (int16_t)((v[1] + v[2 * H + 1] + v[H] + v[H + 2] + 2) >> 2);
					} else {
						if (iabs(v[2] - v[1]) < 5 && iabs(v[1] - v[0]) < 5 && iabs(v[0] - v[H]) < 5 && iabs(v[2] - v[H + 2]) < 5 &&
						    iabs(v[2 * H + 1] - v[H]) < 5 && iabs(v[H] - v[H + 1]) < 8)
							v[H + 1] = (int16_t)((v[1] + v[2 * H + 1] + v[H] + v[H + 2] + 1) >> 2);
					}
				}
	}

	c->exw[c->exw_len++] = 0; c->exw[c->exw_len++] = 0;                  /* :2489 (U), :2770 (V) */
	a = comp ? (Q >> 2) + (Q >> 4) : (Q >> 2);
	for (r = 0; r < H / 4; r++)                                  /* :2491-2525 LL2 emission */
		for (j = 0; j < H / 4; j++) {
			int s = p[r * H + j];
			if ((s > 255 || s < 0) && (j > 0 || r > 0)) {
				int mag;
				c->exw[c->exw_len++] = (uint8_t)r;
				if (s > 255) { c->exw[c->exw_len++] = (uint8_t)(j + 128); mag = s - 255; }
				else { c->exw[c->exw_len++] = (uint8_t)j; mag = -s; }
				c->exw[c->exw_len++] = (uint8_t)(mag > 255 ? 255 : mag);
				c->ll_bytes[a] = c->ll_bytes[a - 1]; a++;
			} else {
				if (s > 255) s = 255; else if (s < 0) s = 0;
				c->ll_bytes[a++] = (uint8_t)(s & 254);
			}
			p[r * H + j] = 0;
		}
	if (q > 15) {                                                /* bit 1 of every LL2 sample (:2517-2537, :2815-2836) */
		uint8_t *dst = comp ? c->res_v64 : c->res_u64;
		const uint8_t *sb = c->ll_bytes + (comp ? 20480 : 16384);
		for (i = 0; i < 16 * H / 8; i++) {
			int b, v = 0;
			for (b = 0; b < 8; b++) v = (v << 1) | ((sb[8 * i + b] >> 1) & 1);
			dst[i] = (uint8_t)v;
		}
	}
	trace_planes(c, "pre_offsetUV", NULL, 0, p, 2 * Q);
	nhwo_quantise_chroma(c);
	trace_planes(c, "offsetUV", NULL, 0, p, 2 * Q);

	{                                                            /* serpentine, 32 strips of 8 columns, U even / V odd bytes (:2553-2570) */
		uint8_t *s = c->scan + 4 * Q + comp;
		int strip, t = 0;
		for (strip = 0; strip < H / 8; strip++)
			for (r = 0; r < H; r++) {
				const int16_t *row = p + r * H + 8 * strip;
				for (j = 0; j < 8; j++) s[2 * (t + j)] = (uint8_t)row[(r & 1) ? 7 - j : j];
				t += 8;
			}
	}
	return NHWO_OK;
}

/* ---------------------------------------------------------------- container (nhw_encoder.c:3112-3218) */
typedef struct { uint8_t *p; size_t cap, n; int ovf; } sink;
static void put(sink *s, const void *d, size_t n) { if (s->n + n > s->cap) { s->ovf = 1; return; } memcpy(s->p + s->n, d, n); s->n += n; }
static void put16(sink *s, unsigned v) { uint8_t b[2] = { (uint8_t)v, (uint8_t)(v >> 8) }; put(s, b, 2); }
static void put32(sink *s, uint32_t v) { uint8_t b[4] = { (uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24) }; put(s, b, 4); }

size_t nhwo_container(nhwo_ctx *c, uint8_t *out, size_t cap)
{
	sink s = { out, cap, 0, 0 };
	const int q = c->q;
	uint8_t b;
	int i;
	b = (uint8_t)(c->res_high + c->wavelet_type); put(&s, &b, 1);
	b = (uint8_t)q; put(&s, &b, 1);
	put16(&s, (unsigned)c->size_book1); put16(&s, (unsigned)c->size_book2);
	put32(&s, (uint32_t)c->size_data1); put32(&s, (uint32_t)c->size_data2);
	put16(&s, (unsigned)c->tree_end); put16(&s, (unsigned)c->exw_len);
	if (q > 12) put16(&s, (unsigned)c->res1.list_len);
	if (q >= 19) { put16(&s, (unsigned)c->res3.list_len); put16(&s, (unsigned)c->res3.bits_len); }
	if (q > 17) put16(&s, (unsigned)c->res4_len);
	if (q > 12) put16(&s, (unsigned)c->res1.bits_len);
	if (q >= 21) { put16(&s, (unsigned)c->res5.list_len); put16(&s, (unsigned)c->res5.bits_len); }
	if (q > 21) { put32(&s, (uint32_t)c->res6.list_len); put16(&s, (unsigned)c->res6.bits_len); put16(&s, (unsigned)c->char_res1_len); }
	if (q > 22) put16(&s, (unsigned)c->qsetting3_len);
	put16(&s, (unsigned)c->select1); put16(&s, (unsigned)c->select2);
	if (q > 15) put16(&s, (unsigned)c->ll_word_len);
	put16(&s, (unsigned)c->ch_res_len);

	put(&s, c->book1, (size_t)c->size_book1); put(&s, c->book2, (size_t)c->size_book2);
	put(&s, c->exw, (size_t)c->exw_len);
	if (q > 12) { put(&s, c->res1.list, (size_t)c->res1.list_len); put(&s, c->res1.bits, (size_t)c->res1.bits_len); put(&s, c->res1.word, (size_t)c->res1.word_len); }
	if (q > 17) put(&s, c->res4, (size_t)c->res4_len);
	if (q >= 19) { put(&s, c->res3.list, (size_t)c->res3.list_len); put(&s, c->res3.bits, (size_t)c->res3.bits_len); put(&s, c->res3.word, (size_t)c->res3.word_len); }
	if (q >= 21) { put(&s, c->res5.list, (size_t)c->res5.list_len); put(&s, c->res5.bits, (size_t)c->res5.bits_len); put(&s, c->res5.word, (size_t)c->res5.word_len); }
	if (q > 21) {
		put(&s, c->res6.list, (size_t)c->res6.list_len); put(&s, c->res6.bits, (size_t)c->res6.bits_len); put(&s, c->res6.word, (size_t)c->res6.word_len);
		for (i = 0; i < c->char_res1_len; i++) put16(&s, c->char_res1[i]);
	}
	if (q > 22) for (i = 0; i < c->qsetting3_len; i++) put32(&s, c->qsetting3[i]);
	put(&s, c->sel_word1, (size_t)c->select1); put(&s, c->sel_word2, (size_t)c->select2);
	if (q > 15) { put(&s, c->res_u64, 2 * H); put(&s, c->res_v64, 2 * H); put(&s, c->ll_word, (size_t)c->ll_word_len); }
	put(&s, c->ch_res, (size_t)c->ch_res_len);
	for (i = 0; i < c->size_data2; i++) put32(&s, c->packet[i]);
	return s.ovf ? 0 : s.n;
}

/* ---------------------------------------------------------------- public entry point */
int nhwo_quality_supported(int quality) { return quality >= 1 && quality <= 23; }
int nhwo_oob_mode = NHWO_OOB_ZERO;

int nhwo_encode(const uint8_t *bgr, int quality, uint8_t *out, size_t cap, size_t *out_len, nhwo_trace *trace)
{
	nhwo_ctx ctx, *c = &ctx;
	int rc;
	size_t n;
	if (!nhwo_quality_supported(quality)) return NHWO_E_QUALITY;
	memset(c, 0, sizeof *c);
	c->q = quality; c->trace = trace;
	c->arena.cap = 24u << 20;
	c->arena.base = (uint8_t *)calloc(c->arena.cap, 1);
	if (!c->arena.base) return NHWO_E_ALLOC;
#define GET(T, n) ((T *)arena_get(&c->arena, sizeof(T) * (size_t)(n)))
	c->jpeg = GET(int16_t, 4 * Q); c->proc = GET(int16_t, 4 * Q);
	c->cjpeg = GET(int16_t, Q); c->cproc = GET(int16_t, Q);
	c->pu = GET(uint8_t, Q); c->pv = GET(uint8_t, Q);
	c->ll1 = GET(int16_t, Q); c->l2save = GET(int16_t, Q + 256);   /* + what Y20 reads behind it at quality <= 13 (zeros here) */
	c->cll1 = GET(int16_t, Q >> 2); c->cl2save = GET(int16_t, Q >> 2);
	c->keep = GET(int16_t, 2 * Q); c->first_order = GET(int16_t, Q); c->band = GET(int16_t, Q);
	c->scan = GET(uint8_t, 6 * Q);
	c->ll_bytes = GET(uint8_t, 96 * H + 1); c->ll_full = GET(uint8_t, Q >> 2);
	c->exw = GET(uint8_t, 32 * H * 2);
	c->ll_comp = GET(uint8_t, Q >> 1); c->ll_word = GET(uint8_t, Q >> 2); c->ll_mem = GET(uint16_t, Q >> 2);
	c->res_u64 = GET(uint8_t, 2 * H); c->res_v64 = GET(uint8_t, 2 * H);
	c->packet = GET(uint32_t, 80000); c->book1 = GET(uint8_t, 354 * 2); c->book2 = GET(uint8_t, 354 * 2);
#undef GET
	c->res_low = 3;

	nhwo_color(bgr, quality, c->jpeg, c->pu, c->pv);
	if (trace) {
		const void *bl[3] = { c->jpeg, c->pu, c->pv };
		const uint32_t ln[3] = { 8 * Q, Q, Q };
		nhwo_trace_put(trace, "downsample_YUV420", 3, bl, ln);
	}
	rc = nhwo_luma(c);
	if (!rc) rc = nhwo_chroma(c, 0);
	if (!rc) rc = nhwo_chroma(c, 1);
	if (!rc) {
		if (trace) {
			const void *bl[2] = { c->ll_bytes, c->scan };
			const uint32_t ln[2] = { (Q >> 2) + (Q >> 3) + 1, 6 * Q };
			nhwo_trace_put(trace, "pre_highres_compression", 2, bl, ln);
		}
		nhwo_ll_code_chroma(c);
		trace_planes(c, "highres_compression", c->ch_res, (uint32_t)c->ch_res_len, NULL, 0);
		rc = nhwo_packetise(c);
	}
	if (!rc) {
		if (trace) {
			const void *bl[3] = { c->packet, c->book1, c->book2 };
			const uint32_t ln[3] = { (uint32_t)c->size_data2 * 4, (uint32_t)c->size_book1, (uint32_t)c->size_book2 };
			nhwo_trace_put(trace, "wavlts2packet", 3, bl, ln);
		}
		n = nhwo_container(c, out, cap);
		if (!n) rc = NHWO_E_SPACE; else *out_len = n;
	}
	free(c->arena.base);
	return rc;
}
