/*
 * nhwo_entropy.c -- oracle: serial byte/bit coders of the encode path.  TEST INFRASTRUCTURE ONLY.
 *   - LL2 byte coders            reference encoder/compress_pixel.c:471-876 (luma), 878-1022 (chroma)
 *   - RLE + rank-ordered VLC     reference encoder/compress_pixel.c:53-469, code table encoder/tree.h:58-140
 *   - position-list side streams reference encoder/nhw_encoder.c:1498-1637 (res1), 1641-1768 (res3),
 *                                1772-1887 (res5), encoder/wavelet_filterbank.c:584-704 (res6)
 */
#include "nhwo_internal.h"

/* ------------------------------------------------------------------------------------------
 * LL2 luma coder (Y_highres_compression)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	nhwo_ctx *c;
	const uint8_t *s;  /* LL2 samples (even values), followed by zeros */
	uint8_t *o;        /* staging output */
	int j, mem;
} llc;

/* sample sent outside the DPCM stream: marker + two halved samples; the exact first sample goes
 * to the verbatim list (q>15: compress_pixel.c:564-572) */
static int ll_verbatim(llc *k, int i)
{
	k->o[k->j++] = 128;
	k->o[k->j++] = (uint8_t)(128 + (k->s[i] >> 1));
	if (k->c->q <= 15) return i;                       /* one halved sample, nothing verbatim (:573-577) */
	k->o[k->j++] = (uint8_t)(128 + (k->s[i + 1] >> 1));
	k->c->ll_word[k->mem++] = k->c->ll_full[i];
	k->c->ll_mem[k->c->ll_mem_len++] = (uint16_t)i;
	return i + 1;
}

/* three deltas in three bytes, or verbatim when a delta sits on the range edge (COMP3/COMP4) */
static int ll_triple(llc *k, int i, int d0, int d1, int d2)
{
	if (d0 == 64 || d1 == 32 || d2 == 64) return ll_verbatim(k, i);
	d1 >>= 1;
	k->o[k->j++] = 64;
	k->o[k->j++] = (uint8_t)(64 + d0 + (d1 >> 3));
	k->o[k->j++] = (uint8_t)(((d1 & 7) << 5) + (d2 >> 1));
	return i + 2;
}

void nhwo_ll_code_luma(nhwo_ctx *c)
{
	const uint8_t *s = c->ll_bytes;
	const int n = Q >> 2;
	uint8_t *o = c->ll_comp;
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
	c->res_low = mode;
	c->ll_mem_len = 0;

	k.c = c; k.s = s; k.o = o; k.j = 1; k.mem = 0;
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
		uint8_t *tmp = (uint8_t *)calloc((size_t)j + 8, 1); /* reads one past the end return 0 */
		int w = 1;
		memcpy(tmp, o, (size_t)j);
		for (i = 1; i < j - 1; i++) {
			if (tmp[i] == 64) { o[w++] = tmp[i + 1]; o[w++] = tmp[i + 2]; i += 2; }
			else if (tmp[i] == 128) { if (c->q > 15) { o[w++] = tmp[i + 2]; i += 2; } else { o[w++] = tmp[i + 1]; i++; } }
			else o[w++] = tmp[i];
		}
		if (i < j) o[w++] = tmp[j - 1];
		free(tmp);
		c->ll_comp_y_len = w;
	}
	c->ll_word_len = k.mem;
}

/* ------------------------------------------------------------------------------------------
 * LL2 chroma coder (highres_compression), appended behind the luma stream
 * ------------------------------------------------------------------------------------------ */
void nhwo_ll_code_chroma(nhwo_ctx *c)
{
	uint8_t *s = c->ll_bytes;
	uint8_t *o = c->ll_comp;
	const int lo = Q >> 2, hi = (Q >> 2) + (Q >> 3);
	int i, j, a = 0, wide = 0;

	for (i = lo; i < hi; i++) s[i] &= 252;           /* compress_pixel.c:886 */
	c->res_high = c->res_low;                         /* :887 */
	j = c->ll_comp_y_len;
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
	c->ch_res = o;          /* the reference copies highres_comp into a fresh ch_res (:1015-1017) */
	c->ch_res_len = j;
}

/* ------------------------------------------------------------------------------------------
 * position-list side streams
 *   raw: column indices per row with a 254 marker closing every row; payload: one symbol per entry
 *   word_mode 1: one bit per payload symbol, res1 sizing (Y+1 bytes)
 *   word_mode 2: two bits per payload symbol
 * ------------------------------------------------------------------------------------------ */
void nhwo_poslist_finish(nhwo_ctx *c, nhwo_poslist *pl, uint8_t *raw, int raw_len, const uint8_t *payload,
                         int payload_len, int word_mode)
{
	uint8_t *cc = (uint8_t *)calloc((size_t)raw_len + 16, 1);
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
	half = (uint8_t *)calloc((size_t)n + 16, 1);
	for (i = 0; i < n; i++) half[i] = cc[i] >> 1;
	pl->list = (uint8_t *)arena_get(&c->arena, (size_t)n + 8);
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
	pl->list_len = packed;

	/* plane of the dropped low bits, markers excluded (nhw_encoder.c:1594-1615) */
	for (i = 0, nb = 0; i < n; i++) if (cc[i] != H - 2) half[nb++] = cc[i];
	for (i = nb; i < nb + 8; i++) half[i] = 0;
	groups = (nb >> 3) + 1;
	pl->bits = (uint8_t *)arena_get(&c->arena, (size_t)groups + 8);
	for (i = 0; i < groups; i++) {
		int b, v = 0;
		for (b = 0; b < 8; b++) v = (v << 1) | (half[8 * i + b] & 1);
		pl->bits[i] = (uint8_t)v;
	}
	pl->bits_len = groups;

	/* payload symbols (nhw_encoder.c:1620-1631, 1751-1763); symbols behind payload_len read as 0 */
	groups = (payload_len >> 3) + 1;
	pl->word = (uint8_t *)arena_get(&c->arena, (size_t)groups * 2 + 8);
	pl->word_len = 0;
	for (i = 0; i < groups; i++) {
		int b, sym[8];
		for (b = 0; b < 8; b++) sym[b] = (8 * i + b < payload_len) ? payload[8 * i + b] : 0;
		if (word_mode == 2) {
			pl->word[pl->word_len++] = (uint8_t)(((sym[0] & 3) << 6) | ((sym[1] & 3) << 4) | ((sym[2] & 3) << 2) | (sym[3] & 3));
			pl->word[pl->word_len++] = (uint8_t)(((sym[4] & 3) << 6) | ((sym[5] & 3) << 4) | ((sym[6] & 3) << 2) | (sym[7] & 3));
		} else {
			int v = 0;
			for (b = 0; b < 8; b++) v = (v << 1) | (sym[b] & 1);
			pl->word[pl->word_len++] = (uint8_t)v;
		}
	}
	free(half);
	free(cc);
}

/* ------------------------------------------------------------------------------------------
 * RLE + VLC packetiser (wavlts2packet)
 * ------------------------------------------------------------------------------------------ */

/* rank -> (code, length) of the fixed prefix code, tree.h:58-140, kept as {first code, length, count}
 * runs of consecutive code words */
static const struct { uint32_t first; uint8_t len; uint16_t count; } k_vlc_runs[] = {
	{0x0000,2,1},{0x0002,3,1},{0x0004,3,1},{0x000a,4,2},{0x0006,4,2},{0x0018,5,3},{0x0036,6,2},{0x0070,7,2},
	{0x00e8,8,12},{0x01c8,9,8},{0x01e8,9,8},{0x03e8,10,8},{0x03e4,10,4},{0x07c0,11,2},{0x07e0,11,2},
	{0x07f0,11,16},{0x07e8,11,8},{0x0f88,12,8},{0x0fc8,12,8},{0x1f08,13,4},{0x3f10,14,8},
	{0x1f0c0,17,64},{0x1f8c0,17,46},{0x3f1dc,18,12},{0x7e3d0,19,38},{0xfc7ec,20,20}
};
static uint32_t g_vlc_code[290];
static uint8_t g_vlc_len[290];
static int g_vlc_ready = 0;
static void vlc_init(void)
{
	size_t r; int k = 0, t;
	if (g_vlc_ready) return;
	for (r = 0; r < sizeof k_vlc_runs / sizeof k_vlc_runs[0]; r++)
		for (t = 0; t < k_vlc_runs[r].count; t++, k++) {
			g_vlc_code[k] = k_vlc_runs[r].first + (uint32_t)t;
			g_vlc_len[k] = k_vlc_runs[r].len;
		}
	g_vlc_ready = (k == 290);
}

typedef struct { uint32_t *w; int a, fill; } bitsink;
static inline void put_bits(bitsink *b, uint32_t code, int len)   /* MSB-first into 32-bit words: :334-345 */
{
	b->fill += len;
	if (b->fill <= 32) b->w[b->a] |= code << (32 - b->fill);
	else {
		const int spill = b->fill - 32;
		b->w[b->a] |= code >> spill;
		b->a++;
		b->w[b->a] |= (code & ((1u << spill) - 1)) << (32 - spill);
		b->fill = spill;
	}
}

static inline int book_symbol_ok(int v) /* which byte codes can enter the code book (:131-160) */
{
	if (v < 109) return !(v & 1);
	if (v == 112) return 1;
	if (v >= 120 && v < 141) return 1;
	if (v >= 144) return !(v & 3);
	return 0;
}

static int pack_part(nhwo_ctx *c, int part, bitsink *bs)
{
	uint8_t *d = c->scan;
	const int p1 = part ? 4 * Q : 0, p2 = part ? 6 * Q : 4 * Q;
	int hist[256], runs[256];
	unsigned weight[354];
	uint16_t entry[580];
	/* the reference's `codebook[580]` is one stack array for both parts and is never cleared: when the second part's table ends in
	 * a run of 128s, the collapse below reads on into what the first part left there (its de-interleaved table; behind that, 0 in
	 * the canonical build's stack) */
	uint8_t *tmp_book = c->book_tmp;
	int select = part ? 3 : 4, i, j, k, e, zone, top_is_zero;
	uint8_t *s1, *s2;
	int n1 = 0, n2 = 0;

	memset(hist, 0, sizeof hist); memset(runs, 0, sizeof runs);

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
		if (++select >= 100) return NHWO_E_CODEBOOK;
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
	if (part == 0 && !top_is_zero && k > 290) return NHWO_E_CODEBOOK;   /* :269-271 */
	if (part == 1 && select != 4 && k > 290) return NHWO_E_CODEBOOK;
	zone = (part == 0 && select == 4 && top_is_zero);

	s1 = (uint8_t *)calloc((size_t)c->select1 + 16, 1);
	s2 = (uint8_t *)calloc((size_t)c->select2 + 16, 1);

	{
		int tag = 0, pos;
		e = 1;
		for (i = p1; i < p2 - 1; i++) {                      /* :280-361 */
			const int px = d[i];
			int have_pos = 0;
			if (px == 153) { s1[n1++] = 0; continue; }
			if (px == 155) { s1[n1++] = 1; continue; }
			if (px == 157) { s2[n2++] = 0; continue; }
			if (px == 159) { s2[n2++] = 1; continue; }
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
					put_bits(bs, g_vlc_code[pos], g_vlc_len[pos]);
				}
				e = 1;
				if (tag > 0) { tag--; if (tag > 0) { i++; continue; } }
				break;
			}
		}
	}

	if (part == 0) {
		int b, w;
		c->size_data1 = bs->a + 1;
		c->wavelet_type = (select > 4 || !top_is_zero) ? 4 : 0;            /* :367-368 */
		/* sign bits of the isolated +-8 symbols and of the +-8 pairs (:370-398) */
		b = (n1 >> 3) + 1;
		c->sel_word1 = (uint8_t *)arena_get(&c->arena, (size_t)b + 8);
		for (i = 0; i < b; i++) { int t, v = 0; for (t = 0; t < 8; t++) v = (v << 1) | (s1[8 * i + t] & 1); c->sel_word1[i] = (uint8_t)v; }
		c->select1 = b;
		b = (n2 >> 3) + 1;
		c->sel_word2 = (uint8_t *)arena_get(&c->arena, (size_t)b + 8);
		for (i = 0; i < b; i++) { int t, v = 0; for (t = 0; t < 8; t++) v = (v << 1) | (s2[8 * i + t] & 1); c->sel_word2[i] = (uint8_t)v; }
		c->select2 = b;

		/* code book 1: symbols, a run entry is (3, length); de-interleave even/odd positions and
		 * collapse consecutive 3s into (3, count) (:400-424) */
		for (i = 0, e = 0; i < k; i++) {
			if ((entry[i] >> 8) == 1) c->book1[e++] = (uint8_t)(entry[i] & 0xFF);
			else { c->book1[e++] = 3; c->book1[e++] = (uint8_t)(entry[i] >> 8); }
		}
		for (i = 0, b = 0; i < e; i += 2) tmp_book[b++] = c->book1[i];
		for (i = 1; i < e; i += 2) tmp_book[b++] = c->book1[i];
		memset(tmp_book + e, 0, 600 - (size_t)e);
		for (i = 0, w = 0, b = 0; i < e; i++) {
			while (tmp_book[i] == 3) { b++; i++; }
			if (b > 0) { c->book1[w++] = 3; c->book1[w++] = (uint8_t)b; b = 0; i--; }
			else c->book1[w++] = tmp_book[i];
		}
		c->size_book1 = w;
	} else {
		int b, w;
		c->size_data2 = bs->a + 1;
		for (i = 0, e = 0; i < k; i++) {                                   /* :431-459 */
			if ((entry[i] >> 8) == 1) c->book2[e++] = (uint8_t)((entry[i] & 0xFF) | 1);
			else { c->book2[e++] = (uint8_t)(entry[i] & 0xFF); c->book2[e++] = (uint8_t)(entry[i] >> 8); }
		}
		c->tree_end = e;
		for (i = 0, b = 0; i < e; i += 2) tmp_book[b++] = c->book2[i];
		for (i = 1; i < e; i += 2) tmp_book[b++] = c->book2[i];
		for (i = 0, w = 0, b = 0; i < e; i++) {
			while (tmp_book[i] == 128) { b++; i++; }
			if (b > 0) { c->book2[w++] = 128; c->book2[w++] = (uint8_t)b; b = 0; i--; }
			else c->book2[w++] = tmp_book[i];
		}
		c->size_book2 = w;
	}
	free(s1); free(s2);
	return NHWO_OK;
}

int nhwo_packetise(nhwo_ctx *c)
{
	bitsink bs;
	uint8_t saved;
	int rc;
	vlc_init();
	bs.w = c->packet; bs.a = 0; bs.fill = 0;
	saved = c->scan[4 * Q]; c->scan[4 * Q] = 3;               /* sentinel behind the luma part (:66) */
	rc = pack_part(c, 0, &bs);
	if (rc) return rc;
	bs.a++; bs.fill = 0;                                       /* :464 a++ ; L1 re-runs pack=0 (:278) */
	c->scan[4 * Q] = saved;
	c->scan[6 * Q - 1] = c->scan[6 * Q - 2];                   /* :465 */
	return pack_part(c, 1, &bs);
}
