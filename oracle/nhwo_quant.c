/*
 * nhwo_quant.c -- oracle: the quantiser (offsetY / offsetUV) and the decoder-simulating
 * dequantisers used by the encoder's closed loop (offsetY_recons256 / offsetUV_recons256).
 * TEST INFRASTRUCTURE ONLY (see nhwo.h).  Quality 1..23.
 * Reference: encoder/image_processing.c:108-521, 2600-3353; tables encoder/tree.h:54-55.
 */
#include "nhwo_internal.h"

/* escape codes for |coefficient| > 127 (tree.h:54-55): 10+{0,2,4}+8k and 60+{0,2,6}... as published */
static const uint8_t k_big_pos[19] = { 10, 12, 14, 18, 20, 22, 26, 28, 30, 34, 36, 38, 42, 44, 46, 50, 52, 54, 58 };
static const uint8_t k_big_neg[19] = { 60, 62, 66, 68, 70, 74, 76, 78, 82, 84, 86, 90, 92, 94, 98, 100, 102, 106, 108 };

static inline int odd(int v) { return (v & 1) == 1; }
static inline int in_4_7(int v) { return v > 3 && v <= 7; }
static inline int in_m7_m4(int v) { return v < -3 && v >= -7; }
static inline int is_567(int v) { return v == 5 || v == 6 || v == 7; }
static inline int is_m567(int v) { return v == -5 || v == -6 || v == -7; }
static inline int16_t clear_bit0(int v) { return (int16_t)(v & 0xFFFE); }

/* shared tail of the dequantiser: dead zone, bias by 128, floor the magnitude to a multiple of 8,
 * then the decoder's reconstruction offsets (image_processing.c:3003-3015) */
static inline int dequant_value(int a)
{
	if (a < DEADZONE && a > -DEADZONE) return 0;
	a += 128;
	if (a < 0) a = -((-a) & 0xFFF8); else a &= 0xFFF8;
	return a > 128 ? a - 125 : a - 131;
}

/* triple / vertical-pair pattern marking shared by rows<128 (cols 129..254) and rows 128..254
 * (cols 1..254): image_processing.c:2759-2853 */
static void mark_small_runs(int16_t *p, int16_t *jp, int row0, int row1, int col0)
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
static void mark_pairs(int16_t *p, int row0, int row1, int col0)
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
/* q<=16: negative magnitudes keep their low bits only on a ration: of the 15s in a row every sixth is floored to 8, of the
 * x7 above 22 every fourth (:2938-2989, :357-410); everything else is floored.  Returns the magnitude to carry on with. */
static inline int ration_low_bits(int a, int *n15, int *nx7, int mask)
{
	if (a == 15) { if (!*n15) a &= mask; *n15 = (*n15 + 1) % 6; }
	else if (a > 22 && (a & 7) == 7) { if (!*nx7) a &= mask; *nx7 = (*nx7 + 1) % 4; }
	else a &= mask;
	return a;
}

static void dequant_rows(int16_t *p, int16_t *jp, int row0, int row1, int col0, int part, int q)
{
	int r, j;
	for (r = row0; r < row1; r++) {
		int n15 = 0, nx7 = 0;
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
				if (q <= 16) a = ration_low_bits(a, &n15, &nx7, 0xFFF8);
				else if ((a & 7) < 7) a &= 0xFFF8;
				a = -a;
			}
			else if (a == 8 && j < H - 1 && p[at + 1] == -7) p[at + 1] = -8;
			else if (a > 12 && !part && (a & 7) >= 6) { if (j < H - 1 && p[at + 1] == 7) p[at + 1] = 8; }
			jp[at] = (int16_t)dequant_value(a);
		}
	}
}

/* a8: offsetY_recons256, image_processing.c:2600-3190.  `part` 1 = first closed loop, 0 = second. */
void nhwo_dequant_sim_luma(nhwo_ctx *c, int part)
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
		int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (Q >> 2));
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
		for (i = 0; i < c->ll_mem_len; i++) {
			const int idx = c->ll_mem[i];
			jp[((idx >> 7) << 9) + (idx & 127)] = tmp[idx];
		}
		free(tmp);
	}

	if (q > 16) {                                    /* :2759-2907 */
		mark_small_runs(p, jp, 0, H / 2, H / 2 + 1);
		mark_small_runs(p, jp, H / 2, H - 1, 1);
		if (!part) {
			mark_pairs(p, 0, H / 2, H / 2);
			mark_pairs(p, H / 2, H, 0);
		}
	}
	dequant_rows(p, jp, 0, H / 2, H / 2, part, q);
	dequant_rows(p, jp, H / 2, H, 0, part, q);

	if (!part) {                                     /* :3135-3188 isolated coefficient shrink; q<=16 lets diagonal neighbours up to 15 pass */
		const int diag = q <= 16 ? 16 : 8;
		for (r = 1; r < H - 1; r++)
			for (j = 1; j < H - 1; j++) {
				const int e = r * W + j;
				if (iabs(jp[e]) >= 8) {
					if (iabs(jp[e - W - 1]) >= diag || iabs(jp[e - W]) >= 8 || iabs(jp[e - W + 1]) >= diag ||
					    iabs(jp[e - 1]) >= 8 || iabs(jp[e + 1]) >= 8 ||
					    iabs(jp[e + W - 1]) >= diag || iabs(jp[e + W]) >= 8 || iabs(jp[e + W + 1]) >= diag) continue;
					if (r >= H / 2 || j >= H / 2) { if (jp[e] > 0) jp[e]--; else jp[e]++; }
				}
			}
	}
}

/* offsetUV_recons256, image_processing.c:3192-3353 (q>15 form of the LL part) */
static void dequant_rows_chroma(int16_t *p, int16_t *jp, int row0, int row1, int col0, int comp)
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

void nhwo_dequant_sim_chroma(nhwo_ctx *c, int comp)
{
	int16_t *p = c->cproc, *jp = c->cjpeg;
	int r, j;
	for (r = 0; r < H / 4; r++)
		for (j = 0; j < H / 4; j++) {
			const int i = r * H + j;
			if (comp && c->q <= 15) jp[i] = (int16_t)((p[i] & 0xFFFC) + 1);   /* :3221-3230 two low bits dropped, midpoint */
			else if (comp) {                         /* :3198-3219 alternate which sample of a pair keeps bit 0 */
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

static inline int big_code(int a, const uint8_t *tab)
{
	int k = ((a & 0xFFF8) - 128) >> 3;
	return tab[k > 18 ? 18 : k];
}

/* a10: offsetY, image_processing.c:185-521 (q>16 branches) */
void nhwo_quantise_luma(nhwo_ctx *c)
{
	int16_t *p = c->proc;
	const int low = c->q <= 16;
	int i, r, j;
	int n15 = 0, nx7 = 0;        /* quant, quant6: per row */
	int pair_turn = 0;           /* quant4: runs through the whole plane */

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

	for (r = 0; r < (low ? 0 : H); r++)              /* :241-284 (q>16) */
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
	for (r = 0; r < (low ? 0 : H); r++)              /* :286-311 (q>16) */
		for (j = 0; j < H - 1; j++) {
			const int a = r * W + j;
			if (is_567(p[a])) { if (is_567(p[a + 1])) { p[a] = 10300; j++; } }
			else if (is_m567(p[a])) { if (is_m567(p[a + 1])) { p[a] = 10204; j++; } }
		}

	for (i = 0; i < 4 * Q; i++) {                    /* :314-519 */
		const int col = i & (W - 1);
		int a = p[i];
		if (!col) { n15 = 0; nx7 = 0; }
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
			if (low) a = ration_low_bits(a, &n15, &nx7, 504);
			else if ((a & 7) < 7) a &= 504;
			a = -a;
		}
		else if (a == 8 && p[i + 1] == -7 && col < W - 1) p[i + 1] = -8;
		else if (a > 12 && (a & 7) >= 6) { if (col < W - 1 && p[i + 1] == 7) p[i + 1] = 9; }

		/* q<=16: two neighbours that both sit on x6/x7 in a detail band: every third such pair is pushed apart by 2 so
		 * that one of them reaches the next quantisation step, unless a negative neighbour on that side forbids it (:427-510) */
		if (low && a >= 14 && p[i + 1] >= 14 && (i >= 2 * Q || col >= H)) {
			const int nx = p[i + 1];
			if (((a & 510) & 7) == 6 && ((nx & 510) & 7) == 6 && ((a & 1) || (nx & 1))) {
				int veto_l = 0, veto_r = 0;
				if (col > 0 && col < W - 2) {
					const int l = p[i - 1], rr = p[i + 2];
					veto_l = (l < -2 && l > -8) || (l < -7 && ((-l) & 7) >= 6);
					veto_r = (rr < -2 && rr > -8) || (rr < -7 && ((-rr) & 7) >= 6);
				}
				if (!pair_turn) {
					int push_left;
					if ((a & 504) == (nx & 504)) push_left = a >= nx; else push_left = a <= nx;
					if (push_left) { if (!veto_l) { a += 2; p[i + 1] -= 2; } }
					else { if (!veto_r) p[i + 1] += 2; }
				}
				pair_turn = (pair_turn + 1) % 3;
			}
		}

		if (a < DEADZONE && a > -DEADZONE) p[i] = 128;
		else p[i] = (int16_t)((a + 128) & 248);
	}
}

/* offsetUV, image_processing.c:108-183 */
void nhwo_quantise_chroma(nhwo_ctx *c)
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
