/*
 * nhwo_luma.c -- oracle: the luma half of encode_image, quality 1..23.
 * TEST INFRASTRUCTURE ONLY (see nhwo.h).  Reference: encoder/nhw_encoder.c:103-2252.
 * Pass ids (Y2..Y31) are those of SURVEY.md Appendix A.
 */
#include "nhwo_internal.h"

static inline int odd(int v) { return (v & 1) == 1; }
static inline int in_4_7(int v) { return v > 3 && v <= 7; }
static inline int in_m7_m4(int v) { return v < -3 && v >= -7; }
static inline int mult8_or_7(int m) { return !(m & 7) || (m & 7) == 7; } /* on a magnitude */

/* Y5: mark L2 detail coefficients whose quantisation error sign is predictable (nhw_encoder.c:144-177) */
static void tag_l2_details(nhwo_ctx *c)
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
static void apply_tags(nhwo_ctx *c)
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

static inline int big_step(int d) /* correction for a large closed-loop error (:225-232) */
{
	if (d > 11) return -7; if (d > 7) return -4; if (d > 5) return -2; if (d > 4) return -1;
	if (d < -11) return 7; if (d < -7) return 4; if (d < -5) return 2; if (d < -4) return 1;
	return 0;
}

/* Y9: LL1 pre-compensation, strictly left to right (:218-279) */
static void precompensate_ll1(nhwo_ctx *c)
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
static void tag_res4(nhwo_ctx *c)
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
	c->res4_len = n;
	c->res4 = (uint8_t *)arena_get(&c->arena, (size_t)n + 8);
}

/* Y15: LL2 emission (:661-741) */
static void emit_ll2(nhwo_ctx *c)
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
	c->exw_len = e;
}

/* Y21: +-5..7 run tagging (:970-1073) */
static void tag_small_runs(nhwo_ctx *c)
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
static void nudge_up_small(int16_t *v)                       /* L_W1 (:1251-1262) */
{
	if (v[0] == 7) { if (v[-1] >= 0 && v[-1] < 8) v[0] += 2; }
	else if (v[0] == 8) { if (v[-1] >= -2 && v[-1] < 8) v[0] += 2; }
}
static void nudge_m2(int16_t *v)                             /* L_W2 (:1264-1275) */
{
	if (v[0] < -14) { if (mult8_or_7(-v[0])) v[0]++; }
	else if (v[0] == 7 || (v[0] & 0xFFFE) == 8) { if (v[-1] >= -2) v[0] += 3; }
}
static void nudge_m3(nhwo_ctx *c, int16_t *v, int16_t *cell)  /* L_W3 (:1277-1294) */
{
	if (c->q >= 21) *cell = 14500;
	else if (v[0] < -14) { if (mult8_or_7(-v[0])) v[0]++; }
	else if (v[0] >= 0 && ((v[0] + 2) & 0xFFFC) == 8) { if (v[-1] >= -2) v[0] = 10; }
	else if (v[0] > 14 && (v[0] & 7) == 7) v[0]++;
}
static void mark_m_large(nhwo_ctx *c, int16_t *v, int16_t *cell, int res) /* L_W5 (:1296-1325) */
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
static void classify_residuals(nhwo_ctx *c, int res_setting)
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
static void code_residuals(nhwo_ctx *c, int res_setting)
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
static void adjust_first_order(nhwo_ctx *c)
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
static void build_poslists(nhwo_ctx *c)
{
	int16_t *o = c->ll1;
	uint8_t *raw = (uint8_t *)calloc(Q + 64, 1);   /* reference: 96*H+1 bytes; sized for the worst case here */
	uint8_t *pay = (uint8_t *)calloc(Q + 64, 1);
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
		nhwo_poslist_finish(c, pass == 0 ? &c->res1 : pass == 1 ? &c->res3 : &c->res5, raw, n, pay, e, pass == 1 ? 2 : 1);
	}
	free(pay); free(raw);
}

/* the "ripple" adjustment that follows each of the three detail clean-ups (:1957-1976 etc.) */
static inline void ripple(int16_t *v, int may_look_two_ahead)
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
static inline int loud_neighbours(const int16_t *v)
{
	return (iabs(v[-1]) + 2 >= 8) + (iabs(v[1]) + 2 >= 8) + (iabs(v[-W]) + 2 >= 8) + (iabs(v[W]) + 2 >= 8);
}

/* Y27: three detail-band clean-ups (:1912-2098) */
static void clean_details(nhwo_ctx *c)
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
static void scan_and_rewrite(nhwo_ctx *c)
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
	c->select1 = 0; c->select2 = 0;
	for (i = 4; i < n - 4; i++) {                         /* :2166-2219 */
		if (s[i] == 136 || s[i] == 120) {
			const int before4 = s[i - 1] == 128 && s[i - 2] == 128 && s[i - 3] == 128 && s[i - 4] == 128;
			const int pair = (s[i + 1] == 120 || s[i + 1] == 136);
			if (s[i + 2] == 128 && pair && before4) { s[i + 1] = (uint8_t)(s[i + 1] == 120 ? 157 : 159); c->select2++; }
			else if (s[i - 1] == 128 && pair && s[i + 2] == 128 && s[i + 3] == 128 && s[i + 4] == 128 && s[i + 5] == 128) {
				s[i + 1] = (uint8_t)(s[i + 1] == 120 ? 157 : 159); c->select2++;
			}
			else if (before4 && s[i + 1] == 128) { s[i] = (uint8_t)(s[i] == 136 ? 153 : 155); c->select1++; }
			else if (s[i - 1] == 128 && s[i + 1] == 128 && s[i + 2] == 128 && s[i + 3] == 128 && s[i + 4] == 128) {
				s[i] = (uint8_t)(s[i] == 136 ? 153 : 155); c->select1++;
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

/* Y11: isolated just-above-dead-zone coefficients of the level-2 bands of rows 128..255 (:285-309), q<=11 */
static void kill_isolated_l2(nhwo_ctx *c)
{
	int16_t *p = c->proc;
	const int lim = c->q > 6 ? 10 : 11;
	int r, j;
	for (r = H / 2; r < H; r++)
		for (j = 0; j < H; j++) {
			int16_t *v = p + r * W + j;
			const int m = iabs(v[0]);
			if (m >= DEADZONE && m < lim) {
				const int ql = iabs(v[-1]) < DEADZONE, qr = iabs(v[1]) < DEADZONE;
				if (ql && qr) v[0] = 0;
				else if (m == DEADZONE && (ql || qr)) v[0] = 0;
			}
		}
}

static inline void zero_if_below(int16_t *v, int lim) { if (iabs(*v) < lim) *v = 0; }

/* the 2x2 level-1 coefficients of all three level-1 bands that sit under the LL2 cell with flat index `cell` */
static void clear_l1_children(int16_t *p, int cell, int lim_a, int lim_b, int lim_c)
{
	const int base = cell << 1;
	const int band[3] = { H, 2 * Q, 2 * Q + H };
	const int lim[3] = { lim_a, lim_b, lim_c };
	int b;
	for (b = 0; b < 3; b++) {
		int16_t *v = p + base + band[b];
		zero_if_below(v, lim[b]); zero_if_below(v + 1, lim[b]); zero_if_below(v + W, lim[b]); zero_if_below(v + W + 1, lim[b]);
	}
}
/* the level-2 detail coefficients at the position of the LL2 cell (q<=11) */
static void clear_l2_siblings(int16_t *p, int cell)
{
	zero_if_below(p + cell + H / 2, 11); zero_if_below(p + cell + Q, 12); zero_if_below(p + cell + Q + H / 2, 13);
}

/* Y12: LL2 smoothing with zeroing of the coefficients under smoothed cells, q<=12 (:311-621) */
static void smooth_ll2(nhwo_ctx *c)
{
	/* wvlt_thrx1..7 by quality (:313-381) */
	static const uint8_t thr_by_q[13][7] = {
		{ 0 }, { 11, 15, 10, 15, 36, 20, 21 }, { 11, 15, 10, 15, 36, 19, 20 }, { 11, 15, 10, 15, 36, 18, 18 },
		{ 11, 15, 10, 15, 36, 17, 17 }, { 11, 15, 10, 15, 36, 17, 17 }, { 11, 15, 10, 15, 36, 17, 17 },
		{ 10, 15, 9, 14, 36, 17, 17 }, { 8, 13, 6, 11, 34, 15, 15 }, { 8, 13, 6, 11, 34, 15, 15 },
		{ 8, 13, 6, 11, 34, 15, 15 }, { 8, 13, 6, 11, 34, 15, 15 }, { 8, 13, 6, 11, 34, 14, 0 } };
	int16_t *p = c->proc;
	const int q = c->q, deep = q <= 11;
	const uint8_t *t = thr_by_q[q];
	const int t1 = t[0], t2 = t[1], t3 = t[2], t4 = t[3], t5 = t[4], t6 = t[5], t7 = t[6];
	/* the reference's `count` variable: loop counter of the inner loops of the first walk, cell index in the others;
	 * the third walk uses it without having set it when its innermost test fails (:571-579).  It enters this pass as
	 * IM_SIZE, left there by the copy loops above (:129-135, :218). */
	int last = Q;
	int r, j, k;

	for (r = 0; r < H / 2; r++)                                  /* five cells in a row (:383-486) */
		for (j = 0; j < H / 2 - 4; j++) {
			int16_t *v = p + r * W + j;
			int hit = 0;
			if (iabs(v[4] - v[0]) < t1 && iabs(v[4] - v[3]) < t1 && iabs(v[1] - v[0]) < t1 &&
			    iabs(v[3] - v[1]) < t1 && iabs(v[3] - v[2]) < t2 - 2) {
				if ((v[3] - v[1]) > 5 && (v[2] - v[3]) >= 0) v[2] = v[3];
				else if ((v[1] - v[3]) > 5 && (v[2] - v[3]) <= 0) v[2] = v[3];
				else if ((v[1] - v[3]) > 5 && (v[2] - v[1]) >= 0) v[2] = v[1];
				else if ((v[3] - v[1]) > 5 && (v[2] - v[1]) <= 0) v[2] = v[1];
				else if ((v[3] - v[2]) > 0 && (v[2] - v[1]) > 0) { }
				else if ((v[1] - v[2]) > 0 && (v[2] - v[3]) > 0) { }
				else v[2] = (int16_t)((v[3] + v[1]) >> 1);
				hit = 1;
			}
			else if (iabs(v[4] - v[0]) < t2 + 1 && iabs(v[4] - v[3]) < t2 + 1 && iabs(v[1] - v[0]) < t2 + 1) {
				if (iabs(v[3] - v[1]) < t2 + 6 && iabs(v[3] - v[2]) < t2 + 6) {
					const int up = v[3] - v[2], dn = v[2] - v[1];
					if ((up >= 0 && dn >= 0) || (up <= 0 && dn <= 0)) hit = 1;
				}
			}
			if (hit) {
				for (k = 1; k < 4; k++) clear_l1_children(p, r * W + j + k, t6, t6 + 6, t5);
				if (deep) for (k = 1; k < 4; k++) clear_l2_siblings(p, r * W + j + k);
				last = 4;
			}
		}

	for (r = 0; r < H / 2 - 2; r++)                              /* centre of a plus shape, rounding +2 (:488-533) */
		for (j = 0; j < H / 2 - 2; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(v[1] - v[2 * W + 1]) < t3 && iabs(v[W] - v[W + 2]) < t3 &&
			    iabs(v[W + 1] - v[W]) < t4 - 1 && iabs(v[1] - v[W + 1]) < t4) {
				const int e = (v[1] + v[2 * W + 1] + v[W] + v[W + 2] + 2) >> 2;
				if (iabs(e - v[W]) < 5 || iabs(e - v[W + 2]) < 5) v[W + 1] = (int16_t)e;
				last = r * W + j + W + 1;
				clear_l1_children(p, last, t6, t6 + 6, 32);
				if (deep) for (k = 0; k < 3; k++) clear_l2_siblings(p, last + k - 1);
			}
		}

	for (r = 0; r < H / 2 - 2; r++)                              /* flat corner, rounding +1 (:535-583) */
		for (j = 0; j < H / 2 - 2; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(v[2] - v[1]) < t3 && iabs(v[1] - v[0]) < t3 && iabs(v[0] - v[W]) < t3 && iabs(v[2] - v[W + 2]) < t3) {
				if (iabs(v[2 * W + 1] - v[W]) < t3 && iabs(v[W] - v[W + 1]) < t4) {
					const int e = (v[1] + v[2 * W + 1] + v[W] + v[W + 2] + 1) >> 2;
					if (iabs(e - v[W]) < 5 || iabs(e - v[W + 2]) < 5) v[W + 1] = (int16_t)e;
					last = r * W + j + W + 1;
					clear_l1_children(p, last, t6, t6 + 6, 32);
				}
				if (deep) for (k = 0; k < 3; k++) clear_l2_siblings(p, last + k - 1);
			}
		}

	if (deep)                                                    /* three flat cells in a row (:585-620) */
		for (r = 0; r < H / 2; r++)
			for (j = 0; j < H / 2 - 2; j++) {
				const int16_t *v = p + r * W + j;
				if (iabs(v[2] - v[1]) < t7 && iabs(v[2] - v[0]) < t7 && iabs(v[1] - v[0]) < t7) {
					clear_l1_children(p, r * W + j + 1, t6, t6 + 6, 34);
					clear_l2_siblings(p, r * W + j + 1);
				}
			}
}

/* zeroing rule shared by the three bands of Y20 at q<=13: a small coefficient goes when its level-2 parent is small,
 * or together with a neighbour when the pair nearly cancels (:875-886 etc.) */
static inline void thin_by_parent(int16_t *v, int parent, int lim_parent, int lim_pair)
{
	if (iabs(parent) < lim_parent) v[0] = 0;
	else if (iabs(v[0] + v[-1]) < lim_pair && iabs(v[1]) < lim_pair) { v[0] = 0; v[-1] = 0; }
	else if (iabs(v[0] + v[1]) < lim_pair && iabs(v[-1]) < lim_pair) { v[0] = 0; v[1] = 0; }
}
static inline int16_t keep_loud(int v, int q) /* :947-953 */
{
	if (q > 10) return (int16_t)(v >= 16 ? 7 : v <= -16 ? -7 : 0);
	return 0;
}

/* Y20 for q<=15: thresholding of the level-1 bands (:804-968) */
static void thin_l1_low(nhwo_ctx *c)
{
	int16_t *p = c->proc;
	const int16_t *par = c->l2save;
	const int q = c->q;
	int r, j, t1, t2, t3, t4, t5;

	if (q >= 14) {                                               /* :804-832 */
		t1 = 11; t2 = (q == 15) ? 19 : 20;
		for (r = H; r < W; r++) {
			for (j = 0; j < H; j++) { int16_t *v = p + r * W + j; if (iabs(*v) >= DEADZONE && iabs(*v) < t1) *v = 0; }
			for (j = H; j < W; j++) {
				int16_t *v = p + r * W + j;
				if (iabs(*v) >= DEADZONE && iabs(*v) < t2) *v = (int16_t)(*v >= 14 ? 7 : *v <= -14 ? -7 : 0);
			}
		}
		return;
	}
	if (q == 13) { t1 = 15; t2 = 27; t3 = 10; t4 = 6; t5 = 3; }
	else {                                                       /* thresholds follow the number of loud coefficients (:836-868) */
		int loud = 0, i;
		t1 = 16; t2 = 28; t3 = 11; t4 = 8; t5 = 5;
		for (i = 2 * Q; i < 4 * Q; i++) if (iabs(p[i]) >= 12) loud++;
		if (loud > 12500) { t1 = 19; t2 = 31; t3 = 13; t4 = 9; t5 = 6; }
		else if (loud > 10000) { t1 = 18; t2 = 30; t3 = 12; t4 = 8; t5 = 6; }
		else if (loud >= 7000) { t1 = 17; t2 = 29; t3 = 11; t4 = 8; t5 = 5; }
		if (q == 11) { if (loud > 12500) { t1++; t2++; t3++; t4++; t5++; } else t1++; }
		else if (q <= 10) {
			if (loud > 12500) { t1 += 3; t2 += 3; t3 += 2; t4 += 3; t5 += 3; }
			else { t1 += 3; t2 += 2; t3 += 2; t4 += 2; t5 += 2; }
		}
	}

	for (r = 0; r < H; r++)                                      /* rows 0..255, columns 256..511 (:871-896) */
		for (j = H; j < W; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(*v) >= DEADZONE && iabs(*v) < t3 + 2) thin_by_parent(v, par[((r * H + (j - H)) >> 1) + H / 2], t4, t5);
			if (iabs(*v) >= DEADZONE && iabs(*v) < t3) { if (iabs(v[-1]) < DEADZONE && iabs(v[1]) < DEADZONE) *v = 0; }
		}
	for (r = H; r < W; r++) {                                    /* rows 256..511 (:898-967) */
		for (j = 0; j < H; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(*v) >= DEADZONE && iabs(*v) < t1 + 2) thin_by_parent(v, par[(((r - H) * H + j) >> 1) + Q / 2], t4, t5);
			if (iabs(*v) >= DEADZONE && iabs(*v) < t1) {
				if (iabs(v[-1]) < DEADZONE && iabs(v[1]) < DEADZONE) *v = 0;
				else if (iabs(*v) < t1 - 4) *v = 0;
			}
		}
		for (j = H; j < W - 1; j++) {
			int16_t *v = p + r * W + j;
			if (iabs(*v) >= DEADZONE && iabs(*v) < t2 + 1)
				thin_by_parent(v, par[(((r - H) * H + (j - H)) >> 1) + Q / 2 + H / 2], t4 + 1, t5);
			if (iabs(*v) >= DEADZONE && iabs(*v) < t2) {
				if (iabs(v[-1]) < DEADZONE && iabs(v[1]) < DEADZONE) *v = keep_loud(*v, q);
				else if (iabs(*v) < t2 - 5) *v = keep_loud(*v, q);
			}
		}
	}
}

int nhwo_luma(nhwo_ctx *c)
{
	const int q = c->q;
	int r, j, res_setting;

	if (q < 22) { nhwo_prefilter(c->jpeg, q); trace_planes(c, "pre_processing", c->jpeg, 8 * Q, NULL, 0); }   /* :116-119 */

	nhwo_analysis(c->jpeg, c->proc, W, W, 0, q > 21 ? c->keep : NULL);                                         /* Y2 :125 */
	trace_planes(c, "wavelet_analysis_512", c->jpeg, 8 * Q, c->proc, 8 * Q);
	for (r = 0; r < H; r++) memcpy(c->ll1 + r * H, c->jpeg + r * W, sizeof(int16_t) * H);                       /* Y3 :127-135 */
	nhwo_analysis(c->jpeg, c->proc, W, H, 1, NULL);                                                             /* Y4 :139 */
	trace_planes(c, "wavelet_analysis_256", c->jpeg, 8 * Q, c->proc, 8 * Q);

	if (q > 6) {                                                                                                /* first closed loop (:141-283) */
	tag_l2_details(c);
	nhwo_dequant_sim_luma(c, 1);
	trace_planes(c, "offsetY_recons256_p1", c->jpeg, 8 * Q, c->proc, 8 * Q);
	nhwo_synthesis(c->jpeg, c->proc, W, H);
	trace_planes(c, "wavelet_synthesis_256", c->jpeg, 8 * Q, c->proc, 8 * Q);
	apply_tags(c);
	precompensate_ll1(c);
	nhwo_analysis(c->jpeg, c->proc, W, H, 1, NULL);                                                             /* Y10 :281 */
	trace_planes(c, "wavelet_analysis_256", c->jpeg, 8 * Q, c->proc, 8 * Q);
	}
	if (q <= 11) kill_isolated_l2(c);                                                                           /* Y11 */
	if (q <= 12) smooth_ll2(c);                                                                                 /* Y12 */

	for (r = 0; r < H; r++) memcpy(c->l2save + r * H, c->proc + r * W, sizeof(int16_t) * H);                    /* Y13 :623-631 */
	if (nhwo_oob_mode) {                                          /* (values 2 and 3 switch on one half only: debugging aid) */
		/* in the stock one-image-per-process binary res256 is followed by 8 bytes of stale nhw_kernel (q<=21; a fresh
		 * heap, i.e. zeros, above), the size word of the next chunk (0x20011) and resIII itself; the residual passes read up to 430
		 * bytes past res256 */
		extern int16_t nhwo_kernel_row128[4];
		int16_t *tail = c->ll1 + Q;
		int k;
		if (nhwo_oob_mode != 3) {
		for (k = 0; k < 4; k++) tail[k] = q < 22 ? nhwo_kernel_row128[k] : 0;
		tail[4] = 0x0011; tail[5] = 0x0002; tail[6] = 0; tail[7] = 0;
		for (k = 0; k < 504; k++) tail[8 + k] = c->l2save[k];
		}
		/* tree1 is carved out of the freed nhw_kernel block as well (q<=21): what it has not written yet is stale kernel map */
		if (q < 22 && nhwo_oob_mode != 2) { extern int16_t nhwo_kernel_stale[16384]; memcpy(c->ll_bytes, nhwo_kernel_stale, 96 * H + 1); }
	}
	if (q > 17) tag_res4(c);
	emit_ll2(c);
	if (c->trace) {
		const void *bl[3] = { c->ll_bytes, c->ll_full, c->exw };
		const uint32_t ln[3] = { Q >> 2, Q >> 2, (uint32_t)c->exw_len };
		nhwo_trace_put(c->trace, "LL2_emit_Y", 3, bl, ln);
	}
	nhwo_ll_code_luma(c);                                                                                       /* Y16 :745 */
	if (c->trace) {
		uint8_t rl = (uint8_t)c->res_low;
		const void *bl[4] = { c->ll_comp, c->ll_word, c->ll_mem, &rl };
		const uint32_t ln[4] = { (uint32_t)c->ll_comp_y_len, (uint32_t)c->ll_word_len, (uint32_t)c->ll_mem_len * 2, 1 };
		nhwo_trace_put(c->trace, "Y_highres_compression", 4, bl, ln);
	}
	for (r = 0; r < H; r++) memcpy(c->proc + r * W, c->l2save + r * H, sizeof(int16_t) * H);                    /* Y17 :749-755 */

	if (q > 12) {                                                                                               /* second closed loop (:759-779) */
	nhwo_dequant_sim_luma(c, 0);
	trace_planes(c, "offsetY_recons256_p0", c->jpeg, 8 * Q, c->proc, 8 * Q);
	nhwo_synthesis(c->jpeg, c->proc, W, H);
	trace_planes(c, "wavelet_synthesis_256", c->jpeg, 8 * Q, c->proc, 8 * Q);
	if (q > 21) for (r = 0; r < H; r++) memcpy(c->first_order + r * H, c->jpeg + r * W, sizeof(int16_t) * H);   /* Y19 :766-777 */
	}

	if (nhwo_oob_mode && q <= 13) {
		/* the same heap once more: Y20 reads its level-2 parents up to 128 entries behind resIII, which is followed by 8 bytes of stale
		 * kernel map (row 256, columns 8..11), the size word of the next chunk (0x6011) and that chunk, tree1 */
		extern int16_t nhwo_kernel_row256[4];
		int16_t *tail = c->l2save + Q;
		int k;
		for (k = 0; k < 4; k++) tail[k] = nhwo_kernel_row256[k];
		tail[4] = 0x6011; tail[5] = 0; tail[6] = 0; tail[7] = 0;
		for (k = 0; k < 120; k++) tail[8 + k] = (int16_t)(c->ll_bytes[2 * k] | c->ll_bytes[2 * k + 1] << 8);
	}
	if (q <= 15) thin_l1_low(c);                                                                                /* Y20, q<=15 (:804-968) */
	else if (q < 20) {                                                                                               /* Y20, 16<=q<=19 (:783-801) */
		int16_t *p = c->proc;
		for (r = H; r < W; r++) {
			for (j = 0; j < H; j++) { int16_t *v = p + r * W + j; if (iabs(*v) >= DEADZONE && iabs(*v) < 9) *v = (int16_t)(*v > 0 ? 7 : -7); }
			for (j = H; j < W; j++) { int16_t *v = p + r * W + j; if (iabs(*v) >= DEADZONE && iabs(*v) <= 14) *v = (int16_t)(*v > 0 ? 7 : -7); }
		}
	}
	if (q > 16) tag_small_runs(c);                                                                              /* Y21 (:970) */

	res_setting = q >= 20 ? 3 : q >= 18 ? 4 : q >= 15 ? 6 : 8;                                                  /* :1075-1079 */
	if (q > 12) {                                                                                               /* :1081, :1498 */
	classify_residuals(c, res_setting);                                                                         /* Y22 */
	code_residuals(c, res_setting);                                                                             /* Y23 */
	if (q > 21) adjust_first_order(c);                                                                          /* Y24 */
	build_poslists(c);                                                                                          /* Y25 */
	}

	{                                                                                                           /* Y26 :1893-1910 */
		int16_t *p = c->proc;
		for (r = 0; r < H; r++)
			for (j = 0; j < H; j++) {
				const int16_t v = c->l2save[r * H + j];
				p[r * W + j] = (r < H / 2 && j < H / 2 && v <= 8000) ? 0 : v;
			}
	}
	clean_details(c);                                                                                           /* Y27 */
	trace_planes(c, "pre_offsetY", NULL, 0, c->proc, 8 * Q);
	nhwo_quantise_luma(c);                                                                                      /* Y28 :2100 */
	trace_planes(c, "offsetY", NULL, 0, c->proc, 8 * Q);
	if (q > 21) { nhwo_band_recons(c); trace_planes(c, "im_recons_wavelet_band", c->band, 2 * Q, NULL, 0); nhwo_hq_settings(c); }                                                   /* Y29 :2102-2106 */
	scan_and_rewrite(c);
	return NHWO_OK;
}
