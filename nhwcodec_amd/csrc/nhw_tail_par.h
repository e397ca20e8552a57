/*
 * nhw_tail_par.h -- workgroup-parallel forms of the order-dependent passes (256 threads = 4 wavefronts per
 * image).  Each pass keeps the reference's result bit for bit; what changes is who walks which chain:
 *   P  pointwise / stencil on read-only data         -> threads stride over elements
 *   R  row-serial, rows independent                   -> one thread per row, serial along the row
 *   C  column-serial (Y22/Y23)                        -> one thread per column, cross-column reads served
 *                                                        from a snapshot taken before the pass
 *   S/G chains that really are serial                 -> thread 0 (to be replaced by skewed wavefronts)
 * The argument why a pass may be re-ordered is written at each function.  Pass ids: SURVEY.md Appendix A.
 */
#ifndef NHW_TAIL_PAR_H
#define NHW_TAIL_PAR_H

#include "nhw_tail_dev.h"

namespace nhw {

#define NT 256
#define BARRIER() __syncthreads()

/* ---------------------------------------------------------------- P passes */

/* Y5 (nhw_encoder.c:144-177): reads proc, writes only its own ll1 cell */
DEV void tag_l2_details_par(Ctx *c, int tid)
{
	const int16_t *p = c->proc;
	for (int idx = tid; idx < Q; idx += NT) {
		const int r = idx >> 8, j = idx & 255;
		if (r < H / 2 && j < H / 2) continue;
		const int at = r * W + j, s = p[at];
		int16_t *cell = c->ll1 + idx;
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

/* Y8 (:183-216): every tagged coefficient owns a distinct target sample */
DEV void apply_tags_par(Ctx *c, int tid)
{
	int16_t *p = c->proc;
	for (int idx = tid; idx < Q; idx += NT) {
		const int r = idx >> 8, j = idx & 255;
		int16_t *cell = c->ll1 + idx;
		int step;
		if (*cell > 14000) { *cell -= 16000; step = 1; }
		else if (*cell > 10000) { *cell -= 12000; step = -1; }
		else continue;
		if (r < H / 2 && j >= H / 2) p[(2 * (j - H / 2) + 1) * W + 2 * r] += step;
		else if (r >= H / 2 && j < H / 2) p[2 * j * W + 2 * (r - H / 2) + 1] += step;
		else if (r >= H / 2 && j >= H / 2) p[(2 * (j - H / 2) + 1) * W + 2 * (r - H / 2) + 1] += step;
	}
}

/* rows x cols block copy between two planes */
DEV void copy_block_par(const int16_t *src, int src_row, int16_t *dst, int dst_row, int rows, int cols, int tid)
{
	const int per = cols >> 2;                       /* 8-byte pieces */
	for (int idx = tid; idx < rows * per; idx += NT) {
		const int r = idx / per, k = idx % per;
		reinterpret_cast<uint2 *>(dst + r * dst_row)[k] = reinterpret_cast<const uint2 *>(src + r * src_row)[k];
	}
}

/* ---------------------------------------------------------------- Y9 (R) */
/* (:218-279) left neighbour is read after its own update, right neighbour before: serial along a row; rows do
 * not interact (column 0 reads proc[r][-1] = the LH1 cell (r-1, 511), which this pass never writes). */
DEV void precompensate_ll1_par(Ctx *c, int tid)
{
	int16_t *p = c->proc, *jp = c->jpeg;
	const int16_t *o = c->ll1;
	const int r = tid;
	for (int j = 0; j < H; j++) {
		const int e = r * W + j, k = r * H + j, d = p[e] - o[k];
		int step = big_step(d);
		if (!step && iabs(d) > 1) {
			int a = p[e + 1] - o[k + 1];
			if (iabs(a) > 4) a += big_step(a);
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

/* ---------------------------------------------------------------- a8 dequantisation simulation */
DEV void mark_pairs_row(int16_t *p, int r, int col0)
{
	for (int j = col0; j < H - 1; j++) {
		const int a = r * W + j;
		if (is_567(p[a])) { if (is_567(p[a + 1])) { p[a] = 15700; j++; } }
		else if (is_m567(p[a])) { if (is_m567(p[a + 1])) { p[a] = 15800; j++; } }
	}
}
DEV void dequant_row(int16_t *p, int16_t *jp, int r, int col0, int part)
{
	for (int j = col0; j < H; j++) {
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

/* offsetY_recons256 (image_processing.c:2600-3190).  Raster-serial pieces (the LL2 walk that bumps the sample
 * below, the triple / vertical-pair marking that writes into the next row) stay on thread 0; everything whose
 * reach is one row runs one row per thread; the isolated-coefficient shrink is pointwise: a coefficient >= 8
 * next to another one >= 8 keeps both from shrinking, so decisions taken on the untouched plane equal the
 * reference's raster-order decisions. */
DEV void dequant_sim_luma_par(Ctx *c, int part, int tid)
{
	int16_t *p = c->proc, *jp = c->jpeg;
	const int q = c->q;

	if (q > 17 && tid < H / 2) {                       /* :2609-2640 four odd LL2 samples in a row (R) */
		const int r = tid;
		for (int j = 0; j < H / 2 - 3; j++) {
			const int a = r * W + j;
			if (odd(p[a]) && odd(p[a + 1]) && odd(p[a + 2]) && odd(p[a + 3]) && iabs(p[a] - p[a + 3]) > 1) {
				if (!part) { p[a] += 16000; p[a + 1] += 16000; p[a + 2] += 16000; p[a + 3] += 16000; }
				else { p[a] += 16000; p[a + 2] += 16000; }
				j += 3;
			}
		}
	}
	BARRIER();
	if (tid == 0) {                                    /* :2642-2695 (G: writes the sample below) */
		for (int r = 0; r < H / 2; r++)
			for (int j = 0; j < H / 2; j++) {
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
	}
	BARRIER();
	if (!part) {                                       /* :2697-2735 (P) */
		int16_t *tmp = c->tmp16;
		for (int idx = tid; idx < Q / 4; idx += NT) {
			const int a = (idx >> 7) * W + (idx & 127);
			if (p[a] < 10000) { tmp[idx] = p[a]; jp[a] = (p[a] >= 0 && p[a] < 256) ? clear_bit0(p[a]) : p[a]; }
			else { p[a] -= 16000; tmp[idx] = p[a]; jp[a] = p[a]; }
		}
		BARRIER();
		const int nm = c->m->ll_mem_len;
		for (int i = tid; i < nm; i += NT) {
			const int idx = c->ll_mem[i];
			jp[((idx >> 7) << 9) + (idx & 127)] = tmp[idx];
		}
		BARRIER();
	}
	if (tid == 0) {                                    /* :2759-2853 (G: marks cells of the next row) */
		mark_small_runs(p, jp, 0, H / 2, H / 2 + 1);
		mark_small_runs(p, jp, H / 2, H - 1, 1);
	}
	BARRIER();
	{
		const int r = tid, col0 = r < H / 2 ? H / 2 : 0;
		if (!part) mark_pairs_row(p, r, col0);           /* :2857-2905 (R) */
		dequant_row(p, jp, r, col0, part);               /* :2909-3124 (R) */
	}
	BARRIER();
	if (!part) {                                       /* :3154-3188 */
		const int r = tid;
		uint32_t hit[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		if (r >= 1 && r < H - 1)
			for (int j = 1; j < H - 1; j++) {
				const int e = r * W + j;
				if (iabs(jp[e]) >= 8 && (r >= H / 2 || j >= H / 2)) {
					if (iabs(jp[e - W - 1]) >= 8 || iabs(jp[e - W]) >= 8 || iabs(jp[e - W + 1]) >= 8 ||
					    iabs(jp[e - 1]) >= 8 || iabs(jp[e + 1]) >= 8 ||
					    iabs(jp[e + W - 1]) >= 8 || iabs(jp[e + W]) >= 8 || iabs(jp[e + W + 1]) >= 8) continue;
					hit[j >> 5] |= 1u << (j & 31);
				}
			}
		BARRIER();
		for (int j = 1; j < H - 1; j++)
			if (hit[j >> 5] & (1u << (j & 31))) { const int e = r * W + j; if (jp[e] > 0) jp[e]--; else jp[e]++; }
	}
	BARRIER();
}

/* ---------------------------------------------------------------- Y21 (R) */
/* (:970-1073) only same-row neighbours are read or written (the vertical branches are unreachable) */
DEV void tag_small_runs_par(Ctx *c, int tid)
{
	int16_t *p = c->proc;
	for (int pass = 0; pass < 2; pass++) {
		const int r = pass ? H + 1 + tid : 1 + tid;
		const int j0 = pass ? 1 : H + 1, j1 = pass ? H - 1 : W - 1;
		if (tid >= H - 2) continue;
		for (int j = j0; j < j1; j++) {
			int16_t *v = p + r * W + j;
			if (v[0] > 4 && v[0] < 8) { if (in_4_7(v[-1]) && in_4_7(v[1])) { v[0] = 12700; v[-1] = 10100; v[1] = 10100; } }
			else if (v[0] < -4 && v[0] > -8) { if (in_m7_m4(v[-1]) && in_m7_m4(v[1])) { v[0] = 12900; v[-1] = 10100; v[1] = 10100; } }
			else if (v[0] == 8) {
				if ((v[-1] & 0xFFFE) == 6 || (v[1] & 0xFFFE) == 6) v[0] = 10;
				else if (!pass && v[1] == 8) { v[0] = 9; v[1] = 9; }
			}
			else if (v[0] == -8) {
				if (((-v[-1]) & 0xFFFE) == 6 || ((-v[1]) & 0xFFFE) == 6) v[0] = -9;
				else if (!pass && v[1] == -8) { v[0] = -9; v[1] = -9; }
			}
		}
	}
}

/* ---------------------------------------------------------------- Y22 / Y23 (C) */
/* one column of Y22 (:1084-1325).  sp/so: where the reads of column j+1 (and of the recon sample (j,255) that
 * serves as lh[-1] at r = 0) are taken from: a snapshot for columns 0..254, the live planes for column 255,
 * which the reference visits last. */
DEV void classify_column(Ctx *c, int j, int res_setting, const int16_t *sp, int sp_row, const int16_t *so, int so_rows)
{
	int16_t *p = c->proc, *o = c->ll1;
	const int q = c->q;
	for (int r = 0; r < H - 1; r++) {
		const int s = r * W + j, k = r * H + j;
		int16_t *cell = o + k;
		int16_t *lh = p + j * W + H + r;
		const int lhm1 = r ? lh[-1] : sp[j * sp_row + H - 1];
		const int res = p[s] - o[k], a = p[s + W] - o[k + H];
		const int d2 = p[s + 2 * W] - o[k + 2 * H];
#define NB(dr) (sp[(r + (dr)) * sp_row + j + 1] - ((r + (dr)) < so_rows ? so[(r + (dr)) * H + j + 1] : 0))
#define MARK(code, step) do { *cell = (code); p[s + W] += (step); p[s + 2 * W] += (step); } while (0)
#define SNAP(code) do { *cell = (code); p[s + W] = o[k + H]; } while (0)
#define NUDGE_UP() do { if (lh[0] == 7) { if (lhm1 >= 0 && lhm1 < 8) lh[0] += 2; } else if (lh[0] == 8) { if (lhm1 >= -2 && lhm1 < 8) lh[0] += 2; } } while (0)
#define NUDGE_M2() do { if (lh[0] < -14) { if (mult8_or_7(-lh[0])) lh[0]++; } else if (lh[0] == 7 || (lh[0] & 0xFFFE) == 8) { if (lhm1 >= -2) lh[0] += 3; } } while (0)
#define NUDGE_M3() do { if (q >= 21) *cell = 14500; else if (lh[0] < -14) { if (mult8_or_7(-lh[0])) lh[0]++; } \
		else if (lh[0] >= 0 && ((lh[0] + 2) & 0xFFFC) == 8) { if (lhm1 >= -2) lh[0] = 10; } else if (lh[0] > 14 && (lh[0] & 7) == 7) lh[0]++; } while (0)
#define MARK_LARGE() do { *cell = 14000; if (res == -4) { if (lh[0] == -7 || lh[0] == -8) { if (lhm1 < 2 && lhm1 > -8) lh[0] = -9; } } \
		else if (res < -6) { if (res < -7 && q >= 21) *cell = 14900; else if (lh[0] < -14) { if (mult8_or_7(-lh[0])) lh[0]++; } \
		else if (lh[0] == 7 || lh[0] == 8) { if (lhm1 >= -1 && lhm1 < 8) lh[0] += 3; } } } while (0)
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
				const int x0 = NB(0), x1 = NB(1);
				if ((x0 == 2 || x0 == 3) && (x1 == 2 || x1 == 3) && NB(2) > 0) MARK(12400, -2);
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
					const int x0 = NB(0), x1 = NB(1);
					if ((x0 == -2 || x0 == -3) && (x1 == -2 || x1 == -3) && NB(2) < 0) MARK(12300, 2);
				}
				else if (res == -2) NUDGE_M2();
				else NUDGE_M3();
			}
			else if (res == -1 && a == -3 && d2 == -2) {
				if (r > 0 && (p[s - W] - o[k - H]) <= 0) MARK(12300, 2);
			}
			else if (res == -1) { if (d2 == -3) MARK(12300, 2); else NUDGE_UP(); }
			else if (res == -4) { if (d2 < -1 && d2 > -4) MARK(12300, 2); else MARK_LARGE(); }
		}
		else if (!res || res == -1) NUDGE_UP();
		else if (res == -2) NUDGE_M2();
		else if (res == -3) NUDGE_M3();
		else if (res < -res_setting) MARK_LARGE();
#undef NB
#undef MARK
#undef SNAP
#undef NUDGE_UP
#undef NUDGE_M2
#undef NUDGE_M3
#undef MARK_LARGE
	}
}

/* Y22.  The reference walks column after column; column j only reads column j+1 (not yet visited, i.e. its
 * original values) and the recon sample (j,255) of the last column.  Columns 0..254 therefore run in parallel
 * against a snapshot of the recon plane (rows 0..256) and of ll1; column 255 runs afterwards on the live
 * planes (its "column 256" is the LH1 column written by the other columns, its ll1 neighbour is column 0). */
DEV void classify_residuals_par(Ctx *c, int res_setting, int tid)
{
	int16_t *snap_p = c->hs, *snap_o = c->band;
	for (int idx = tid; idx < (H + 1) * (H / 4); idx += NT) {       /* rows 0..256 x cols 0..255 of proc */
		const int r = idx / (H / 4), k = idx % (H / 4);
		reinterpret_cast<uint2 *>(snap_p + r * H)[k] = reinterpret_cast<const uint2 *>(c->proc + r * W)[k];
	}
	for (int idx = tid; idx < Q / 4; idx += NT) reinterpret_cast<uint2 *>(snap_o)[idx] = reinterpret_cast<const uint2 *>(c->ll1)[idx];
	BARRIER();
	if (tid < H - 1) classify_column(c, tid, res_setting, snap_p, H, snap_o, H);
	BARRIER();
	if (tid == 0) classify_column(c, H - 1, res_setting, c->proc, W, c->ll1, 1 << 30);
	BARRIER();
}

/* Y23 (:1329-1420): cell (r,j) touches its own ll1 cell and the LH1 coefficient (j, 256+r), and reads
 * (j, 256+r-1), which the same column wrote one step earlier: one thread per column, serial in r. */
DEV void code_residuals_par(Ctx *c, int res_setting, int tid)
{
	int16_t *p = c->proc, *o = c->ll1;
	const int q = c->q, j = tid;
	for (int r = 0; r < H; r++) {
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

/* Y24 (:1426-1496): commutative adds; thread j owns row j of the first-order plane, the two cells that spill
 * into row j+1 (r = 254, 255) are added in a second step */
DEV void adjust_first_order_par(Ctx *c, int tid)
{
	int16_t *f = c->first_order;
	const int j = tid;
	for (int step = 0; step < 2; step++) {
		if (j < H - 2)
			for (int r = step ? H - 2 : 0; r < (step ? H : H - 2); r++) {
				const int code = c->ll1[r * H + j];
				int16_t *t = f + j * H + r;
				switch (code) {
				case 141: t[0] -= 5; break;            case 140: t[0] += 5; break;
				case 144: t[0] -= 3; break;            case 145: t[0] += 3; break;
				case 121: t[0] -= 4; t[1] -= 3; break; case 122: t[0] += 4; t[1] += 3; break;
				case 123: t[0] += 2; t[1] += 2; t[2] += 2; break;
				case 124: t[0] -= 2; t[1] -= 2; t[2] -= 2; break;
				case 126: t[0] += 9; t[1] += 3; break; case 125: t[0] -= 9; t[1] -= 3; break;
				case 148: t[0] -= 8; break;            case 149: t[0] += 8; break;
				default: break;
				}
			}
		BARRIER();
	}
}

/* ---------------------------------------------------------------- Y27 (R) */
/* (:1912-2098).  LH1 and HL1 zero everything below 6 and never move a value across 6, so "loud" (|v| >= 6) of
 * any cell is the same before, during and after these two passes: the vertical neighbour test does not care
 * which row went first -> one thread per row, serial along the row.  HH1 zeroes below 7, so a 6 above (already
 * visited in raster order) reads as quiet while a 6 below (not yet visited) reads as loud: the vertical
 * neighbours are taken from a snapshot of the band, ">= 7" for the row above, ">= 6" for the row below (a cell
 * is >= 7 after its visit exactly when it was >= 7 before).  HH1 also reads column 256, which HL1's ripple may
 * have written, hence the barrier between them. */
DEV void clean_row(int16_t *p, int r, int j0, int j1, int thresh, int lim, int lim2, int mode, int last_look, const int16_t *snap)
{
	for (int j = j0; j < j1; j++) {
		int16_t *v = p + r * W + j;
		if (iabs(v[0]) >= thresh) {
			if (iabs(v[0]) < lim2) {
				int n;
				if (mode == 2) {
					const int16_t *sv = snap + (r - H) * H + (j - H);
					n = (iabs(v[-1]) >= 6) + (iabs(v[1]) >= 6) + (r == H ? iabs(v[-W]) >= 6 : iabs(sv[-H]) >= 7) + (r == W - 1 ? 0 : (iabs(sv[H]) >= 6));
				} else n = loud_neighbours(v);
				if (mode == 0) { if (n < 3 && v[0] < lim && v[0] > -lim) { if (v[0] < -6) v[0] = -7; else if (v[0] > 6) v[0] = 7; } }
				else if (mode == 1) { if ((n < 3 && v[0] < lim && v[0] > -lim) || !n) v[0] = (int16_t)(v[0] < 0 ? -7 : 7); }
				else { if (n < 3) v[0] = (int16_t)(v[0] < 0 ? -7 : 7); }
			}
		} else v[0] = 0;
		ripple(v, j < last_look);
	}
}
DEV void clean_details_par(Ctx *c, int tid)
{
	int16_t *p = c->proc;
	const int q = c->q;
	if (tid < H - 2) clean_row(p, 1 + tid, H + 1, W - 1, DEADZONE - 2, q > 22 ? 8 : 9, q > 22 ? 4 : 9, 0, W - 2, nullptr);
	if (tid < H - 1) clean_row(p, H + tid, 1, H, DEADZONE - 2, q > 17 ? 8 : 9, q > 22 ? 4 : 9, 1, H - 2, nullptr);
	BARRIER();
	copy_block_par(p + H * W + H, W, c->hs, H, H, H, tid);           /* HH1 band (rows 256..511, cols 256..511) before its pass */
	BARRIER();
	if (tid < H - 1) { const int lim = q > 22 ? 8 : 11; clean_row(p, H + tid, H + 1, W - 1, DEADZONE - 1, lim, lim, 2, W - 2, c->hs); }
	BARRIER();
}

/* ---------------------------------------------------------------- a10 quantiser */
DEV void quant_pairs_row(int16_t *p, int r)              /* image_processing.c:195-238, one row */
{
	for (int col = (r < H ? H : 0); col < W; col++) {
		const int i = r * W + col;
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
}
DEV void quant_code_row(int16_t *p, int r, int next_first)   /* image_processing.c:314-519, one row */
{
	for (int col = 0; col < W; col++) {
		const int i = r * W + col;
		int a = p[i];
		const int nx = col < W - 1 ? p[i + 1] : next_first;
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

		if (a < -12 && ((-a) & 7) == 6) { if (col < W - 1 && nx == -7) p[i + 1] = -9; }
		if (a < 0) {
			if (a == -7 && nx == 8 && col < W - 1) { p[i] = -8; a = -8; }
			a = -a;
			if (a > 14 && (a & 7) == 7 && nx > 0 && nx < 8) a -= 2;
			if ((a & 7) < 7) a &= 504;
			a = -a;
		}
		else if (a == 8 && nx == -7 && col < W - 1) p[i + 1] = -8;
		else if (a > 12 && (a & 7) >= 6) { if (col < W - 1 && nx == 7) p[i + 1] = 9; }

		if (a < DEADZONE && a > -DEADZONE) p[i] = 128;
		else p[i] = (int16_t)((a + 128) & 248);
	}
}
/* offsetY.  Loops 1, 3, 4 reach at most two cells ahead in their own row (the one unguarded look at the first
 * cell of the next row, :389, is served from a value read before any row is rewritten); loop 2 marks cells of
 * the next row and stays serial for now. */
DEV void quantise_luma_par(Ctx *c, int tid)
{
	int16_t *p = c->proc;
	quant_pairs_row(p, tid); quant_pairs_row(p, tid + H);
	BARRIER();
	if (tid == 0) {
		for (int r = 0; r < H; r++)                    /* :241-284 (G) */
			for (int j = 1; j < H - 1; j++) {
				const int a = r * W + j;
				if (p[a] > 3 && p[a] < 8) {
					if (in_4_7(p[a - 1])) {
						if (in_4_7(p[a + 1])) { p[a] = 12700; p[a - 1] = 10100; j++; }
						else if (in_4_7(p[a + W - 1]) && in_4_7(p[a + W])) { p[a - 1] = 12100; p[a] = 10100; p[a + W - 1] = 10100; p[a + W] = 10100; j++; }
					}
				} else if (p[a] < -3 && p[a] > -8) {
					if (in_m7_m4(p[a - 1])) {
						if (in_m7_m4(p[a + 1])) { p[a] = 12900; p[a - 1] = 10100; j++; }
						else if (in_m7_m4(p[a + W - 1]) && in_m7_m4(p[a + W])) { p[a - 1] = 12200; p[a] = 10100; p[a + W - 1] = 10100; p[a + W] = 10100; j++; }
					}
				}
			}
	}
	BARRIER();
	{                                                  /* :286-311 (R) */
		const int r = tid;
		for (int j = 0; j < H - 1; j++) {
			const int a = r * W + j;
			if (is_567(p[a])) { if (is_567(p[a + 1])) { p[a] = 10300; j++; } }
			else if (is_m567(p[a])) { if (is_m567(p[a + 1])) { p[a] = 10204; j++; } }
		}
	}
	BARRIER();
	const int nf0 = p[(tid + 1) * W], nf1 = p[(tid + H + 1) * W];   /* row 511's successor is the zero guard */
	BARRIER();
	quant_code_row(p, tid, nf0); quant_code_row(p, tid + H, nf1);
	BARRIER();
}

/* offsetUV (image_processing.c:108-183): pairs never span rows; the look at the next cell is unguarded at the
 * end of a row, so the first cell of the next row is read before any row is rewritten */
DEV void quantise_chroma_par(Ctx *c, int tid)
{
	int16_t *p = c->cproc;
	const int r = tid;
	const int next_first = p[(r + 1) * H];
	BARRIER();
	for (int col = 0; col < H; col++) {
		const int i = r * H + col;
		int a = p[i];
		const int nx = col < H - 1 ? p[i + 1] : next_first;
		if (a > 10000) {
			if (a == 12400) { p[i] = 124; continue; }
			else if (a == 12600) { p[i] = 126; continue; }
			else if (a == 12900) { p[i] = 122; continue; }
			else if (a == 13000) { p[i] = 130; continue; }
		}
		if (a > 127) { p[i] = (int16_t)big_code(a, k_big_pos); continue; }
		else if (a < -127) { p[i] = (int16_t)big_code(-a, k_big_neg); continue; }
		if ((a == -7 || a == -8) && col < H - 1 && (nx == -7 || nx == -8)) { p[i] = 120; p[i + 1] = 120; col++; continue; }
		if (a < 0) {
			a = -a;
			if (nx < 0 && nx > -8) { if ((a & 7) < 6) a &= 504; }
			else { if ((a & 7) < 7) a &= 504; }
			a = -a;
		}
		else if (a > 6 && (a & 7) >= 6) { if (col < H - 1 && nx == 7) p[i + 1] = 8; }
		if (a < DEADZONE && a > -DEADZONE) p[i] = 128;
		else p[i] = (int16_t)((a + 128) & 248);
	}
	BARRIER();
}

/* ---------------------------------------------------------------- Y30 + Y31 */
DEV bool is_pm8(int v) { return v == 136 || v == 120; }
DEV bool pair_cand(const uint8_t *s, int c, int n)   /* (+-8, 0, 0, 0, +-8) starting at c */
{
	return c >= 0 && c <= n - 5 && is_pm8(s[c]) && s[c + 1] == 128 && s[c + 2] == 128 && s[c + 3] == 128 && is_pm8(s[c + 4]);
}
DEV void fix_sign_code(uint8_t *s, int at) { if (s[at] == 153) s[at] = 124; else if (s[at] == 155) s[at] = 123; }

/* (:2108-2252).  Gather is pointwise.  Rewrite 1: a match consumes its second +-8 as a possible start, so
 * along a chain of candidates spaced 4 apart every other one is taken, starting at the chain head; all tests
 * are on values the pass never changes before they are read, so the selection is computed first (bitmap) and
 * applied after a barrier.  Rewrite 2: a position is skipped exactly when its left neighbour matched the
 * pair rule (two adjacent positions cannot both match it), every other test compares against 128, which is
 * never written.  Rewrite 3 touches only sign codes next to zero runs of >= 252: each thread scans the runs
 * that start in its slice and replays the rare long ones. */
DEV void scan_and_rewrite_par(Ctx *c, int tid, int *sh_counts)
{
	const int16_t *p = c->proc;
	uint8_t *s = c->scan;
	const int n = 4 * Q;
	uint32_t *bits = reinterpret_cast<uint32_t *>(c->half);      /* n bits of selection flags */

	for (int idx = tid; idx < Q; idx += NT) {                    /* 4 columns of one row -> 4 stream bytes */
		const int r = idx >> 7, strip = idx & 127;
		const uint2 v = *reinterpret_cast<const uint2 *>(p + r * W + 4 * strip);
		const uint32_t b0 = v.x & 0xFF, b1 = (v.x >> 16) & 0xFF, b2 = v.y & 0xFF, b3 = (v.y >> 16) & 0xFF;
		const uint32_t w = (r & 1) ? (b3 | (b2 << 8) | (b1 << 16) | (b0 << 24)) : (b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
		*reinterpret_cast<uint32_t *>(s + strip * (4 * W) + 4 * r) = w;
	}
	if (tid < 4) reinterpret_cast<uint32_t *>(s + n)[tid] = 0;     /* im_nhw is calloc'ed: bytes behind the luma part read 0 here */
	if (tid == 0) { sh_counts[0] = 0; sh_counts[1] = 0; }
	BARRIER();

	const int per = n / NT, lo = tid * per, hi = lo + per;         /* 1024 stream positions per thread */
	{
		bool sel[4];
		for (int k = 0; k < 4; k++) {                              /* selected(lo-4+k) by walking its chain back */
			int cpos = lo - 4 + k, m = 0;
			while (pair_cand(s, cpos, n)) { m++; cpos -= 4; }
			sel[(lo + k) & 3] = (m & 1) != 0;
		}
		uint32_t word = 0;
		for (int cpos = lo; cpos < hi; cpos++) {
			const bool take = pair_cand(s, cpos, n) && !sel[cpos & 3];
			sel[cpos & 3] = take;
			if (take) word |= 1u << (cpos & 31);
			if ((cpos & 31) == 31) { bits[cpos >> 5] = word; word = 0; }
		}
	}
	BARRIER();
	for (int w = lo >> 5; w < (hi >> 5); w++) {
		uint32_t word = bits[w];
		while (word) {
			const int cpos = (w << 5) + __ffs((int)word) - 1;
			word &= word - 1;
			const int x = s[cpos], y = s[cpos + 4];
			s[cpos] = (uint8_t)(x == 136 ? (y == 136 ? 132 : 133) : (y == 136 ? 134 : 135));
			s[cpos + 4] = 201;
		}
	}
	BARRIER();
	if (tid < 4) { s[tid] = 128; s[n - 4 + tid] = 128; }
	BARRIER();

	{                                                              /* rewrite 2 */
		int n1 = 0, n2 = 0;
		for (int i = (lo < 4 ? 4 : lo); i < (hi > n - 4 ? n - 4 : hi); i++) {
			if (!is_pm8(s[i])) continue;
			const bool before4 = s[i - 1] == 128 && s[i - 2] == 128 && s[i - 3] == 128 && s[i - 4] == 128;
			if (i > 4 && is_pm8(s[i - 1])) {                       /* did the left neighbour take me as the second of a pair? */
				const bool b4l = s[i - 2] == 128 && s[i - 3] == 128 && s[i - 4] == 128 && s[i - 5] == 128;
				if (s[i + 1] == 128 && (b4l || (s[i - 2] == 128 && s[i + 2] == 128 && s[i + 3] == 128 && s[i + 4] == 128))) continue;
			}
			const bool pair = is_pm8(s[i + 1]);
			if ((s[i + 2] == 128 && pair && before4) ||
			    (s[i - 1] == 128 && pair && s[i + 2] == 128 && s[i + 3] == 128 && s[i + 4] == 128 && s[i + 5] == 128)) {
				s[i + 1] = (uint8_t)(s[i + 1] == 120 ? 157 : 159); n2++;
			}
			else if ((before4 && s[i + 1] == 128) || (s[i - 1] == 128 && s[i + 1] == 128 && s[i + 2] == 128 && s[i + 3] == 128 && s[i + 4] == 128)) {
				s[i] = (uint8_t)(s[i] == 136 ? 153 : 155); n1++;
			}
		}
		if (n1) atomicAdd(&sh_counts[0], n1);
		if (n2) atomicAdd(&sh_counts[1], n2);
	}
	BARRIER();
	if (tid == 0) { c->m->select1 = sh_counts[0]; c->m->select2 = sh_counts[1]; }

	for (int i = lo; i < hi; i++) {                                /* rewrite 3 */
		if (s[i] != 128 || s[i + 1] != 128) continue;
		if (i > 0 && s[i - 1] == 128) {                            /* inside a run that started earlier: its owner handles it */
			if (i == lo) { while (i < n && s[i] == 128) i++; i--; }
			continue;
		}
		int b = i;
		while (s[b + 1] == 128) b++;                               /* run [i, b]; s[n] is 0 */
		if (b - i >= 252) {                                        /* replay the reference's walk over this run */
			int k = i, run = 0;
			while (s[k] == 128 && s[k + 1] == 128) {
				run++;
				if (run > 255) { for (int t = 0; t < 4; t++) fix_sign_code(s, k + t); k--; run = 0; }
				else k++;
			}
			if (run >= 252) fix_sign_code(s, k + 1);
		}
		i = b;
	}
	BARRIER();
}


/* ---------------------------------------------------------------- chroma pieces */
DEV void dequant_row_chroma(int16_t *p, int16_t *jp, int r, int col0, int comp)
{
	for (int j = col0; j < H / 2; j++) {
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
/* offsetUV_recons256 (image_processing.c:3192-3353): p is only read, every row writes its own jp cells */
DEV void dequant_sim_chroma_par(Ctx *c, int comp, int tid)
{
	int16_t *p = c->cproc, *jp = c->cjpeg;
	for (int idx = tid; idx < (H / 4) * (H / 4); idx += NT) {
		const int r = idx >> 6, j = idx & 63, i = r * H + j;
		if (comp) {
			if (j & 1) continue;
			if (r == 0) { jp[i] = p[i]; jp[i + 1] = clear_bit0(p[i + 1]); }
			else { jp[i] = clear_bit0(p[i]); jp[i + 1] = p[i + 1]; }
		} else jp[i] = (p[i] > 0 && p[i] < 256) ? clear_bit0(p[i]) : p[i];
	}
	if (tid < H / 2) dequant_row_chroma(p, jp, tid, tid < H / 4 ? H / 4 : 0, comp);
}

/* ---------------------------------------------------------------- phases (256 threads per image) */
DEV void luma_p1_par(Ctx *c, int tid)
{
	PROF_BEGIN();
	tag_l2_details_par(c, tid);
	BARRIER();
	if (!tid) PROF(c, 0);
	dequant_sim_luma_par(c, 1, tid);
	if (!tid) PROF(c, 1);
}
DEV void luma_p2_par(Ctx *c, int tid)
{
	PROF_BEGIN();
	apply_tags_par(c, tid);
	BARRIER();
	if (!tid) PROF(c, 2);
	precompensate_ll1_par(c, tid);
	if (!tid) PROF(c, 3);
}
DEV void luma_p3_par(Ctx *c, int tid)
{
	PROF_BEGIN();
	for (int i = (Q >> 2) + tid; i < (Q >> 2) + (Q >> 3) + 64; i += NT) c->ll_bytes[i] = 0;
	BARRIER();
	if (tid == 0) {
		if (c->q > 17) tag_res4(c);
		emit_ll2(c);
		PROF(c, 4);
		ll_code_luma(c);
		PROF(c, 5);
	}
	BARRIER();
	copy_block_par(c->l2save, H, c->proc, W, H, H, tid);          /* Y17 :749-755 */
	BARRIER();
	if (!tid) PROF(c, 6);
	dequant_sim_luma_par(c, 0, tid);
	if (!tid) PROF(c, 7);
}
DEV void luma_p4_par(Ctx *c, int tid, int *sh_counts)
{
	const int q = c->q;
	PROF_BEGIN();
	if (q > 21) copy_block_par(c->jpeg, W, c->first_order, H, H, H, tid);   /* Y19 :766-777 */
	if (q < 20) {                                                           /* Y20 (:783-801) */
		int16_t *p = c->proc;
		for (int idx = tid; idx < 2 * Q; idx += NT) {
			int16_t *v = p + 2 * Q + idx;
			const int col = idx & (W - 1), m = iabs(*v);
			if (m >= DEADZONE && (col < H ? m < 9 : m <= 14)) *v = (int16_t)(*v > 0 ? 7 : -7);
		}
	}
	BARRIER();
	if (!tid) PROF(c, 8);
	tag_small_runs_par(c, tid);                                             /* Y21 */
	BARRIER();
	if (!tid) PROF(c, 9);
	const int res_setting = q >= 20 ? 3 : (q >= 18 ? 4 : 6);
	classify_residuals_par(c, res_setting, tid);                            /* Y22 */
	if (!tid) PROF(c, 10);
	code_residuals_par(c, res_setting, tid);                                /* Y23 */
	BARRIER();
	if (!tid) PROF(c, 11);
	if (q > 21) adjust_first_order_par(c, tid);                             /* Y24 */
	if (tid == 0) build_poslists(c);                                        /* Y25 */
	BARRIER();
	if (!tid) PROF(c, 12);
	for (int idx = tid; idx < Q; idx += NT) {                               /* Y26 :1893-1910 */
		const int r = idx >> 8, j = idx & 255;
		const int16_t v = c->l2save[idx];
		c->proc[r * W + j] = (r < H / 2 && j < H / 2 && v <= 8000) ? 0 : v;
	}
	BARRIER();
	if (!tid) PROF(c, 13);
	clean_details_par(c, tid);                                              /* Y27 */
	if (!tid) PROF(c, 14);
	quantise_luma_par(c, tid);                                              /* Y28 */
	if (!tid) PROF(c, 15);
	if (q > 21 && tid == 0) { band_recons(c); hq_settings(c); }             /* Y29 */
	BARRIER();
	if (!tid) PROF(c, 16);
	scan_and_rewrite_par(c, tid, sh_counts);                                /* Y30, Y31 */
	if (!tid) PROF(c, 17);
}

DEV void chroma_p0_par(Ctx *c, int comp, int tid)
{
	const uint8_t *src = comp ? c->pv : c->pu;
	for (int idx = tid; idx < Q / 4; idx += NT) {
		const uint32_t v = reinterpret_cast<const uint32_t *>(src)[idx];
		uint2 o;
		o.x = (v & 0xFF) | (((v >> 8) & 0xFF) << 16);
		o.y = ((v >> 16) & 0xFF) | ((v >> 24) << 16);
		reinterpret_cast<uint2 *>(c->cjpeg)[idx] = o;
	}
}
DEV void chroma_p3_par(Ctx *c, int comp, int tid)                     /* :2316-2336 (U), :2629-2648 (V): pointwise */
{
	int16_t *jp = c->cjpeg, *p = c->cproc, *o = c->cll1;
	for (int idx = tid; idx < Q / 4; idx += NT) {
		const int r = idx >> 7, j = idx & 127;
		const int e = r * H + j, k = idx, d = p[e] - o[k];
		const int nx = p[e + 1] - o[k + 1];
		int step = 0;
		if (d > 10) step = -6; else if (d > 7) step = -3; else if (d > 4) step = -2; else if (d > 3) step = -1;
		else if (d > 2 && (comp ? nx > 0 : nx >= 0)) step = -1;
		else if (d < -10) step = 6; else if (d < -7) step = 3; else if (d < -4) step = 2; else if (d < -3) step = 1;
		else if (d < -2 && (comp ? nx < 0 : nx <= 0)) step = 1;
		jp[e] = (int16_t)(o[k] + step);
	}
}
DEV void chroma_p5_par(Ctx *c, int comp, int tid)
{
	int16_t *p = c->cproc, *o = c->cll1;
	const int q = c->q;
	PROF_BEGIN();
	if (tid == 0 && q >= 18) {                                   /* :2372-2427 (serial: running index, skip-ahead) */
		const int res_uv = q > 17 ? 4 : 5;
		int k = 0;
		for (int r = 0; r < H / 2; r++)
			for (int j = 0; j < H / 2; j++, k++) {
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
	BARRIER();
	copy_block_par(c->cl2save, H / 2, p, H, H / 2, H / 2, tid);    /* :2431-2439 */
	BARRIER();
	if (tid == 0) {
		int e = c->m->exw_len;
		int a = comp ? (Q >> 2) + (Q >> 4) : (Q >> 2);
		c->exw[e++] = 0; c->exw[e++] = 0;                          /* :2489 (U), :2770 (V) */
		for (int r = 0; r < H / 4; r++)                            /* :2491-2525 LL2 emission */
			for (int j = 0; j < H / 4; j++) {
				int s = p[r * H + j];
				if ((s > 255 || s < 0) && (j > 0 || r > 0)) {
					int mag;
					c->exw[e++] = (uint8_t)r;
					if (s > 255) { c->exw[e++] = (uint8_t)(j + 128); mag = s - 255; }
					else { c->exw[e++] = (uint8_t)j; mag = -s; }
					c->exw[e++] = (uint8_t)(mag > 255 ? 255 : mag);
					c->ll_bytes[a] = c->ll_bytes[a - 1]; a++;
				} else {
					if (s > 255) s = 255; else if (s < 0) s = 0;
					c->ll_bytes[a++] = (uint8_t)(s & 254);
				}
				p[r * H + j] = 0;
			}
		c->m->exw_len = e;
	}
	BARRIER();
	{                                                              /* bit 1 of every LL2 sample (:2527-2548) */
		uint8_t *dst = comp ? c->res_v64 : c->res_u64;
		const uint8_t *sb = c->ll_bytes + (comp ? 20480 : 16384);
		for (int i = tid; i < 16 * H / 8; i += NT) {
			int v = 0;
			for (int b = 0; b < 8; b++) v = (v << 1) | ((sb[8 * i + b] >> 1) & 1);
			dst[i] = (uint8_t)v;
		}
	}
	if (!tid) PROF(c, 21);
	quantise_chroma_par(c, tid);
	if (!tid) PROF(c, 22);
	{                                                              /* serpentine, 32 strips of 8 columns, U even / V odd bytes (:2553-2570) */
		uint8_t *s = c->scan + 4 * Q + comp;
		for (int idx = tid; idx < Q; idx += NT) {
			const int r = idx >> 8, col = idx & 255, strip = col >> 3, k = col & 7;
			s[2 * (strip * (8 * H) + 8 * r + ((r & 1) ? 7 - k : k))] = (uint8_t)p[idx];
		}
	}
}

} // namespace nhw
#endif
