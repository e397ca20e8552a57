/*
 * nhwo_prelow.c -- oracle: the luma pre-filter of quality 1..16 and the chroma pre-filter of quality 1..14.
 * TEST INFRASTRUCTURE ONLY (see nhwo.h).
 * Reference: encoder/image_processing.c:558-2426 (pre_processing, the `quality_setting<=LOW4` branches) and
 *            :2428-2464 (pre_processing_UV).
 *
 * The reference walks the picture four times in raster order:
 *   pass A (:601-764)   contrast map ("kernel") with the 4-bit error carry of the high-quality form plus, here, a
 *                       marker rule: borderline cells (contrast just above the sharpening threshold although the plain
 *                       sum is not) get +-20000 / 7000 markers, rationed by a handful of counters that live across
 *                       the whole picture;
 *   pass B (:770-1992)  pixel pairs (1,2) (3,4) ...: optional smoothing (q<=14), then a sharpening step whose strength
 *                       is rationed by a machine of ~50 integer counters ("bursts": the first pair of a burst gets +-2,
 *                       the following ones +-1; the burst length, pauses and strength classes follow fixed schedules),
 *                       then the +-1/+-2 rules the high-quality form has too (q 15,16 and 8..10);
 *   pass C (:1994-2310) resolves the markers and sharpens low-contrast neighbours of high-contrast pixels, with
 *                       backward jumps of the pair cursor;
 *   pass D (:2312-2420) equal-sign / opposite-sign pair rules on what is left, the cursor sliding by one or two.
 * The counters have no documented meaning; they are kept as numbered cells t[1..44], w[1..8] exactly as the
 * reference numbers its variables, so that a reader can put the two side by side.
 */
#include "nhwo_internal.h"

#define S NHWO_DIM

/* what the stock binary finds behind res256 / in tree1 (GLIBC_ONESHOT mode, nhwo_luma.c); defined in nhwo_front.c */
extern int16_t nhwo_kernel_row128[4];
extern int16_t nhwo_kernel_stale[16384];
extern int16_t nhwo_kernel_row256[4];

typedef struct {
	int sharp;      /* `sharpness` (:573-587) */
	int sharp2;     /* `sharpn2`  (:589) */
	int smooth_hi;  /* `n1`       (:591-598) */
} pf_params;

static pf_params params_for(int q)
{
	static const int sharp_by_q[17] = { 0, 48, 45, 36, 24, 24, 0, 0, 0, 1, 17, 35, 41, 44, 49, 54, 59 };
	pf_params p;
	p.sharp = sharp_by_q[q];
	p.sharp2 = p.sharp < 10 ? 10 : p.sharp;
	if (q > 9) p.smooth_hi = 36;
	else if (q == 9) p.smooth_hi = 24;
	else if (q == 8) p.smooth_hi = 10;
	else if (q == 7) p.smooth_hi = 6;
	else if (q >= 3) p.smooth_hi = 36;      /* LOW14 and LOW15..LOW17 */
	else if (q == 2) p.smooth_hi = 56;
	else p.smooth_hi = 60;
	return p;
}

/* ---------------------------------------------------------------------------------------- pass A */
static void contrast_map_low(const int16_t *src, int16_t *km, const pf_params *pp)
{
	const int s2 = pp->sharp2, half = pp->sharp >> 1;
	int r, c, carry = 0;
	int neg_run = 0, neg_cycle = 0;             /* res3, t1 */
	int pos_run = 0, pos_cycle = 0;             /* a, t2 */
	int pos_alt = 0, pos_neg_alt = 0;           /* t4, t5 */
	int exact_count = 0, bump_count = 0;        /* t6, t7 */

	for (r = 1; r < S - 1; r++)
		for (c = 1; c < S - 1; c++) {
			const int16_t *p = src + r * S + c;
			int16_t *k = km + r * S + c;
			const int ctr = p[0];
			int sum = 0, mag = 0, dy, dx, acc, val;
			for (dy = -1; dy <= 1; dy++)
				for (dx = -1; dx <= 1; dx++) {
					int d;
					if (!dy && !dx) continue;
					d = ctr - p[dy * S + dx];
					sum += d; mag += iabs(d);
				}
			if (sum == 0) { k[0] = 0; carry = 0; continue; }           /* :757-762 */
			acc = 15 * iabs(sum) + mag + ((carry + 2) >> 2);
			val = acc >> 4;
			carry = acc & 15;

			if (sum < 0) {                                              /* :620-671 */
				val = -val;
				if (val == -s2 && bump_count < 3) { val = -s2 - 1; bump_count++; }
				if (-sum <= s2 && -val > s2 && -val <= s2 + 20) {
					if (c > 1 && iabs(k[-1]) <= half) neg_run = 0;
					if (!neg_run) { k[0] = -20000; neg_run = 1; }
					else {
						k[0] = (int16_t)val;
						if (!neg_cycle) { neg_run = 0; neg_cycle = 1; }
						else if (neg_run == 1) neg_run = 2;
						else { neg_run = 0; neg_cycle = (neg_cycle == 1) ? 2 : (neg_cycle == 2) ? 3 : 0; }
					}
				}
				else k[0] = (int16_t)val;
			} else {                                                    /* :672-756 */
				if (sum <= s2 && val > s2 && val <= s2 + 20) {
					if (c > 1) {
						const int left = k[-1];
						if (iabs(left) <= half) pos_run = 0;
						else if (iabs(left) > 10000 || left == s2 + 21) {
							if (!pos_alt) { pos_run = 0; if (!pos_cycle) pos_cycle = 1; pos_alt = 1; }
							else pos_alt = 0;
						}
						else if (left == -(s2 + 21)) {
							if (!pos_neg_alt) pos_neg_alt = 1;
							else {
								if (!pos_alt) { pos_run = 0; if (!pos_cycle) pos_cycle = 1; pos_alt = 1; }
								else pos_alt = 0;
								pos_neg_alt = (pos_neg_alt == 1) ? 2 : 0;
							}
						}
						else if (left == s2 + 22) k[-1] = 7000;
					}
					if (!pos_run) { k[0] = 20000; pos_run = 1; }
					else {
						k[0] = (int16_t)val;
						if (!pos_cycle) { pos_run = 0; pos_cycle = 1; }
						else if (pos_run == 1) pos_run = 2;
						else { pos_run = 0; pos_cycle = (pos_cycle == 1) ? 2 : (pos_cycle == 2) ? 3 : 0; }
					}
				}
				else if (val == s2 + 21) {
					k[0] = (int16_t)(exact_count ? val : 7000);
					exact_count++;
				}
				else k[0] = (int16_t)val;
			}
		}
}

/* ---------------------------------------------------------------------------------------- pass B */
typedef struct { int t[45], w[9]; } pf_machine;
#define T(n) (m->t[n])
#define Wv(n) (m->w[n])

static void machine_reset(pf_machine *m)          /* :770 */
{
	memset(m, 0, sizeof *m);
	T(6) = 8; T(10) = 10; T(11) = 15; T(18) = 8; T(44) = 2; Wv(3) = 20;
}

static inline void set_window(pf_machine *m, int wide) /* the (t10,t11) pair only takes the values (10,15), (8,12), (6,9) */
{
	if (wide) { T(10) = 10; T(11) = 15; } else { T(10) = 8; T(11) = 12; }
}

/* schedule taken once the burst counter t7 has reached 4 (:1203-1448): position t16, sub-position t24 */
static void long_schedule(pf_machine *m)
{
	/* t16 == 8 walks this table by t24: { next t16, value for t4 (0: keep), value for t1 (0: keep) } */
	static const int sub[14][3] = {
		{ 1, 1000000, 0 }, { 2, 0, 0 }, { 1, 1000000, 0 }, { 2, 0, 0 }, { 1, 0, 2999998 }, { 0, 0, 0 }, { 3, 0, 0 },
		{ 3, 0, 7 }, { 1, 0, 0 }, { 8, 1000000, 0 }, { 1, 8, 11 }, { 0, 0, 0 }, { 1, 0, 0 }, { 0, 0, 0 }
	};
	switch (T(16)) {
	case 0:
		set_window(m, 1); T(16) = 1;
		if ((Wv(7) == 2 || Wv(7) == 4) && T(24) == 14) { if (Wv(7) == 2) T(1) = 2000005; }
		else { T(4) = 1000000; T(1) = 9; }
		break;
	case 1:
		set_window(m, 0); T(16) = 2; Wv(5)++;
		if (Wv(5) == 3 && T(1) > 0 && T(1) < 30) T(1) = (-T(1)) >> 2;
		else { T(4) = 10; T(1) += 2; }
		break;
	case 2:
		set_window(m, 1); T(16) = 3; T(4) = 1000000; Wv(6)++;
		if (Wv(6) == 6 || Wv(6) == 10) T(1) = 10;
		break;
	case 3: set_window(m, 0); T(16) = 4; T(4) = 8; T(1) -= 4; break;
	case 4: set_window(m, 1); T(16) = 5; break;
	case 5: set_window(m, 1); T(16) = 6; T(4) = 10; T(1) = 2000000; break;
	case 6: set_window(m, 0); T(16) = 7; T(4) = 8; T(1) = 3000000; break;
	case 7: set_window(m, 0); T(16) = 8; T(4) = 1000000; break;
	case 8:
		set_window(m, 0);
		if (T(24) >= 0 && T(24) < 14) {
			const int *e = sub[T(24)];
			T(16) = e[0]; if (e[1]) T(4) = e[1]; if (e[2]) T(1) = e[2];
			T(24)++;
		}
		else if (T(24) == 14) {
			T(16) = 1; T(24) = 15; Wv(7)++;
			T(1) = Wv(2) == 0 ? 1999978 : Wv(2) == 1 ? 1999982 : 1999993;
		}
		else if (T(24) == 15) {
			T(16) = 0; T(24) = 12;
			T(1) = (Wv(2) == 1 || Wv(2) == 3) ? -5 : 2000005;
			Wv(2)++;
		}
		break;
	default: break;                                 /* t16 never leaves 0..8 */
	}
}

/* end of a burst, or the forced end above two million (:1053-1456) */
static void burst_end(pf_machine *m)
{
	if (!T(6)) {
		T(6) = 1; T(14) = 0;
		if (!T(22)) T(7)++;
		if (T(22) == 1) T(22) = 0;
	} else {
		T(6)++; T(1)++;
		if (T(4) > 900000 && T(1) == 12) T(4) = 8;
		if (T(1) > 3000000) { T(1) = 12; T(4) = 8; }
		else if (T(1) > 2000006 && T(1) < 2500000) { T(1) = 14; T(4) = 10; }
		if (!T(15)) { T(14) = 1; T(15) = 1; }
		else { T(14) = 0; T(15)++; if (T(15) > 9) T(15) = 0; }
		if (T(6) > 15 && T(7) < 4) { T(6) = 0; if (T(19) > 0) T(20)++; }
	}

	if (T(4) == 8 || (T(4) == 10 && Wv(3) > 16)) {
		if (Wv(3) < 21) { T(4) = 0; Wv(3)++; }
		else if (T(4) == 8) Wv(3) = 0;
		else if (Wv(4) < 2) { T(4) = 8; T(1) = 12; Wv(4)++; }
		else { T(4) = 0; Wv(4) = 0; }
	}
	else T(4) = 0;

	T(8) = 0; T(5) = 0; T(12) = 0;

	if (T(7) == 3) set_window(m, !T(6));
	else if (T(7) == 1) {
		set_window(m, T(9) < 2);
		T(9)++;
		if (T(9) >= 3 && T(10) == 8) T(9) = 0;         /* the wrap is in the narrow branch only (:1175-1187) */
	}
	else if (T(7) == 2) set_window(m, 0);
	else if ((T(6) == 10 || T(6) == 11) && !T(7)) { T(10) = 6; T(11) = 9; }
	else if (T(7) >= 4) long_schedule(m);
	else { T(10) = (T(10) == 8) ? 10 : 8; T(11) = (T(11) == 12) ? 15 : 12; }
}

/* a pair inside a burst that neither ends it nor sits at its cap: the slow schedules (:1504-1873) */
static void burst_idle(pf_machine *m)
{
	if (T(1) == 6 && !Wv(8)) { T(1)++; Wv(8)++; T(44) = -100000; }
	else if (T(44) < -90000) { T(1)++; Wv(8)++; T(44) = 0; }
	else if (T(44) < 3) T(44)++;
	else { T(1) += 3; T(44) = 0; }

	if (!(T(29) > 0 && (T(14) == 4 || T(14) == 5 || T(39) == 2 || T(41) > 0))) return;

	if (T(4) < 2 && T(1) == 15 && (T(14) == 4 || (T(14) == 5 && T(32) > 2))) {
		if (T(32) == 0 || T(32) == 2 || T(32) == 3 || (T(32) > 7 && T(32) < 500000)) {
			if (T(32) > 7 && T(14) == 5) { T(14) = 1; T(32) = 1000000; }
			else if (!T(34)) T(34) = 1;
			else { T(14) = 5; T(34) = 0; }
		}
		if (!T(32)) T(14) = 5;
		T(32)++;
	}
	else if (T(32) == 4 || T(32) == 5 || T(32) == 7) {
		if (T(37) == 4) T(14) = 3;
		else if (T(37) == 15) { T(14) = 3; T(32)++; }
		else if (T(32) == 7 && T(37) > -345000) {
			if (T(14) == 4) {
				if (!T(42)) T(37) -= 10000;
				if (T(38) > 0) {
					T(42)++;
					if (T(42) > 0 || (!T(42) && T(43) > 3)) {
						if (!T(42)) T(14) = (T(43) == 14) ? 3 : (T(43) == 24) ? 4 : 1;
						else T(14) = 1;
						T(39) = 0;
						if (T(42) > 5) { T(42) = -1; T(43)++; }
					}
					else if (T(42) == -1) { T(14) = 3; T(39) = 2; T(40) = -2; T(42) = 0; }
					else T(39) = 0;
				}
				else { T(14) = 5; T(39) = 1; T(42) = 0; }
			}
			else if (T(39) >= 1) {
				T(38)++;
				if (T(39) < 2) T(39) = (T(38) == 2 || T(38) == 4 || T(38) == 6 || T(38) == 9) ? 2 : 0;
				else {
					T(40)++;
					if (T(38) == 8) { T(39) = 0; T(40) = 0; }
					if (T(40) > 2) { T(40) = 0; T(39) = 0; }
				}
				if (T(38) >= 1 && T(38) <= 10) T(14) = 4;
			}
			else { T(40) = 1; if (T(38) == 1) T(39) = 2; }
		}
		if (T(37) >= 0) T(37)++;
	}
	else if (T(32) == 6 && T(36) < 118) {
		/* t36 -> { t14, what happens to t41: 0 reset, 1 increment, 4 set to 4 } */
		static const int at[13][3] = { { 1, 1, 0 }, { 2, 2, 0 }, { 3, 1, 0 }, { 4, 3, 0 }, { 5, 3, 1 }, { 6, 0, 0 }, { 7, 2, 0 },
		                               { 8, 2, 4 }, { 15, 1, 0 }, { 31, 3, 1 }, { 47, 2, 0 }, { 100, 0, 1 }, { 116, 2, 0 } };
		int k;
		if (T(14) == 4 || T(14) == 5 || T(41) == 0 || T(41) > 3) T(36)++;
		if (T(41) > 3 && T(36) < 8) T(41) = 0;
		for (k = 0; k < 13; k++)
			if (at[k][0] == T(36)) {
				T(14) = at[k][1];
				if (at[k][2] == 0) T(41) = 0; else if (at[k][2] == 1) T(41)++; else T(41) = 4;
				break;
			}
	}

	if (T(28) < 14 && T(1) > 7) {                        /* :1711-1871 */
		/* stages 6..12 fire when t30 has run far enough past t33: { distance, t14, t15, t1, t4 (0: keep) } */
		static const int late[7][5] = { { 54, 2, 3, 3, 0 }, { 57, 2, 8, 8, 0 }, { 84, 2, 7, 7, 0 }, { 111, 2, 3, 7, 0 },
		                                { 116, 1, 0, 1, 8 }, { 185, 0, 4, -17, 0 }, { 187, 3, 3, -19, 0 } };
		const int st = T(28);
		if (T(14) == 5 && !st && !T(33) && T(1) > 13 && T(31) > 0) { T(30) = 1; T(33) = 2; }
		else T(30)++;

		if (!st && T(30) > T(33) + 10 && T(33) > 0 && T(14) == 4) { T(14) = 3; T(15) += 6; T(28)++; }
		else if (st == 1 && T(30) > T(33) + 70 && T(14) == 4 && T(1) == 11) { T(15) = 1; T(1) = 13; T(28)++; }
		else if (st == 2 && T(31) > 2 && T(1) == 15 && T(15) > 1) { T(15) = 15; T(33) = T(30); T(1) = 6; T(28)++; }
		else if (st == 3 && T(30) > T(33) + 3 && T(31) > 2) { T(15) = 0; T(28)++; }
		else if (st == 5 && T(30) > T(33) + 22 && T(31) > 2 && T(1) == 12) { T(15) = 3; T(1) = 9; T(28)++; }
		else if (st == 4 && T(30) > T(33) + 6 && T(1) == 15) { T(14) = 1; T(15) += 6; T(1)++; T(28)++; }
		else if (st >= 6 && st <= 12 && T(30) > T(33) + late[st - 6][0]) {
			const int *e = late[st - 6];
			T(14) = e[1]; T(15) = e[2]; T(1) = e[3]; if (e[4]) T(4) = e[4];
			T(28)++;
		}
		else if (T(30) == T(33) + 9) { T(1) += (12 - T(4)) >> 2; T(4) = 10; }
		else if (st > 0 && T(1) == 15 && Wv(1) < 11) {
			if (T(4) != 10) { if (Wv(1) == 4 || Wv(1) == 10) T(4) = 10; Wv(1)++; }
		}
		else if (st == 13 && T(30) > T(33) + 188) { T(14) = 0; T(15) = 3; T(1) = -30; T(28)++; }
	}
}

/* one pixel pair of the sharpening machine (:838-1925).  k0/k1: the pair's contrast values (may be rewritten, in the
 * map and in the caller's copies), o: the pair in the output plane, so: the pair in the "sharpened" flag plane. */
static void machine_pair(pf_machine *m, const pf_params *pp, int row, int *pk0, int *pk1, int16_t *km, int16_t *o, uint8_t *so)
{
	const int sharp = pp->sharp, s2 = pp->sharp2;
	int k0 = *pk0, k1 = *pk1;

	if (!T(1)) {                                     /* first pair of a burst (:840-994) */
		T(2) = 0;
		if (iabs(k0) > sharp) {
			o[0] += (k0 > 0) ? 2 : -2;
			if (iabs(k1) > s2 || T(8) == 1) {
				km[0] = 0;
				if ((T(19) < 4 * Q || (T(20) >= 3 && T(20) < 4 * Q)) && iabs(k0) > sharp + 96 && T(6) > 0 && row > 2) {
					if (T(20) >= 3 && T(19) >= 8 * Q) { T(6) = 7000000; T(20) = 8 * Q; }
					if (T(19) > 0 && T(19) < 4 * Q) {
						if (T(20) > 2 || (T(20) == 2 && T(6) > 3 && !T(23)) || (T(20) == 2 && T(6) > 14 && T(23) > 0)) {
							if (T(23) == 1) T(6) = 5000000;
							T(23)++; T(21)++;
							if (T(21) >= 2) T(19) = 8 * Q;
						}
					}
					if (!T(19)) { T(6)++; T(20) = 1; }
					T(19)++;
				}
			}
			T(2) = 1;
		}
		if (iabs(k1) > sharp) {
			if ((T(2) == 1 || T(12) == 1) && (!T(14) || T(14) == 4 || T(14) == 5)) {
				if (!T(3) && T(2) == 1) {
					if (iabs(k0) > 3000) k0 = (k0 > 0) ? s2 + 5 : -s2 - 5;        /* markers count as just-above-threshold */
					if (iabs(k1) > 3000) k1 = (k1 > 0) ? s2 + 22 : -s2 - 22;
					if (iabs(k0) < (iabs(k1) >> 2)) {
						o[0] += (k0 > 0) ? -1 : 1;
						km[0] = (int16_t)k0;
						o[1] += (k1 > 0) ? 2 : -2;
						if (iabs(k0) > s2) km[1] = 0;
					}
					else o[1] += (k1 > 0) ? 1 : -1;
					T(3) = 1;
				} else {
					o[1] += (k1 > 0) ? 2 : -2;
					if (iabs(k0) > s2) km[1] = 0;
					T(3) = (T(3) == 1) ? 2 : (T(3) == 2) ? 3 : 0;
				}
			} else {
				o[1] += (k1 > 0) ? 2 : -2;
				if (iabs(k0) > s2) km[1] = 0;
			}
			if (T(14) == 2) { T(14) = 1; T(26) = 3; if (T(25) > 0) T(25)++; }
			if (T(14) == 1) { if (T(26) < 4) T(26)++; else { T(14) = 2; T(26) = 0; } }
		}
		if (iabs(k0) > sharp || iabs(k1) > sharp) T(13) = 1;
		if (T(14) == 1 || T(14) == 2) T(27)++; else T(27) = 0;
		if (T(27) > 2) T(14) = 1;
		if (T(14) == 1) {
			T(14) = 4;
			if (!T(25)) { T(15)++; T(25) = 1; }
			else { T(25)++; if (T(25) > 3) T(25) = 0; }
		}
		T(1) = 1;
	} else {                                         /* inside a burst (:995-1910) */
		if (iabs(k0) > sharp) { o[0] += (k0 > 0) ? 1 : -1; T(1)++; T(4)++; }
		if (iabs(k1) > sharp) { o[1] += (k1 > 0) ? 1 : -1; T(1)++; T(4)++; }

		if (T(4) < 10) T(17) = (T(4) == T(10) && T(1) == T(11));
		else if (T(4) > 10 || T(1) != 15) {
			if (!T(18)) { T(17) = 1; T(18) = 1; }
			else { T(17) = 0; T(18)++; if (T(18) > 15) T(18) = 0; }
		}
		else T(17) = (T(4) == T(10) && T(1) == T(11));

		if (T(6) > 6000000) { T(6) = 0; T(22) = 0; }
		else if (T(6) > 4000000) { T(6) = 0; T(22) = (T(21) == 1); }

		if (T(17) == 1 || T(1) > 2000003) burst_end(m);
		else if (T(1) >= 15) {                       /* :1457-1503 */
			if (!T(4)) T(8)++; else { T(8) = 0; T(5) = 0; T(12) = 0; }
			T(1)++;
			if (T(4) < 2 && T(29) > 0 && T(14) == 4) {
				if (T(31) == 0 || T(31) == 1) { T(14) = 3; T(31)++; }
				else if (T(31) == 2) { T(14) = 0; T(15) = 0; T(31)++; }
			}
			if (T(14) == 5 && !T(35) && T(32) > 4 && T(32) < 8) { T(14) = 1; T(32)--; T(35)++; }
		}
		else burst_idle(m);

		if (T(8) > 6 && !T(4) && T(1) > 1 && T(1) < 15) {  /* :1875-1900 */
			T(5)++;
			if (T(5) < 35) {
				T(1) = 0;
				if (!T(13)) { T(12) = 1; T(13) = 1; }
				else { T(12) = 0; T(13)++; if (T(13) > 3) T(13) = 0; }
			}
			else T(12) = 0;
		}
		if (T(1) > 15 && T(1) < 1000000) { T(1) = 0; T(4) = 0; T(29)++; }
	}

	/* opposite signs, both just above the threshold (:1912-1924) */
	if (iabs(k0) > sharp && iabs(k0) <= sharp + 20 && iabs(k1) > sharp && iabs(k1) <= sharp + 20) {
		if (k0 > 0 && k1 < 0) { o[0]++; o[1]--; so[0] = 2; so[1] = 3; }
		else if (k0 < 0 && k1 > 0) { o[0]--; o[1]++; so[0] = 3; so[1] = 2; }
	}
	*pk0 = k0; *pk1 = k1;
}

static inline int flat4(const int16_t *p)          /* the four cross neighbours within 4 of each other round the ring (:786) */
{
	return iabs(p[-S] - p[-1]) < 4 && iabs(p[-1] - p[S]) < 4 && iabs(p[S] - p[1]) < 4 && iabs(p[1] - p[-S]) < 4;
}
static inline int16_t cross_blur(const int16_t *p) /* :788-790 */
{
	return (int16_t)(((p[0] << 2) + p[-1] + p[1] + p[-S] + p[S] + 4) >> 3);
}

static void pair_pass_low(const int16_t *src, int16_t *km, int16_t *y, uint8_t *so, const pf_params *pp, int q)
{
	pf_machine mach, *m = &mach;
	const int smooth = q <= 14;
	const int tail_rules = q > 14 || (q <= 10 && q > 7);       /* :1927 */
	int r, c, prev_big = 0;

	machine_reset(m);
	for (r = 1; r < S - 1; r++)
		for (c = 1; c < S - 2; c += 2) {
			const int at = r * S + c;
			int16_t *o = y + at;
			int k0 = km[at], k1 = km[at + 1];

			if (smooth) {                                        /* :780-807 */
				if (iabs(k0) > 4 && iabs(k0) < pp->smooth_hi && flat4(src + at)) o[0] = cross_blur(src + at);
				if (iabs(k1) > 4 && iabs(k1) < pp->smooth_hi && flat4(src + at + 1)) o[1] = cross_blur(src + at + 1);
			}

			machine_pair(m, pp, r, &k0, &k1, km + at, o, so + at);

			if (!tail_rules) continue;
			/* :1927-1990, as in the high-quality form, on the (possibly rewritten) pair values */
			if (k0 < 32 && k0 > 10) {
				if (iabs(k1) >= 23) {
					if (k0 < 16) { if (k1 > 0 && k1 < 32 && k0 > 11) o[1]++; o[0]++; }
					else o[0] += prev_big ? 1 : 2;
					prev_big = 0;
					continue;
				}
			} else if (k0 > -32 && k0 < -10) {
				if (iabs(k1) >= 23) {
					if (k0 > -16) { if (k1 < 0 && k1 > -32 && k0 < -11) o[1]--; o[0]--; }
					else o[0] -= prev_big ? 1 : 2;
					prev_big = 0;
					continue;
				}
			}
			prev_big = 0;
			if (k1 < 32 && k1 > 10) {
				if (iabs(k0) >= 23) {
					if (k1 < 16) { if (k0 > 0 && k0 < 32 && k1 > 11) o[0]++; o[1]++; }
					else { o[1] += 2; prev_big = 1; }
				}
			} else if (k1 > -32 && k1 < -10) {
				if (iabs(k0) >= 23) {
					if (k1 > -16) { if (k0 < 0 && k0 > -32 && k1 < -11) o[0]--; o[1]--; }
					else { o[1] -= 2; prev_big = 1; }
				}
			}
		}
}
#undef T
#undef Wv

/* ---------------------------------------------------------------------------------------- pass C */

/* marker resolution: every third +-20000 of its kind becomes 0, the others +-5000; 7000 -> sharp2+22 (:2008-2040 etc.) */
static inline void resolve_marker(int16_t *cell, int v, int *pos_cnt, int *neg_cnt, int s2)
{
	if (v == 20000) {
		if (!*pos_cnt) { *cell = 0; *pos_cnt = 1; }
		else { *cell = 5000; *pos_cnt = (*pos_cnt == 1) ? 2 : 0; }
	}
	else if (v == -20000) {
		if (!*neg_cnt) { *cell = 0; *neg_cnt = 1; }
		else { *cell = -5000; *neg_cnt = (*neg_cnt == 1) ? 2 : 0; }
	}
	else if (v == 7000) *cell = (int16_t)(s2 + 22);
}

/* strong pixel with a weak partner: nudge the strong one, the partner if it agrees in sign, and the two pixels above
 * (:2131-2192 for the first of the pair being strong, :2205-2268 for the second).  `up` points at the map cell above
 * the right-hand pixel of the two that are looked at. */
static void sharpen_weak_partner(int strong, int weak, int16_t *ys, int16_t *yw, uint8_t *ss, uint8_t *sw,
                                 const int16_t *kup, int16_t *yup, uint8_t *sup, int have_up, int no_retry)
{
	const int sg = strong > 0 ? 1 : -1;
	*ys += sg; *ss = 1;
	if ((sg > 0 && weak > 0) || (sg < 0 && weak < 0)) { *yw += 2 * sg; *sw = 1; }
	if (have_up) {
		const int a = kup[0] * sg, b = kup[-1] * sg;      /* right / left of the two cells above, seen from the strong sign */
		if (a > 4) { yup[0] += sg; sup[0] = 1; }
		if (b > 4) { yup[-1] += sg; sup[-1] = 1; }
		if (a < -24 && no_retry) { yup[0] -= sg; sup[0] = 1; }
		if (b < -24 && no_retry) { yup[-1] -= sg; sup[-1] = 1; }
	}
}

static void marker_pass(int16_t *km, int16_t *y, uint8_t *so, const pf_params *pp)
{
	const int sharp = pp->sharp, s2 = pp->sharp2, half = sharp >> 1;
	int r, c;
	int skip_toggle = 0, second_toggle = 0;              /* t1, t2 */
	int pos0 = 0, neg0 = 0, pos1 = 0, neg1 = 0;          /* t3, t4 (first of pair), t5, t6 (second) */

	for (r = 1; r < S - 1; r++) {
		int idle = 0, retry = 0, idle_fresh = 0;         /* e, t, f */
		for (c = 1; c < S - 3; c++) {
			int at, k0, k1;
			c++;                                         /* the cursor sits on the second pixel of the pair from here on */
			at = r * S + c;
			k0 = km[at - 1]; k1 = km[at];

			if (iabs(k0) > 6000) {                       /* :2006-2089 */
				resolve_marker(km + at - 1, k0, &pos0, &neg0, s2);
				if (!second_toggle) { resolve_marker(km + at, k1, &pos1, &neg1, s2); second_toggle = 1; }
				else second_toggle = 0;
				if (!skip_toggle) { skip_toggle = 1; continue; }
				skip_toggle = 0;
			}
			else if (iabs(k1) > 6000) {                  /* :2090-2127 */
				resolve_marker(km + at, k1, &pos1, &neg1, s2);
				continue;
			}

			if (iabs(k0) > sharp + 20 && iabs(k1) > half && iabs(k1) <= s2) {          /* :2129-2202 */
				if (k0 != 0)
					sharpen_weak_partner(k0, k1, y + at - 1, y + at, so + at - 1, so + at,
					                     km + at - S, y + at - S, so + at - S, at >= 2 * S + 2, !retry);
				idle = 0; idle_fresh = 0;
				if (retry == 1) { c++; retry = 0; } else if (retry == 2) { c += 3; retry = 0; }
			}
			else if (iabs(k1) > sharp + 20 && iabs(k0) > half && iabs(k0) <= s2) {     /* :2203-2278 */
				if (k1 != 0)
					sharpen_weak_partner(k1, k0, y + at, y + at - 1, so + at, so + at - 1,
					                     km + at - S, y + at - S, so + at - S, at >= 2 * S + 2, !retry);
				idle = 0; idle_fresh = 0;
				if (retry == 1) { c++; retry = 0; } else if (retry == 2) { c += 3; retry = 0; }
			}
			else {                                       /* :2279-2308 the cursor goes back and tries the odd phase */
				idle++;
				if (!retry) idle_fresh++;
				if (idle == 2) { c -= 3; idle = 0; retry = 1; }
				else if (retry == 1) {
					c++; retry = 0; idle = 0;
					if (idle_fresh == 4) {
						const int a2 = r * S + c;
						if (iabs(km[a2 - 5]) <= s2 || iabs(km[a2 - 2]) <= s2) { c -= 5; retry = 2; }
						idle_fresh = 0;
					}
				}
				else if (retry == 2) { c += 3; retry = 0; idle = 0; idle_fresh = 0; }
			}
		}
	}
}

/* ---------------------------------------------------------------------------------------- pass D */
static void final_pair_pass(const int16_t *km, int16_t *y, const uint8_t *so, const pf_params *pp)
{
	const int sharp = pp->sharp, s2 = pp->sharp2;
	int r, c;
#define JUST_ABOVE(v, base) (iabs(v) > (base) && iabs(v) <= (base) + 20)
	for (r = 1; r < S - 1; r++)
		for (c = 1; c < S - 2; c++) {
			int at, k0, k1, slide = 0;
			int16_t *o;
			const uint8_t *f;
			c++;
			at = r * S + c;
			k0 = km[at - 1]; k1 = km[at];
			o = y + at - 1; f = so + at - 1;
			if (iabs(k0) > 4000 || iabs(k1) > 4000) continue;

			if (JUST_ABOVE(k0, sharp) && JUST_ABOVE(k1, sharp)) {                       /* :2324-2363 */
				const int next_same = c < S - 4 && JUST_ABOVE(km[at + 1], sharp) &&
				                      ((k1 > 0 && km[at + 1] > 0) || (k1 < 0 && km[at + 1] < 0));
				if (f[0] != 1 && f[1] != 1) {
					if (k0 > 0 && k1 > 0) {
						if (k0 >= k1) { if (f[0] != 2) o[0]++; else if (f[1] != 2) o[1]++; }
						else { if (f[1] != 2) o[1]++; else if (f[0] != 2) o[0]++; }
					}
					else if (k0 < 0 && k1 < 0) {
						if (k0 <= k1) { if (f[0] != 3) o[0]--; else if (f[1] != 3) o[1]--; }
						else { if (f[1] != 3) o[1]--; else if (f[0] != 3) o[0]--; }
					}
					else slide = next_same;
				}
				else slide = next_same;
			}
			else if (iabs(k0) > sharp + 56 && iabs(k1) > sharp + 56) {                  /* :2364-2382 */
				if (!f[0] && !f[1]) {
					if (k0 > 0 && k1 < 0) { o[0]++; o[1]--; }
					else if (k0 < 0 && k1 > 0) { o[0]--; o[1]++; }
					else if (iabs(k0) > sharp + 96 && iabs(k1) > sharp + 96) {
						if (k0 > 0 && k1 > 0) { if (k0 > k1) o[0]++; else o[1]++; }
						else if (k0 < 0 && k1 < 0) { if (k0 < k1) o[0]--; else o[1]--; }
					}
				}
			}
			else if (iabs(k0) > sharp + 160 && JUST_ABOVE(k1, s2)) {                    /* :2383-2398 */
				if (!f[0] && !f[1]) {
					if (k0 > 0 && k1 > 0) o[1]--;
					else if (k0 < 0 && k1 < 0) o[1]++;
					else slide = c < S - 6 && iabs(km[at + 1]) > sharp + 160 && iabs(km[at + 2]) <= s2;
				}
				else slide = c < S - 6 && iabs(km[at + 1]) > sharp + 160 && iabs(km[at + 2]) > s2 + 20;
			}
			else if (iabs(k1) > sharp + 160 && JUST_ABOVE(k0, s2)) {                    /* :2399-2414 */
				if (!f[0] && !f[1]) {
					if (k0 > 0 && k1 > 0) o[0]--;
					else if (k0 < 0 && k1 < 0) o[0]++;
					else slide = c < S - 4 && JUST_ABOVE(km[at + 1], s2);
				}
				else slide = 1;
			}
			else slide = 1;                                                             /* :2415-2418 */
			if (slide) c--;
		}
#undef JUST_ABOVE
}

/* a2 for quality 1..16: image_processing.c:558-2426 */
void nhwo_prefilter_low(int16_t *y, int quality)
{
	const pf_params pp = params_for(quality);
	int16_t *src = (int16_t *)malloc(sizeof(int16_t) * S * S);
	int16_t *km = (int16_t *)calloc(S * S, sizeof(int16_t));    /* borders are never written: read as 0 */
	uint8_t *so = (uint8_t *)calloc(S * S, 1);                   /* nhw_sharp_on */
	int c;

	memcpy(src, y, sizeof(int16_t) * S * S);                     /* :566 */
	contrast_map_low(src, km, &pp);
	pair_pass_low(src, km, y, so, &pp, quality);
	marker_pass(km, y, so, &pp);
	final_pair_pass(km, y, so, &pp);

	for (c = 0; c < 4; c++) { nhwo_kernel_row128[c] = km[128 * S + c]; nhwo_kernel_row256[c] = km[256 * S + 8 + c]; }
	memcpy(nhwo_kernel_stale, km + 262176 / 2, sizeof nhwo_kernel_stale);
	free(so); free(km); free(src);
}

/* pre_processing_UV, image_processing.c:2428-2464: 8-neighbour Laplacian on the chroma plane, one or two steps back */
void nhwo_prefilter_chroma(int16_t *plane, int quality)
{
	int16_t *src = (int16_t *)malloc(sizeof(int16_t) * Q);
	int r, c;
	memcpy(src, plane, sizeof(int16_t) * Q);
	for (r = 1; r < H - 1; r++)
		for (c = 1; c < H - 1; c++) {
			const int16_t *p = src + r * H + c;
			const int lap = (p[0] << 3) - p[-1] - p[1] - p[-H] - p[H] - p[-H - 1] - p[H - 1] - p[-H + 1] - p[H + 1];
			int16_t *o = plane + r * H + c;
			if (quality < 14) {
				if (iabs(lap) >= 14) *o += (lap > 0) ? -2 : 2;
				else if (iabs(lap) > 5) *o += (lap > 0) ? -1 : 1;
			} else {
				if (lap > 5) (*o)--; else if (lap < -5) (*o)++;
			}
		}
	free(src);
}
