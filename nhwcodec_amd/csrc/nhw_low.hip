/*
 * nhw_low.hip -- kernels that exist only for quality 1..16 of the NHW encoder (gfx950).
 *
 *   k_low_prefilter   the luma pre-filter of those settings (rcanut/nhwcodec encoder/image_processing.c:558-2426, the
 *                     `quality_setting<=LOW4` branches) and k_low_prefilter_chroma (pre_processing_UV, :2428-2464, q <= 14)
 *   k_low_ll2         Y11 + Y12 of SURVEY.md App. A (encoder/nhw_encoder.c:285-621): isolated level-2 coefficients, LL2 smoothing
 *
 * Why these are serial walks here.  Below quality 17 the reference rations its sharpening with about sixty integer counters that
 * live across the whole picture (bursts of pixel pairs whose lengths, pauses and strengths follow fixed schedules, :838-1925), resolves
 * markers with every-third-one counters (:1994-2127) and moves its pair cursor backwards (:2279-2308).  A pair's action depends on the
 * counters as the raster walk left them, and the counters on every pair before: there is no bounded-state summary of a row to compose
 * (the counters reach into the millions).  So the dependency chain is walked by ONE lane per image, and everything that is not on the
 * chain is taken off it:
 *   * one wavefront per image; the 64 lanes load the three source rows of a step with 16-byte loads into LDS, compute the 8-neighbour
 *     sums of the row in parallel, apply the q <= 14 smoothing in parallel and store finished rows;
 *   * the four raster passes of the reference (contrast map, pair machine, marker pass, final pair pass) run as ONE sweep over the rows:
 *     pass C of row r touches rows r and r-1 only and pass D of row r-1 nothing that a later row changes, so A(r) B(r) C(r) D(r-1) keep
 *     two rows of every plane in LDS and the contrast map never travels to HBM at all;
 *   * 4096 images = 4096 independent chains = 16 wavefronts per CU; the chain lane's instructions interleave with those of the other
 *     fifteen wavefronts of its CU.
 * Traffic per image: 512 KiB read, 512 KiB written (the reference makes four passes over three planes).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "nhw_ws.h"

#define DEVI __device__ static __forceinline__
#define DEVN __device__ static
#define DEVM __device__ __forceinline__
#define PF_LOOK 16         /* pixels of look-back for a lane's entry state of the carry (the 16 states have merged within 16 for everything measured; a row where they have not is replayed serially) */

namespace {

DEVI int iabs_(int v) { return v < 0 ? -v : v; }
DEVI uint64_t low_bits64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }   /* bits 0 .. n-1 */

/* The chain walks below are executed by the whole wavefront on wave-uniform values: what comes out of LDS goes through
 * v_readfirstlane, so counters and cursors live in scalar registers and every branch is a scalar branch (a lone lane inside a
 * divergent region pays an exec-mask save / restore and a vector compare per branch: measured 4x slower); stores are issued by lane 0. */
#define LDK(p) __builtin_amdgcn_readfirstlane((int)*(p))
#define STK(p, v) do { if (threadIdx.x == 0) *(p) = (v); } while (0)

/* ------------------------------------------------------------------------------------------------ parameters (:570-598) */
struct PfP { int sharp, s2, half, smooth_hi, smooth, tail_rules; };
__device__ static const uint8_t k_sharp_by_q[17] = { 0, 48, 45, 36, 24, 24, 0, 0, 0, 1, 17, 35, 41, 44, 49, 54, 59 };
DEVI PfP pf_params(int q)
{
	PfP p;
	p.sharp = k_sharp_by_q[q];
	p.s2 = p.sharp < 10 ? 10 : p.sharp;
	p.half = p.sharp >> 1;
	p.smooth_hi = q > 9 ? 36 : q == 9 ? 24 : q == 8 ? 10 : q == 7 ? 6 : q >= 3 ? 36 : q == 2 ? 56 : 60;
	p.smooth = q <= 14;
	p.tail_rules = q > 14 || (q <= 10 && q > 7);
	return p;
}

/* ------------------------------------------------------------------------------------------------ pass A: contrast map (:601-764) */
struct MapState { int carry, neg_run, neg_cycle, pos_run, pos_cycle, pos_alt, pos_neg_alt, exact_count, bump_count; };

/* The carry itself (4 bits, reset where the sum is zero) is the machine the quality 17..21 pre-filter has too: its past is forgotten within
 * a dozen pixels, so every lane finds the entry state of its 8 pixels by running the candidate states through the 16 pixels before them
 * (after one step only five neighbouring states are left: 5-bit fields of one dword) and replays its own 8 -- no walk along the row.
 * What IS order-dependent below quality 17 is the marker rule: it fires only on borderline pixels (sum not above the threshold, carried value above it), on the first three values
 * that hit the threshold from below and on the first value of threshold + 21; those few pixels are visited in raster order by the lane
 * that owns the image in the serial phases (map_cell), everything else is written by the lanes that computed it. */
/* one pixel of pass A behind the carry: sm = its 8-neighbour sum (non-zero), val = the signed carried value; k = the map row, c its column */
DEVI void map_cell(MapState &s, const PfP &pp, int c, int sm, int val, int16_t *k)
{
	const int s2 = pp.s2;
	if (sm < 0) {
		if (val == -s2 && s.bump_count < 3) { val = -s2 - 1; s.bump_count++; }
		if (-sm <= s2 && -val > s2 && -val <= s2 + 20) {          /* borderline: the plain sum is not above the threshold, the contrast is */
			if (c > 1 && iabs_(LDK(k + c - 1)) <= pp.half) s.neg_run = 0;
			if (!s.neg_run) { STK(k + c, (int16_t)(-20000)); s.neg_run = 1; }
			else {
				STK(k + c, (int16_t)((int16_t)val));
				if (!s.neg_cycle) { s.neg_run = 0; s.neg_cycle = 1; }
				else if (s.neg_run == 1) s.neg_run = 2;
				else { s.neg_run = 0; s.neg_cycle = s.neg_cycle == 1 ? 2 : s.neg_cycle == 2 ? 3 : 0; }
			}
		}
		else STK(k + c, (int16_t)((int16_t)val));
	} else {
		if (sm <= s2 && val > s2 && val <= s2 + 20) {
			if (c > 1) {
				const int left = LDK(k + c - 1);
				if (iabs_(left) <= pp.half) s.pos_run = 0;
				else if (iabs_(left) > 10000 || left == s2 + 21) {
					if (!s.pos_alt) { s.pos_run = 0; if (!s.pos_cycle) s.pos_cycle = 1; s.pos_alt = 1; }
					else s.pos_alt = 0;
				}
				else if (left == -(s2 + 21)) {
					if (!s.pos_neg_alt) s.pos_neg_alt = 1;
					else {
						if (!s.pos_alt) { s.pos_run = 0; if (!s.pos_cycle) s.pos_cycle = 1; s.pos_alt = 1; }
						else s.pos_alt = 0;
						s.pos_neg_alt = s.pos_neg_alt == 1 ? 2 : 0;
					}
				}
				else if (left == s2 + 22) STK(k + c - 1, (int16_t)7000);
			}
			if (!s.pos_run) { STK(k + c, (int16_t)(20000)); s.pos_run = 1; }
			else {
				STK(k + c, (int16_t)((int16_t)val));
				if (!s.pos_cycle) { s.pos_run = 0; s.pos_cycle = 1; }
				else if (s.pos_run == 1) s.pos_run = 2;
				else { s.pos_run = 0; s.pos_cycle = s.pos_cycle == 1 ? 2 : s.pos_cycle == 2 ? 3 : 0; }
			}
		}
		else if (val == s2 + 21) { STK(k + c, (int16_t)((int16_t)(s.exact_count ? val : 7000))); s.exact_count++; }
		else STK(k + c, (int16_t)((int16_t)val));
	}
}
/* does pixel (sm, val) need the serial visit?  Three kinds of cell do: borderline ones (the marker rule proper, a few dozen per image), the
 * first three values that hit the threshold from below (bump_count) and the first value of threshold + 21 (exact_count).  The last two
 * kinds are thousands of cells per image at the busy settings, and all but the first few leave the value as the lanes wrote it: with the
 * counts as the row finds them they are not candidates at all (a count that fills up inside the row only makes map_cell a no-op). */
DEVI bool map_candidate(const PfP &pp, int sm, int val, bool bumps_left, bool exact_left)
{
	const int s2 = pp.s2;
	if (sm < 0) return (bumps_left && val == -s2) || (-sm <= s2 && -val > s2 && -val <= s2 + 20);
	return sm > 0 && ((sm <= s2 && val > s2 && val <= s2 + 20) || (exact_left && val == s2 + 21));
}

/* a test as 0 / 1 pinned in a vector register: the lanes' flags are combined with vector instructions (as lane masks in scalar register pairs every
 * AND / OR would be an instruction of the scalar unit, the one thing a CU's sixteen chains share) */
__device__ static __forceinline__ int pf_bit_(int v) { asm volatile("" : "+v"(v)); return v; }
#define PF_BIT(x) pf_bit_((int)(x))
#include "nhw_low_machine.h"


DEVI void tail_rules(int k0, int k1, int &prev_big, int &d0, int &d1)      /* :1927-1990, on the (possibly rewritten) pair values */
{
	if (k0 < 32 && k0 > 10) {
		if (iabs_(k1) >= 23) {
			if (k0 < 16) { if (k1 > 0 && k1 < 32 && k0 > 11) d1++; d0++; }
			else d0 += prev_big ? 1 : 2;
			prev_big = 0;
			return;
		}
	} else if (k0 > -32 && k0 < -10) {
		if (iabs_(k1) >= 23) {
			if (k0 > -16) { if (k1 < 0 && k1 > -32 && k0 < -11) d1--; d0--; }
			else d0 -= prev_big ? 1 : 2;
			prev_big = 0;
			return;
		}
	}
	prev_big = 0;
	if (k1 < 32 && k1 > 10) {
		if (iabs_(k0) >= 23) {
			if (k1 < 16) { if (k0 > 0 && k0 < 32 && k1 > 11) d0++; d1++; }
			else { d1 += 2; prev_big = 1; }
		}
	} else if (k1 > -32 && k1 < -10) {
		if (iabs_(k0) >= 23) {
			if (k1 > -16) { if (k0 < 0 && k0 > -32 && k1 < -11) d0--; d1--; }
			else { d1 -= 2; prev_big = 1; }
		}
	}
}
/* what the tail rules (:1927-1990) leave in `prev_big` behind a pair: every path assigns it, so it is a function of the pair alone */
DEVI int tail_flag(int k0, int k1)
{
	if (((k0 < 32 && k0 > 10) || (k0 > -32 && k0 < -10)) && iabs_(k1) >= 23) return 0;
	if ((k1 < 32 && k1 > 10 && k1 >= 16) || (k1 > -32 && k1 < -10 && k1 <= -16)) return iabs_(k0) >= 23;
	return 0;
}
/* The picture side of one pair of pass B: what machine_step's answer `act` does to the pair's two pixels (d0, d1: additions), its two map
 * cells (k0, k1 in: values behind pass A; out: what passes C and D find) and its two flags (s0, s1).  e0 / e1: the pair's values as the
 * rules behind the machine see them (after the marker substitution).  :840-917 (first pair of a burst), :996-1001 (inside one),
 * :1912-1924 (opposite signs). */
DEVI void pair_apply(const PfP &pp, int act, int &k0, int &k1, int &e0, int &e1, int &d0, int &d1, int &s0, int &s1)
{
	const int sharp = pp.sharp, s2 = pp.s2;
	const bool f0 = iabs_(k0) > sharp, f1 = iabs_(k1) > sharp;
	const int c0 = k0, c1 = k1;
	e0 = k0; e1 = k1;
	if (act & ACT_FIRST) {
		if (f0) { d0 += c0 > 0 ? 2 : -2; if (act & ACT_ZERO0) k0 = 0; }
		if (f1) {
			if (act & ACT_SUBST) {
				if (iabs_(e0) > 3000) e0 = e0 > 0 ? s2 + 5 : -s2 - 5;          /* a marker counts as just above the threshold */
				if (iabs_(e1) > 3000) e1 = e1 > 0 ? s2 + 22 : -s2 - 22;
				if (iabs_(e0) < (iabs_(e1) >> 2)) {
					d0 += e0 > 0 ? -1 : 1;
					k0 = e0;
					d1 += e1 > 0 ? 2 : -2;
					if (iabs_(e0) > s2) k1 = 0;
				}
				else d1 += e1 > 0 ? 1 : -1;
			} else {
				d1 += c1 > 0 ? 2 : -2;
				if (iabs_(c0) > s2) k1 = 0;
			}
		}
	} else {
		if (f0) d0 += c0 > 0 ? 1 : -1;
		if (f1) d1 += c1 > 0 ? 1 : -1;
	}
	if (iabs_(e0) > sharp && iabs_(e0) <= sharp + 20 && iabs_(e1) > sharp && iabs_(e1) <= sharp + 20) {
		if (e0 > 0 && e1 < 0) { d0++; d1--; s0 = 2; s1 = 3; }
		else if (e0 < 0 && e1 > 0) { d0--; d1++; s0 = 3; s1 = 2; }
	}
}

/* ------------------------------------------------------------------------------------------------ pass C: markers, weak partners (:1994-2310) */
struct MarkState { int skip_toggle, second_toggle, pos0, neg0, pos1, neg1; };
struct CWalk { int v, idle, retry, fresh; };                        /* the cursor (second pixel of the pair) and the dance's counters (:1998-2000) */
/* magnitude classes of the 64 map cells of a window, one bit a cell (bit i = column wb + i) */
struct CMasks { uint64_t strong, weak, small, marker; };             /* |k| > sharp + 20; half < |k| <= sharp2; |k| <= sharp2; |k| > 6000 */
DEVI void c_classify(const PfP &pp, int k, bool &strong, bool &weak, bool &small, bool &marker)
{
	const int a = iabs_(k);
	strong = a > pp.sharp + 20; weak = a > pp.half && a <= pp.s2; small = a <= pp.s2; marker = a > 6000;
}
/* what a marker becomes when the walk meets it: every third +-20000 of its kind 0, the others +-5000; 7000 -> sharp2 + 22 (:2008-2040) */
DEVI int resolved_marker(int v, int &pos_cnt, int &neg_cnt, int s2)
{
	if (v == 20000) { if (!pos_cnt) { pos_cnt = 1; return 0; } pos_cnt = pos_cnt == 1 ? 2 : 0; return 5000; }
	if (v == -20000) { if (!neg_cnt) { neg_cnt = 1; return 0; } neg_cnt = neg_cnt == 1 ? 2 : 0; return -5000; }
	if (v == 7000) return s2 + 22;
	return v;
}
/* Pass C of one row over the window of columns wb .. wb + 63, from the cursor in `st` while it is <= vmax (the caller picks vmax so that
 * the cells a block looks at, up to v + 6, are inside the window; at the row's end W - 3).  The walk only asks which class a cell is in:
 * it runs on the window's bit masks.  Its dance is periodic while nothing fires: from (idle, retry, fresh) = 0 at cursor v it looks at the
 * pairs that start at v-1, v+1, v, v+3, v+5, v+4 (and at v+2 if one of the cells v+1, v+4 is small) and is back in that state at v + 8
 * (:2279-2308) -- a block without a "strong next to weak" pair (or a marker) among those is skipped by one test.
 * fx: the picture side.  fx.k(col) / fx.kup(col): map cells of the row / the row above; fx.own(col, d) / fx.up(col, d): add d to the
 * row's / the upper row's pixel and raise its flag; fx.resolve(col, value): a marker cell takes its value.  MARKERS: rows with markers
 * (their counters `ks` run across the rows of the picture: those rows go in order); without, `ks` is not touched. */
template <bool MARKERS, class FX>
DEVI void c_walk_window(CWalk &st, MarkState &ks, const PfP &pp, CMasks &mk, int wb, int vmax, bool have_up, FX &fx)
{
	int v = st.v, idle = st.idle, retry = st.retry, fresh = st.fresh;
	while (v <= vmax) {
		const int b = v - wb;
		/* bit i: the pair (wb + i, wb + i + 1) fires one way or the other (:2129, :2203), or holds a marker */
		uint64_t event = (mk.strong & (mk.weak >> 1)) | ((mk.strong >> 1) & mk.weak);
		if (MARKERS) event |= mk.marker | (mk.marker >> 1);
		if (!(idle | retry | fresh) && v + 6 <= W - 3 && !((event >> (b - 1)) & 0x7F)) { v += 8; continue; }
		const bool s0 = (mk.strong >> (b - 1)) & 1, s1 = (mk.strong >> b) & 1, w0 = (mk.weak >> (b - 1)) & 1, w1 = (mk.weak >> b) & 1;
		int k0 = 0, k1 = 0;
		bool have_k = false;
		if (MARKERS && ((mk.marker >> (b - 1)) & 3)) {                 /* :2006-2127 */
			const bool m0 = (mk.marker >> (b - 1)) & 1;
			k0 = fx.k(v - 1); k1 = fx.k(v); have_k = true;
			int n0 = k0, n1 = k1;
			bool through = false;
			if (m0) {
				n0 = resolved_marker(k0, ks.pos0, ks.neg0, pp.s2);
				if (!ks.second_toggle) { n1 = resolved_marker(k1, ks.pos1, ks.neg1, pp.s2); ks.second_toggle = 1; }
				else ks.second_toggle = 0;
				if (!ks.skip_toggle) ks.skip_toggle = 1; else { ks.skip_toggle = 0; through = true; }
			}
			else n1 = resolved_marker(k1, ks.pos1, ks.neg1, pp.s2);
			for (int h = 0; h < 2; h++) {
				const int nv = h ? n1 : n0, ov = h ? k1 : k0;
				if (nv == ov) continue;
				bool cs, cw, cl, cm;
				c_classify(pp, nv, cs, cw, cl, cm);
				const uint64_t bit = 1ull << (b - 1 + h);
				mk.strong = cs ? mk.strong | bit : mk.strong & ~bit;
				mk.weak = cw ? mk.weak | bit : mk.weak & ~bit;
				mk.small = cl ? mk.small | bit : mk.small & ~bit;
				mk.marker = cm ? mk.marker | bit : mk.marker & ~bit;
				fx.resolve(v - 1 + h, nv);
			}
			if (!through) { v += 2; continue; }                        /* the pair's rules below see the values as they were (s0 .. w1, k0, k1) */
		}
		int dv = 0;
		const bool first = s0 && w1;
		const bool second = !first && s1 && w0;
		if (first || second) {                                         /* strong pixel with a weak partner (:2129-2278) */
			if (!have_k) { k0 = fx.k(v - 1); k1 = fx.k(v); }
			const int strong = first ? k0 : k1, weak = first ? k1 : k0;
			const int cs = first ? v - 1 : v, cw = first ? v : v - 1;
			const int sg = strong > 0 ? 1 : -1;
			fx.own(cs, sg);
			if ((sg > 0 && weak > 0) || (sg < 0 && weak < 0)) fx.own(cw, 2 * sg);
			if (have_up) {
				const int a = fx.kup(v) * sg, bb = fx.kup(v - 1) * sg;
				int da = 0, db = 0;
				if (a > 4) da += sg;
				if (bb > 4) db += sg;
				if (a < -24 && !retry) da -= sg;
				if (bb < -24 && !retry) db -= sg;
				if (da) fx.up(v, da);
				if (db) fx.up(v - 1, db);
			}
			idle = 0; fresh = 0;
			if (retry == 1) dv = 1; else if (retry == 2) dv = 3;
			retry = 0;
		} else {                                                       /* the cursor goes back and tries the other pairing (:2279-2308) */
			idle++;
			if (!retry) fresh++;
			if (idle == 2) { dv = -3; idle = 0; retry = 1; }
			else if (retry == 1) {
				dv = 1; retry = 0; idle = 0;
				if (fresh == 4) {
					if (((mk.small >> (b - 4)) | (mk.small >> (b - 1))) & 1) { dv = -4; retry = 2; }
					fresh = 0;
				}
			}
			else if (retry == 2) { dv = 3; retry = 0; idle = 0; fresh = 0; }
		}
		v += 2 + dv;
	}
	st.v = v; st.idle = idle; st.retry = retry; st.fresh = fresh;
}

/* 64 cells of a 512-cell bit mask kept as bytes in cell order, from cell wb (a multiple of 32) on */
DEVI uint64_t window64(const uint8_t *mask, int wb)
{
	const uint32_t lo = reinterpret_cast<const uint32_t *>(mask)[wb >> 5];
	const uint32_t hi = wb + 32 < W ? reinterpret_cast<const uint32_t *>(mask)[(wb >> 5) + 1] : 0u;
	return (uint64_t)lo | ((uint64_t)hi << 32);
}
/* the picture side of pass C in k_low_machine (rows with markers, one lane): the rows in LDS, directly */
struct MachFx {
	int16_t *km; const int16_t *kmu; int16_t *y, *yu; uint8_t *so, *sou; uint8_t *cmask; PfP pp;
	DEVM int k(int col) const { return km[col]; }
	DEVM int kup(int col) const { return kmu[col]; }
	DEVM void own(int col, int d) { y[col] = (int16_t)(y[col] + d); so[col] = 1; }
	DEVM void up(int col, int d) { yu[col] = (int16_t)(yu[col] + d); sou[col] = 1; }
	DEVM void resolve(int col, int v)
	{
		bool c[4];
		km[col] = (int16_t)v;
		c_classify(pp, v, c[0], c[1], c[2], c[3]);
		for (int h = 0; h < 4; h++) { uint8_t &m = cmask[h * 64 + (col >> 3)]; m = c[h] ? (uint8_t)(m | (1u << (col & 7))) : (uint8_t)(m & ~(1u << (col & 7))); }
	}
};

/* ------------------------------------------------------------------------------------------------ pass D (:2312-2420) */
/* one pair of pass D: km is a row pointer indexed by column, p the pair's first column, f0 / f1 the pair's flags; d0 / d1: what the pair
 * adds to its two pixels; returns the next pair's first column (p + 2, or p + 1 where the walk slides by one) */
DEVI int final_pair(const PfP &pp, const int16_t *km, int f0, int f1, int p, int &d0, int &d1)
{
	const int sharp = pp.sharp, s2 = pp.s2;
#define JUST_ABOVE(v, base) (iabs_(v) > (base) && iabs_(v) <= (base) + 20)
	const int c = p + 1;
	const int k0 = km[c - 1], k1 = km[c];
	bool slide = false;
	if (iabs_(k0) > 4000 || iabs_(k1) > 4000) return p + 2;
	if (JUST_ABOVE(k0, sharp) && JUST_ABOVE(k1, sharp)) {
		const int k2 = km[c + 1];
		const bool next_same = c < W - 4 && JUST_ABOVE(k2, sharp) && ((k1 > 0 && k2 > 0) || (k1 < 0 && k2 < 0));
		if (f0 != 1 && f1 != 1) {
			if (k0 > 0 && k1 > 0) {
				if (k0 >= k1) { if (f0 != 2) d0++; else if (f1 != 2) d1++; }
				else { if (f1 != 2) d1++; else if (f0 != 2) d0++; }
			}
			else if (k0 < 0 && k1 < 0) {
				if (k0 <= k1) { if (f0 != 3) d0--; else if (f1 != 3) d1--; }
				else { if (f1 != 3) d1--; else if (f0 != 3) d0--; }
			}
			else slide = next_same;
		}
		else slide = next_same;
	}
	else if (iabs_(k0) > sharp + 56 && iabs_(k1) > sharp + 56) {
		if (!f0 && !f1) {
			if (k0 > 0 && k1 < 0) { d0++; d1--; }
			else if (k0 < 0 && k1 > 0) { d0--; d1++; }
			else if (iabs_(k0) > sharp + 96 && iabs_(k1) > sharp + 96) {
				if (k0 > 0 && k1 > 0) { if (k0 > k1) d0++; else d1++; }
				else if (k0 < 0 && k1 < 0) { if (k0 < k1) d0--; else d1--; }
			}
		}
	}
	else if (iabs_(k0) > sharp + 160 && JUST_ABOVE(k1, s2)) {
		if (!f0 && !f1) {
			if (k0 > 0 && k1 > 0) d1--;
			else if (k0 < 0 && k1 < 0) d1++;
			else slide = c < W - 6 && iabs_(km[c + 1]) > sharp + 160 && iabs_(km[c + 2]) <= s2;
		}
		else slide = c < W - 6 && iabs_(km[c + 1]) > sharp + 160 && iabs_(km[c + 2]) > s2 + 20;
	}
	else if (iabs_(k1) > sharp + 160 && JUST_ABOVE(k0, s2)) {
		if (!f0 && !f1) {
			if (k0 > 0 && k1 > 0) d0--;
			else if (k0 < 0 && k1 < 0) d0++;
			else slide = c < W - 4 && JUST_ABOVE(km[c + 1], s2);
		}
		else slide = true;
	}
	else slide = true;
#undef JUST_ABOVE
	return slide ? p + 1 : p + 2;
}

} // namespace

/* Passes A..C of the quality 1..16 luma pre-filter as THREE kernels (round 6; until round 5 one kernel, k_low_machine, did all of it a row at
 * a time on one wavefront: its vector stages -- pass A, codes, apply -- and its scalar chain took turns, and each waited for the other).
 *
 *   k_low_pre     pass A (contrast map, :601-764) with its few order-dependent cells, the picture copy with the q <= 14 smoothing (:780-807),
 *                 and the CODE of every pixel pair (four threshold tests of its two map cells: all the pair machine ever asks about the
 *                 picture) as ONE stream of 510 x 255 bytes in pass B's order -- the reference's pass B (:770-1992) walks the rows one after
 *                 the other with its counters running on, so a row's end means nothing to the machine;
 *   k_low_chain   the pair machine and nothing else: codes in, three answer bits a pair out.  Bursts run across row ends;
 *   k_low_apply   the answers applied to the picture and the map (pass B's picture side, :840-917 / :996-1001 / :1912-1990), the marker
 *                 classes, and pass C (:1994-2310) of the rows that hold a marker, in row order (its counters run on from marker to marker).
 *
 * What makes the split legal: a row's codes are made from the map as pass A left it (source plane + pass A's own state only) and the
 * machine's answers touch the map only behind the chain, so row r + 1's codes never depend on row r's answers.
 * src: the luma plane as the colour kernel wrote it (read only); y: the filtered plane (every row is written); km: the contrast map.
 * Codes and answers: B_KEEP (the q >= 22 plane, free below): CH_BYTES of codes, then CH_BYTES of answers. */
#define CH_N ((W - 2) * 255)                                           /* pixel pairs of a picture in pass B's order: rows 1 .. 510, pairs 0 .. 254 (cells 1 + 2 p, 2 + 2 p) */
#define CH_CHUNKS ((CH_N + 255) / 256)
#define CH_BYTES (CH_CHUNKS * 256)

#define PRE_RB 32                                                      /* rows of a band of k_low_pre / k_low_apply */
#define PRE_NB ((W - 2 + PRE_RB - 1) / PRE_RB)
#define MASK_ROW 320                                                   /* a row's five candidate masks (64 bytes each: border, bump, exact, negative sum, small sum) */
/* Pass A but for its order-dependent cells, a wavefront a band of 32 rows (until round 6: a wavefront a picture, 510 rows in turn -- 3.4 ms of
 * latency whatever the batch).  What ties the rows of pass A together is the 4-bit carry that runs on from row to row -- and forgets: the
 * band's first row takes its entry state from a look-back over the row above (that row is worked through without output; where the last
 * lane's sixteen candidate states have not merged, the band goes further back, row by row, and comes forward again with the exact state) --
 * and the marker counters (MapState), which only the few candidate cells move: those are left to k_low_mapfix, with the cells marked here
 * for every state the counters can be in (five bit masks a row). */
__global__ __launch_bounds__(64) void k_low_pre(const int16_t *__restrict__ srcb, size_t src_stride, int16_t *__restrict__ kmb, size_t km_stride,
                                                uint8_t *__restrict__ maskb, size_t mask_stride, int q, int dbg)
{
	__shared__ __attribute__((aligned(16))) int16_t s_src[3][W];
	__shared__ __attribute__((aligned(16))) int16_t s_km[W + 8];
	__shared__ __attribute__((aligned(16))) int16_t s_vb[W];              /* pass A's base values */
	__shared__ int s_misc[4];
	const int lane = threadIdx.x, band = blockIdx.x, img = blockIdx.y;
	const PfP pp = pf_params(q);
	const int c0 = lane * 8;
	const int16_t *src = srcb + (size_t)img * src_stride;
	int16_t *kmo = kmb + (size_t)img * km_stride;                      /* the contrast map as pass A leaves it (but for the cells k_low_mapfix visits) */
	uint8_t *mask = maskb + (size_t)img * mask_stride;
	const int r0 = 1 + PRE_RB * band, r1 = r0 + PRE_RB - 1 < W - 2 ? r0 + PRE_RB - 1 : W - 2;
	auto load_row = [&](int r) { *reinterpret_cast<uint4 *>(&s_src[r % 3][c0]) = *reinterpret_cast<const uint4 *>(src + (size_t)r * W + c0); };
	auto ring_init = [&](int r) { __syncthreads(); load_row(r - 1); load_row(r); };   /* rows r - 1 and r stand in the ring: row r can be worked on */
	auto load10u = [&](const int16_t *row, uint32_t *out) {           /* cells c0 - 1 .. c0 + 8 of a row in LDS as 16-bit unsigned values, sign bit flipped */
		const uint4 q4 = *reinterpret_cast<const uint4 *>(row + c0);
		const uint32_t w4[4] = { q4.x ^ 0x80008000u, q4.y ^ 0x80008000u, q4.z ^ 0x80008000u, q4.w ^ 0x80008000u };
		for (int e = 0; e < 4; e++) { out[1 + 2 * e] = w4[e] & 0xFFFFu; out[2 + 2 * e] = w4[e] >> 16; }
		out[0] = (uint32_t)(uint16_t)row[lane ? c0 - 1 : 0] ^ 0x8000u; out[9] = (uint32_t)(uint16_t)row[lane < 63 ? c0 + 8 : W - 1] ^ 0x8000u;
	};
	for (int k = lane; k < W + 8; k += 64) s_km[k] = 0;
	int16_t *km = s_km;
	/* One row of pass A.  known: the carry the row starts from is `carry_in`; otherwise nobody knows it, every lane looks back, and only
	 * the row's last lanes can be right.  Returns the carry behind the row in carry_out, and whether it is exact; emit: the row's map cells
	 * and candidate masks go out. */
	auto do_row = [&](int r, bool known, int carry_in, bool emit, int &carry_out) -> bool {
		load_row(r + 1);
		__syncthreads();
		const int16_t *up = s_src[(r - 1) % 3], *mid = s_src[r % 3], *dn = s_src[(r + 1) % 3];
		/* every lane: 8-neighbour sum and magnitude sum of its 8 pixels (:605-618), as the signed base value 15 |sum| + mag of the carry.
		 * The three rows' cells c0 - 1 .. c0 + 8 come in as one 16-byte read and two cells a row (a read a neighbour was 72 of them) */
		int smv[8], vbv[8];
		{
			/* the cells as unsigned values with the sign bit flipped (differences are what counts): the sum of the eight differences is
			 * 9 centre - the 3 x 3 block's sum, whose column sums the neighbouring pixels share; a magnitude is one v_sad_u16 */
			uint32_t u10[10], m10[10], d10[10], t10[10];
			load10u(up, u10); load10u(mid, m10); load10u(dn, d10);
			for (int i = 0; i < 10; i++) t10[i] = u10[i] + m10[i] + d10[i];
			for (int e = 0; e < 8; e++) {
				const int c = c0 + e;
				smv[e] = 0; vbv[e] = 0;
				if (c < 1 || c > W - 2) continue;
				const uint32_t ctr = m10[e + 1];
				const int sm = 9 * (int)ctr - (int)(t10[e] + t10[e + 1] + t10[e + 2]);
				uint32_t mg = __builtin_amdgcn_sad_u16(ctr, m10[e], 0u);
				mg = __builtin_amdgcn_sad_u16(ctr, m10[e + 2], mg); mg = __builtin_amdgcn_sad_u16(ctr, u10[e], mg); mg = __builtin_amdgcn_sad_u16(ctr, u10[e + 1], mg);
				mg = __builtin_amdgcn_sad_u16(ctr, u10[e + 2], mg); mg = __builtin_amdgcn_sad_u16(ctr, d10[e], mg); mg = __builtin_amdgcn_sad_u16(ctr, d10[e + 1], mg);
				mg = __builtin_amdgcn_sad_u16(ctr, d10[e + 2], mg);
				smv[e] = sm;
				vbv[e] = sm == 0 ? 0 : (sm < 0 ? -(15 * -sm + (int)mg) : 15 * sm + (int)mg);
			}
		}
		{ uint32_t w4[4];
		  for (int e = 0; e < 4; e++) w4[e] = (uint32_t)(uint16_t)vbv[2 * e] | ((uint32_t)(uint16_t)vbv[2 * e + 1] << 16);
		  *reinterpret_cast<uint4 *>(&s_vb[c0]) = make_uint4(w4[0], w4[1], w4[2], w4[3]); }
		__syncthreads();
		/* entry state of my 8 pixels: one step maps all 16 states onto the five (|v| + 0..4) & 15, so five candidates are all there is to
		 * follow: 5-bit fields of one dword (a field's c + 2 and |v| + 4 stay below 32), every step a handful of whole-dword operations.  The 16
		 * cells before mine come in as two 16-byte reads; the lanes the row's own entry state reaches (cells 1 .. c0 - 1 are fewer than 16)
		 * start from it in all five fields and skip the cells that do not exist */
		int carry;
		bool merged;
		{
			const uint32_t R = 0x108421u;                               /* 1 in each field */
			int lb[16];
			{ const uint4 a4 = *reinterpret_cast<const uint4 *>(&s_vb[c0 >= 16 ? c0 - 16 : 0]), b4 = *reinterpret_cast<const uint4 *>(&s_vb[c0 >= 8 ? c0 - 8 : 0]);
			  const uint32_t w8[8] = { a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w };
			  for (int e = 0; e < 8; e++) { lb[2 * e] = (int16_t)(w8[e] & 0xFFFF); lb[2 * e + 1] = (int16_t)(w8[e] >> 16); } }
			const bool far = c0 > PF_LOOK;
			uint32_t x = far ? (lb[0] == 0 ? 0u : ((((uint32_t)iabs_(lb[0]) & 15u) * R + 0x418820u) & (15u * R))) : (uint32_t)carry_in * R;
			for (int i = 1; i < 16; i++) {
				const int vb = lb[i];
				const uint32_t nx = vb == 0 ? 0u : ((((uint32_t)iabs_(vb) & 15u) * R + (((x + 2u * R) >> 2) & (7u * R))) & (15u * R));
				x = (far || c0 - 16 + i >= 1) ? nx : x;
			}
			merged = x == (x & 31u) * R;
			if (!known && !far) merged = false;                         /* (these lanes started from a state nobody knows) */
			carry = (int)(x & 15u);
		}
		int valv[8];
		auto run8 = [&](int cr) -> int {                                /* my eight cells from the entry state cr: their map values, the state behind them */
			for (int e = 0; e < 8; e++) {
				const int vb = vbv[e];
				valv[e] = 0;
				if (c0 + e < 1 || c0 + e > W - 2) continue;
				if (vb == 0) cr = 0;
				else {
					const int acc = iabs_(vb) + ((cr + 2) >> 2);
					valv[e] = vb < 0 ? -(acc >> 4) : (acc >> 4);
					cr = acc & 15;
				}
			}
			return cr;
		};
		int exit_c = run8(carry);
		if (!known) {                                                   /* a row worked through for its exit state only: the last lane's, if its candidates merged */
			carry_out = __builtin_amdgcn_readlane(exit_c, 63);
			return __builtin_amdgcn_readlane((int)merged, 63) != 0;
		}
#ifndef PRE_TIMING_NO_SERIAL   /* (developer timing experiment, tools/dev/pre_ab.sh: what the open lanes cost -- results are wrong without this) */
		/* Lanes whose sixteen states had not merged within the look-back (stretches of constant contrast: the carry is periodic there and never
		 * forgets) take their entry state from the lane on their left as soon as that one has its own -- lane 0 always has: it starts from the
		 * row's entry state --, a lane of every open run a round.  (Until round 6 one lane replayed the whole row, at the same cost: the runs are long.  9 % of this kernel's time) */
		while (__any(!merged)) {
			const int up_c = __shfl_up(exit_c, 1), up_ok = __shfl_up((int)merged, 1);
			if (!merged && up_ok) { merged = true; exit_c = run8(up_c); }
		}
#endif
		carry_out = __builtin_amdgcn_readlane(exit_c, 63);
		if (emit) {
			{ uint32_t w4[4]; for (int e = 0; e < 4; e++) w4[e] = (uint32_t)(uint16_t)valv[2 * e] | ((uint32_t)(uint16_t)valv[2 * e + 1] << 16);
			  *reinterpret_cast<uint4 *>(kmo + (size_t)r * W + c0) = make_uint4(w4[0], w4[1], w4[2], w4[3]); }
			/* the cells the marker rule may want to see (map_candidate, for every state of its two counters), by kind, and what map_cell asks of their sums */
			unsigned mb = 0, mu = 0, me = 0, mn = 0, msm = 0;
			const int s2 = pp.s2;
			for (int e = 0; e < 8; e++) {
				const int sm = smv[e], val = valv[e];
				if (sm == 0) continue;
				if (sm < 0) { mn |= 1u << e; if (-sm <= s2) msm |= 1u << e; if (val == -s2) mu |= 1u << e; if (-sm <= s2 && -val > s2 && -val <= s2 + 20) mb |= 1u << e; }
				else { if (sm <= s2) msm |= 1u << e; if (sm <= s2 && val > s2 && val <= s2 + 20) mb |= 1u << e; else if (val == s2 + 21) me |= 1u << e; }
			}
			uint8_t *mr = mask + (size_t)r * MASK_ROW + lane;
			mr[0] = (uint8_t)mb; mr[64] = (uint8_t)mu; mr[128] = (uint8_t)me; mr[192] = (uint8_t)mn; mr[256] = (uint8_t)msm;
		}
		return true;
	};
	/* the state the band's first row starts from */
	int carry = 0;
	if (r0 > 1) {
		int rr = r0 - 1;
		bool known = false;
		for (;;) {                                                      /* back, row by row, to a row whose exit state the look-back gives (nearly always the first) */
			if (rr == 0) { known = true; carry = 0; break; }                /* (the picture's first row starts from 0) */
			ring_init(rr);
			known = do_row(rr, false, 0, false, carry);
			if ((dbg & 32) && rr > r0 - 4 && rr > 1) known = false;        /* tests: the band goes three rows back */
			if (known) break;
			rr--;
		}
		for (int f = rr + 1; f < r0; f++) { ring_init(f); do_row(f, true, carry, false, carry); }   /* and forward again with the exact state */
	}
	ring_init(r0);
	for (int r = r0; r <= r1; r++) { do_row(r, true, carry, true, carry); __syncthreads(); }
}

/* Pass A's order-dependent cells (:620-756, map_cell): the marker rule fires only on borderline pixels, on the first three values that hit the
 * threshold from below and on the first value of threshold + 21, and its counters run on through the picture -- a few dozen cells a picture,
 * walked in raster order by one wavefront on the masks k_low_pre left (a row without a candidate costs a look at 64 bytes).  Also clears the
 * list of rows k_low_apply is about to fill (flag plane, row 0). */
__global__ __launch_bounds__(64) void k_low_mapfix(int16_t *__restrict__ kmb, size_t km_stride, const uint8_t *__restrict__ maskb, size_t mask_stride,
                                                   uint8_t *__restrict__ sob, size_t so_stride, int q, int dbg)
{
	__shared__ __attribute__((aligned(16))) int16_t s_km[W + 8];
	const int lane = threadIdx.x, img = blockIdx.x;
	const PfP pp = pf_params(q);
	int16_t *kmo = kmb + (size_t)img * km_stride;
	const uint32_t *mask = reinterpret_cast<const uint32_t *>(maskb + (size_t)img * mask_stride);
	if (lane < 16) reinterpret_cast<uint32_t *>(sob + (size_t)img * so_stride)[lane] = 0;
	if (dbg & 1) return;
	MapState ms = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	for (int k = lane; k < W + 8; k += 64) s_km[k] = 0;
	/* lane l < 16 holds dword l of each mask of the row (pixels 32 l .. 32 l + 31) */
	auto load_masks = [&](int r, uint32_t &b, uint32_t &u, uint32_t &e, uint32_t &n, uint32_t &sm) {
		const uint32_t *m = mask + (size_t)r * (MASK_ROW / 4) + (lane & 15);
		b = m[0]; u = m[16]; e = m[32]; n = m[48]; sm = m[64];
	};
	uint32_t nb, nu, ne, nn, nsm;
	load_masks(1, nb, nu, ne, nn, nsm);
	for (int r = 1; r < W - 1; r++) {
		const uint32_t b = nb, u = nu, e = ne, n = nn, sm = nsm;
		if (r + 1 < W - 1) load_masks(r + 1, nb, nu, ne, nn, nsm);
		const uint32_t cand = b | (ms.bump_count < 3 ? u : 0u) | (ms.exact_count == 0 ? e : 0u);
		const unsigned long long any = __ballot(cand != 0 && lane < 16);
		if (!any) continue;
		__syncthreads();
		*reinterpret_cast<uint4 *>(&s_km[8 * lane]) = *reinterpret_cast<const uint4 *>(kmo + (size_t)r * W + 8 * lane);
		__syncthreads();
		unsigned long long words = any;
		while (words) {
			const int w = __builtin_ctzll(words);
			words &= words - 1;
			uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)cand, w);
			const uint32_t nw = (uint32_t)__builtin_amdgcn_readlane((int)n, w), sw = (uint32_t)__builtin_amdgcn_readlane((int)sm, w);
			while (cw) {
				const int bit = __builtin_ctz(cw);
				cw &= cw - 1;
				const int c = 32 * w + bit;
				const int neg = (nw >> bit) & 1, small = (sw >> bit) & 1;
				const int smx = neg ? (small ? -1 : -(pp.s2 + 1)) : (small ? 1 : pp.s2 + 1);   /* map_cell asks for the sum's sign and whether it is above the threshold */
				map_cell(ms, pp, c, smx, LDK(&s_km[c]), s_km);
			}
		}
		__syncthreads();
		*reinterpret_cast<uint4 *>(kmo + (size_t)r * W + 8 * lane) = *reinterpret_cast<const uint4 *>(&s_km[8 * lane]);
	}
}

/* The pair machine over a picture's code stream: TWO wavefronts a picture.
 *
 * The chain (wavefront 0): everything wave-uniform on the scalar unit (nhw_low_machine.h).  The stream is taken in chunks of 256 pairs.  A
 * first pair and the burst behind it are two look-ups (the burst table's entry, the first pair's rules: table_take in nhw_low_machine.h is
 * the step, written out below on scalars of its own); a burst that starts elsewhere is decided by the lanes -- its longest clean run in one
 * step (gen_lane1 / gen_lane2 / gen_word) --, and only a pair that moves a slow schedule or ends a burst through t17 goes through
 * machine_step.  What such a step costs with sixteen chains to a CU (tools/dev/issue_probe.hip): a scalar instruction 17 cycles, a vector
 * instruction 10, a dependent LDS look-up 72 -- so the step is look-ups and vector work wherever it can be.
 *
 * The table's wavefront (wavefront 1) runs one chunk ahead on the vector unit, which the chain leaves idle: the inclusive prefix sums of
 * the pairs' hits into a ring of four chunks in LDS (16-bit: only differences are asked for; the chain's lanes read them too), their
 * inverse for the chunk's window -- the first pair at which the sum reaches a value, which answers "where does a burst's hit count reach
 * K" with one read --, and from those the four entries of each of the chunk's 256 positions (table_entries: a lane takes positions lane,
 * lane + 64, ..).  One barrier a chunk between the two.  (Until the two were one kernel the table was a kernel of its own over the whole
 * batch, 4.2 ms and 8.6 GB of traffic per 4096 pictures, in front of a chain of 3 ms at quality 1.)
 * The answers collect in LDS and leave 256 bytes a chunk. */
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_low_chain(const int16_t *__restrict__ kmb, size_t km_stride,
                                                                                             uint8_t *__restrict__ actb, size_t act_stride, int q, int dbg)
{
	__shared__ __attribute__((aligned(16))) uint16_t s_h[1024];
	__shared__ __attribute__((aligned(16))) uint16_t s_tab[2 * 1024 + 4];    /* the table's entries of two chunks (a position's four side by side); behind them four entries that say "not here" */
	__shared__ __attribute__((aligned(16))) uint8_t s_lut[512];              /* first_lut: the first pair's rules */
	__shared__ __attribute__((aligned(16))) uint8_t s_act[512];              /* the answers of this chunk's pairs (the other half: zeros for the next) */
	__shared__ __attribute__((aligned(16))) uint16_t s_inv[704];             /* the table's wavefront: s_inv[val] = the first pair of the window at which the hits' sum reaches val (999: none) */
	__shared__ __attribute__((aligned(16))) uint8_t s_code[1024];            /* the pairs' codes, a ring of four chunks: the table's wavefront makes them two chunks ahead of the chain */
	const int lane = threadIdx.x & 63, img = blockIdx.x;
	const int16_t *km = kmb + (size_t)img * km_stride;                  /* the contrast map as pass A left it */
	uint32_t *ap = reinterpret_cast<uint32_t *>(actb + (size_t)img * act_stride);
#ifdef NHW_DEV   /* developer builds: the picture's time in this kernel, in units of 64 clock ticks, in the last (unused) dword of its answers (tools/dev/gpu_chain_spread.py) */
	const long long dev_t0 = clock64();
#endif
	/* the codes of my four pairs of chunk k, made from their map cells (pair n of the stream: row 1 + n / 255, cells 1 + 2 (n % 255) and the next;
	 * 0 behind the stream's end): all the machine ever asks about the picture (nhw_low_machine.h) */
	const PfP pp = pf_params(q);
	auto load_codes = [&](int k) -> uint32_t {
		uint32_t w = 0;
		for (int j = 0; j < 4; j++) {
			const int n = 256 * k + 4 * lane + j;
			if (n >= CH_N) break;
			const int row = n / 255, pr = n - row * 255;
			const int16_t *c = km + (size_t)(row + 1) * W + 1 + 2 * pr;
			const int k0 = c[0], k1 = c[1];
			w |= ((iabs_(k0) > pp.sharp ? 1u : 0u) | (iabs_(k1) > pp.sharp ? 2u : 0u) | (iabs_(k1) > pp.s2 ? 4u : 0u) | (iabs_(k0) > pp.sharp + 96 ? 8u : 0u)) << (8 * j);
		}
		return w;
	};
	if (threadIdx.x >= 64) {
		/* ---- the table's wavefront ---- */
		auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
		int tot = 0;                                                   /* hits of all pairs before the chunk that is scanned next */
		auto scan_chunk = [&](int k, uint32_t w) {
			const uint32_t hw = (w & 0x01010101u) + ((w >> 1) & 0x01010101u);   /* hits of my four pairs, a byte each */
			const uint32_t pre = hw * 0x01010101u;                              /* their inclusive sums (at most 8) */
			const int h = (int)(pre >> 24);
			int incl = h;
			for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
			const uint32_t base = (uint32_t)(tot + incl - h);
			const uint32_t a0 = (base + (pre & 255u)) & 0xFFFFu, a1 = (base + ((pre >> 8) & 255u)) & 0xFFFFu, a2 = (base + ((pre >> 16) & 255u)) & 0xFFFFu, a3 = (base + (pre >> 24)) & 0xFFFFu;
			*reinterpret_cast<uint2 *>(&s_h[((256 * k) & 1023) + 4 * lane]) = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
			tot += __builtin_amdgcn_readlane(incl, 63);
		};
		/* the entries of chunk k (its codes: w; the sums of chunks k and k + 1 stand in the ring) */
		auto build = [&](int k) {
			const int base = 256 * k;
			const uint8_t *codes = &s_code[base & 1023];
			const uint32_t hbase = k ? (uint32_t)s_h[(base - 1) & 1023] : 0u;   /* hits of all pairs before the window */
			auto rel = [&](int wdx) { return (int)(uint16_t)((uint32_t)s_h[(base + wdx) & 1023] - hbase); };   /* hits of the window's pairs 0 .. wdx */
			if (base + 320 < CH_N && __builtin_amdgcn_readfirstlane(rel(319)) == 0) {
				/* not a hit in the window (flat stretches; most of a picture where the threshold is high): every burst that starts here runs its
				 * 20 - v pairs to the wrap -- t1 goes 1, 4, .. 16 with every fourth idle pair -- and every code is 0 */
				const uint32_t lo = 19u | (18u << 16), hi = 17u | (16u << 16);
				for (int kk = 0; kk < 4; kk++) reinterpret_cast<uint2 *>(&s_tab[1024 * (k & 1)])[lane + 64 * kk] = make_uint2(lo, hi);
				return;
			}
			for (int i = lane; i < 704 / 4; i += 64) reinterpret_cast<uint2 *>(s_inv)[i] = make_uint2(999u | (999u << 16), 999u | (999u << 16));
			wave_sync();
			auto put = [&](int w0) {
				int before = w0 ? rel(w0 - 1) : 0;
				for (int e = 0; e < 4; e++) {
					const int now = rel(w0 + e);
					if (now > before) s_inv[before + 1] = (uint16_t)(w0 + e);      /* a pair adds one or two: the values it is the first to reach */
					if (now > before + 1) s_inv[before + 2] = (uint16_t)(w0 + e);
					before = now;
				}
			};
			put(4 * lane);
			if (lane < 16) put(256 + 4 * lane);
			wave_sync();
			for (int kk = 0; kk < 4; kk++) {
				const int r = lane + 64 * kk, p = base + r;
				const int hb = rel(r);                                      /* hits of the window's pairs 0 .. r: the burst's pairs start behind pair r */
				auto g = [&](int j) { return rel(r + 1 + j) - hb; };
				auto first_ge = [&](int K) { const int wv = (int)s_inv[hb + K] - (r + 1); return wv < 32 ? wv : 32; };
				unsigned out[4];
				table_entries(g, first_ge, CH_N - (p + 1), (int)codes[r], out);
				reinterpret_cast<uint2 *>(&s_tab[1024 * (k & 1)])[r] = make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16));
			}
		};
		auto codes_chunk = [&](int k) { const uint32_t w = load_codes(k); *reinterpret_cast<uint32_t *>(&s_code[((256 * k) & 1023) + 4 * lane]) = w; scan_chunk(k, w); };
		codes_chunk(0); codes_chunk(1);
		wave_sync();
		build(0);
		__syncthreads();
		for (int k = 0; k < CH_CHUNKS; k++) {                            /* the chain is in chunk k: chunk k + 1's entries, chunk k + 2's codes and sums */
			codes_chunk(k + 2);
			wave_sync();
			build(k + 1);
			__syncthreads();
		}
		return;
	}
	/* ---- the chain ---- */
	PfM mach;
	PfC mcache;
	machine_reset(mach);
	machine_cache(mach, mcache);
	for (int i = lane; i < 512; i += 64) s_lut[i] = (uint8_t)first_lut(i);
	reinterpret_cast<uint2 *>(s_act)[lane] = make_uint2(0, 0);
	if (lane < 4) s_tab[2048 + lane] = (uint16_t)TAB_NONE;
	int pos = 0;
	__syncthreads();
	for (int k = 0; k < CH_CHUNKS; k++) {
		const int cend = 256 * (k + 1) < CH_N ? 256 * (k + 1) : CH_N;
		if (!(dbg & 2)) {
			while (pos < cend) {
				if (mach.t[1] == 0) {
					/* First pairs with their bursts through the table, as many as go: the counters such a step moves stand in scalars of
					 * their own while it lasts (t18 as (t18 - 1) & 15: the rotation passes 0 where that sum passes 15; t44 doubled: it
					 * indexes 2-byte entries; t3 and the two "is 1" bits where the first pair's table wants them), the bits that stop an
					 * entry in one word, and where no entry may be taken at all the look-up is pointed at four entries that say so. */
					const unsigned flags = (dbg & 16) ? TAB_ALL : tab_flags(mach, mcache);
					const int tsel = flags == TAB_ALL ? 4096 : 2048 * (k & 1), tmask = flags == TAB_ALL ? 0 : 0x7F8;
					int v2 = (mach.t[44] & 3) << 1, x18 = (mach.t[18] - 1) & 15, t29 = mach.t[29];
					int t8125 = mach.t[8] | (mach.t[12] << 8) | (mach.t[5] << 16);   /* the three counters a burst's cap clears (t8 <= 7, t12 <= 1, t5 < 35) */
					int fs = (mach.t[3] << 4) | ((mach.t[8] == 1) << 6) | ((mach.t[12] == 1) << 7) | (mcache.t14_045 << 8);
					unsigned seen = 0;                                   /* the codes of the first pairs taken, or-ed */
					const int pos0 = pos;
					for (;;) {
						if (pos >= cend) break;
						const unsigned entry = (unsigned)LDK(reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(s_tab) + (tsel + (((pos << 3) & tmask) | v2))));
						if (entry & flags) break;
						const int x = x18 + (int)((entry >> 7) & 7u);
						if (x > 15) break;
						x18 = x & 15;
						const unsigned lv = (unsigned)LDK(&s_lut[fs | (int)(entry >> 12)]);
						s_act[pos & 511] = (uint8_t)(lv & 7u);              /* (every lane the same byte) */
						fs = (fs & ~0x30) | (int)(lv & 0x30u);
						seen |= entry;
						t29++;
						const int e = (int)(entry & 31u);
						const int capm = ((int)(entry << 26)) >> 31;        /* all ones if the burst ends by its cap */
						v2 = (v2 + 2 * e) & 6 & capm;
						t8125 &= ~capm;
						fs &= ~(0xC0 & capm);
						pos += e + 2;
					}
					if (pos != pos0) {
						mach.t[44] = v2 >> 1; mach.t[18] = (x18 + 1) & 15; mach.t[8] = t8125 & 255; mach.t[12] = (t8125 >> 8) & 255; mach.t[5] = t8125 >> 16;
						if (seen & 0x3000u) mach.t[13] = 1;
						mach.t[29] = t29; mach.t[3] = (fs >> 4) & 3; mach.t[27] = 0;
						continue;
					}
				}
				bool step_now = false;                                  /* the pair at pos is machine_step's */
				if (mach.t[1] != 0 && burst_entry_ok(mach, mcache)) {
					/* inside a burst: its longest clean run, decided in the lanes (gen_lane1 / gen_lane2 / gen_word); the pair that stops it,
					 * if it is machine_step's, right behind */
					const int hbase = pos ? LDK(&s_h[(pos - 1) & 1023]) : 0;    /* hits of the pairs before pos */
					const int hj = (int)(uint16_t)((uint32_t)s_h[(pos + lane) & 1023] - (uint32_t)hbase);    /* hits of pairs pos .. pos + lane */
					const int hp = lane ? (int)(uint16_t)((uint32_t)s_h[(pos + lane - 1) & 1023] - (uint32_t)hbase) : 0;
					const PfGenU g = { mach.t[1], mach.t[4], mach.t[44], mach.t[10], mach.t[11], mach.t[18], mach.t[29] > 0, mach.t[30], mach.t[33], CH_N - pos };
					const PfGen1 d = gen_lane1(lane, g, mcache, hj, hj - hp);
					auto below = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
					const PfGen2 r = gen_lane2(lane, g, mcache, d, below(__ballot(d.cyc != 0)), below(__ballot(d.counting != 0)));
					const unsigned wd = gen_word(g, d, r);
					const int sl = __builtin_ctzll(__ballot(r.stop != 0));   /* lane 63 always stops */
					const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wd, sl);
					pos += gen_take(mach, sl, w);
					if (!(w & 1u)) continue;                              /* a clean run: the burst is over, or goes on behind the lanes' reach */
					step_now = true;
				}
				{                                                     /* the pair that stopped a run; a first pair the table does not take; a pair the counters do not let into a burst */
					const int code = LDK(&s_code[pos & 1023]);
					int a = step_now ? -1 : machine_step_fast(mach, mcache, code);
					if (a < 0) { a = machine_step(mach, code, 1 + pos / 255); machine_cache(mach, mcache); }
					if (a) STK(&s_act[pos & 511], (uint8_t)a);
					pos++;
				}
			}
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_wave_barrier();
		ap[64 * k + lane] = reinterpret_cast<const uint32_t *>(s_act)[64 * (k & 1) + lane];
		reinterpret_cast<uint32_t *>(s_act)[64 * (k & 1) + lane] = 0;
		__syncthreads();                                            /* the table's wavefront has chunk k + 1's entries and chunk k + 2's sums standing */
	}
#ifdef NHW_DEV
	if (threadIdx.x == 0) ap[CH_BYTES / 4 - 1] = (uint32_t)((clock64() - dev_t0) >> 6);
#endif
}

/* Pass B's picture side, a wavefront a band of 32 rows: the picture copy with the q <= 14 smoothing (:566, :780-807), the machine's answers
 * applied to the picture, the map and the flags (pair_apply: :840-917 / :996-1001 / :1912-1924), the tail rules (:1927-1990), and the list
 * of rows that hold a marker (pass C walks those in row order: k_low_markrows).  What ties the rows together here is one bit, the tail
 * rules' flag: it is a function of the pair before alone, so a band takes it from the last pair of the row above (one pair_apply).
 * act: the chain's answers (a byte a pair, stream order); km in: as passes A left it; out: y, km, flags as pass B leaves them. */
__global__ __launch_bounds__(64) void k_low_apply(const int16_t *__restrict__ srcb, size_t src_stride, int16_t *__restrict__ yb, size_t y_stride, int16_t *__restrict__ kmb, size_t km_stride,
                                                  uint8_t *__restrict__ sob, size_t so_stride, const uint8_t *__restrict__ actb, size_t act_stride, int q, int dbg)
{
	__shared__ __attribute__((aligned(16))) int16_t s_src[3][W];
	__shared__ __attribute__((aligned(16))) int16_t s_km[W + 8];
	const int lane = threadIdx.x, band = blockIdx.x, img = blockIdx.y;
	const PfP pp = pf_params(q);
	const int c0 = lane * 8;
	int16_t *yo = yb + (size_t)img * y_stride;
	int16_t *kmo = kmb + (size_t)img * km_stride;
	uint8_t *soo = sob + (size_t)img * so_stride;
	const uint8_t *act = actb + (size_t)img * act_stride;
	const int16_t *src = srcb + (size_t)img * src_stride;
	const int r0 = 1 + PRE_RB * band, r1 = r0 + PRE_RB - 1 < W - 2 ? r0 + PRE_RB - 1 : W - 2;
	for (int k = lane; k < W + 8; k += 64) s_km[k] = 0;
	/* cells c0 - 1 .. c0 + 8 of a row in LDS (0 outside the row) */
	auto load10 = [&](const int16_t *row, int *out) {
		const uint4 q4 = *reinterpret_cast<const uint4 *>(row + c0);
		const uint32_t w4[4] = { q4.x, q4.y, q4.z, q4.w };
		for (int e = 0; e < 4; e++) { out[1 + 2 * e] = (int16_t)(w4[e] & 0xFFFF); out[2 + 2 * e] = (int16_t)(w4[e] >> 16); }
		const int lo = row[lane ? c0 - 1 : 0], hi = row[lane < 63 ? c0 + 8 : W - 1];
		out[0] = lane ? lo : 0; out[9] = lane < 63 ? hi : 0;
	};
	auto load_act = [&](int r) -> uint32_t { const uint8_t *p = act + (size_t)(r - 1) * 255 + 4 * lane; return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | (lane < 63 ? (uint32_t)p[3] << 24 : 0u); };
	if (band == 0) *reinterpret_cast<uint4 *>(yo + c0) = *reinterpret_cast<const uint4 *>(src + c0);                                    /* rows 0 and 511 are not touched by any pass */
	if (band == PRE_NB - 1) *reinterpret_cast<uint4 *>(yo + (size_t)(W - 1) * W + c0) = *reinterpret_cast<const uint4 *>(src + (size_t)(W - 1) * W + c0);
	/* the map cells c0 .. c0 + 8 of a row as pass A left them, out of LDS */
	auto cells9 = [&](int *kc) {
		const uint4 kw = *reinterpret_cast<const uint4 *>(&s_km[c0]);
		const uint32_t w4[4] = { kw.x, kw.y, kw.z, kw.w };
		for (int e = 0; e < 4; e++) { kc[2 * e] = (int16_t)(w4[e] & 0xFFFF); kc[2 * e + 1] = (int16_t)(w4[e] >> 16); }
		kc[8] = s_km[c0 + 8];
	};
	int prev_big = 0;                                                  /* wave-uniform: the tail rules' flag behind the last pair of the row above */
	if (r0 > 1 && pp.tail_rules && !(dbg & 2)) {
		*reinterpret_cast<uint4 *>(&s_km[c0]) = *reinterpret_cast<const uint4 *>(kmo + (size_t)(r0 - 1) * W + c0);
		const uint32_t aw = load_act(r0 - 1);
		__syncthreads();
		int kc[9], e0 = 0, e1 = 0, d0 = 0, d1 = 0, f0 = 0, f1 = 0;
		cells9(kc);
		pair_apply(pp, (int)((aw >> 16) & 7), kc[5], kc[6], e0, e1, d0, d1, f0, f1);      /* lane 63's third pair is pair 254 */
		prev_big = __builtin_amdgcn_readlane(tail_flag(e0, e1), 63);
		__syncthreads();
	}
	uint4 nk = *reinterpret_cast<const uint4 *>(kmo + (size_t)r0 * W + c0), ns = *reinterpret_cast<const uint4 *>(src + (size_t)(r0 + 1) * W + c0);
	uint32_t na = load_act(r0);
	*reinterpret_cast<uint4 *>(&s_src[(r0 - 1) % 3][c0]) = *reinterpret_cast<const uint4 *>(src + (size_t)(r0 - 1) * W + c0);
	*reinterpret_cast<uint4 *>(&s_src[r0 % 3][c0]) = *reinterpret_cast<const uint4 *>(src + (size_t)r0 * W + c0);
	uint32_t marks = 0;                                                /* lane 0: the band's rows that hold a marker, bit r - r0 */
	for (int r = r0; r <= r1; r++) {
		*reinterpret_cast<uint4 *>(&s_km[c0]) = nk;
		*reinterpret_cast<uint4 *>(&s_src[(r + 1) % 3][c0]) = ns;
		const uint32_t aw = na;
		if (r < r1) {
			nk = *reinterpret_cast<const uint4 *>(kmo + (size_t)(r + 1) * W + c0); ns = *reinterpret_cast<const uint4 *>(src + (size_t)(r + 2) * W + c0);
			na = load_act(r + 1);
		}
		__syncthreads();
		int kc[9];
		cells9(kc);
		uint32_t yw[4];
		{
			/* the row's picture copy (:566) with the q <= 14 smoothing (:780-807: reads the source copy and the map as pass A left it) */
			const int16_t *up = s_src[(r - 1) % 3], *mid = s_src[r % 3], *dn = s_src[(r + 1) % 3];
			if (pp.smooth) {
				int u10[10], m10[10], d10[10];
				load10(up, u10); load10(mid, m10); load10(dn, d10);
				for (int e2 = 0; e2 < 4; e2++) {
					int v2[2];
					for (int h = 0; h < 2; h++) {
						const int e = 2 * e2 + h, c = c0 + e;
						const int ctr = m10[e + 1], lf = m10[e], rt = m10[e + 2], ab = u10[e + 1], bl = d10[e + 1];
						int v = ctr;
						if (c >= 1 && c <= W - 2) {
							const int kk = kc[e];
							if (iabs_(kk) > 4 && iabs_(kk) < pp.smooth_hi && iabs_(ab - lf) < 4 && iabs_(lf - bl) < 4 && iabs_(bl - rt) < 4 && iabs_(rt - ab) < 4)
								v = ((ctr << 2) + lf + rt + ab + bl + 4) >> 3;
						}
						v2[h] = v;
					}
					yw[e2] = (uint32_t)(uint16_t)v2[0] | ((uint32_t)(uint16_t)v2[1] << 16);
				}
			} else {
				const uint4 mq = *reinterpret_cast<const uint4 *>(&mid[c0]);
				yw[0] = mq.x; yw[1] = mq.y; yw[2] = mq.z; yw[3] = mq.w;
			}
		}
		/* the answers applied to the row (pass B's picture side), the tail rules, and the test for markers */
		{
			int dd[9], sv[9], e0[4], e1[4];
			for (int e = 0; e < 9; e++) { dd[e] = 0; sv[e] = 0; }
			const int npair = lane == 63 ? 3 : 4;                      /* pair 255 does not exist */
			for (int j = 0; j < 4; j++) {
				e0[j] = 0; e1[j] = 0;
				if (j < npair && !(dbg & 2)) pair_apply(pp, (int)((aw >> (8 * j)) & 7), kc[2 * j + 1], kc[2 * j + 2], e0[j], e1[j], dd[2 * j + 1], dd[2 * j + 2], sv[2 * j + 1], sv[2 * j + 2]);
			}
			if (pp.tail_rules && !(dbg & 2)) {
				int pbo[4];
				for (int j = 0; j < 4; j++) pbo[j] = tail_flag(e0[j], e1[j]);
				int pb_in = __shfl_up(pbo[3], 1);
				if (lane == 0) pb_in = prev_big;
				for (int j = 0; j < 4; j++) {
					int pb = j ? pbo[j - 1] : pb_in;
					if (j < npair) tail_rules(e0[j], e1[j], pb, dd[2 * j + 1], dd[2 * j + 2]);
				}
				prev_big = __builtin_amdgcn_readlane(pbo[2], 63);
			}
			/* cell 8 l is the second cell of the previous lane's last pair */
			{ const int pk = __shfl_up(kc[8], 1), pd = __shfl_up(dd[8], 1), ps = __shfl_up(sv[8], 1);
			  if (lane) { kc[0] = pk; dd[0] = pd; sv[0] = ps; } }
			bool mark = false;
			for (int e = 0; e < 8; e++) mark |= iabs_(kc[e]) > 6000;
			{ uint32_t w4[4]; for (int e = 0; e < 4; e++) w4[e] = (uint32_t)(uint16_t)kc[2 * e] | ((uint32_t)(uint16_t)kc[2 * e + 1] << 16);
			  *reinterpret_cast<uint4 *>(kmo + (size_t)r * W + c0) = make_uint4(w4[0], w4[1], w4[2], w4[3]); }
			{ for (int e = 0; e < 4; e++) {
				const int lo = (int16_t)(yw[e] & 0xFFFF) + dd[2 * e], hi = (int16_t)(yw[e] >> 16) + dd[2 * e + 1];
				yw[e] = (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16);
			  }
			  *reinterpret_cast<uint4 *>(yo + (size_t)r * W + c0) = make_uint4(yw[0], yw[1], yw[2], yw[3]); }
			{ uint32_t lo = 0, hi = 0;
			  for (int e = 0; e < 4; e++) { lo |= (uint32_t)sv[e] << (8 * e); hi |= (uint32_t)sv[4 + e] << (8 * e); }
			  *reinterpret_cast<uint2 *>(soo + (size_t)r * W + c0) = make_uint2(lo, hi); }
			if (__any(mark) && !(dbg & 4)) marks |= 1u << (r - r0);
		}
		__syncthreads();
	}
	if (lane == 0 && marks) {                                         /* flag plane, row 0 (k_low_mapfix cleared it): bit r of its first 512 = row r holds a marker */
		uint32_t *rm = reinterpret_cast<uint32_t *>(soo);
		atomicOr(&rm[r0 >> 5], marks << (r0 & 31));
		if ((r0 & 31) && (marks >> (32 - (r0 & 31)))) atomicOr(&rm[(r0 >> 5) + 1], marks >> (32 - (r0 & 31)));
	}
}

/* Pass C (:1994-2310) of the rows that hold a marker, in row order (its counters only move at markers and run on from marker to marker;
 * every other row is k_low_marks', a lane a row): one wavefront a picture takes the listed rows one after the other -- the row and the row
 * above in LDS, the row's cells by class as bit masks, lane 0 walks (c_walk_window), both rows go back. */
__global__ __launch_bounds__(64) void k_low_markrows(int16_t *__restrict__ yb, size_t y_stride, int16_t *__restrict__ kmb, size_t km_stride, uint8_t *__restrict__ sob, size_t so_stride, int q)
{
	__shared__ __attribute__((aligned(16))) int16_t s_km[2][W + 8];
	__shared__ __attribute__((aligned(16))) int16_t s_y[2][W];
	__shared__ __attribute__((aligned(16))) uint8_t s_so[2][W];
	__shared__ __attribute__((aligned(8))) uint8_t s_cmask[4][64];          /* the row's cells by class (c_classify), a bit a cell */
	const int lane = threadIdx.x, img = blockIdx.x;
	const PfP pp = pf_params(q);
	const int c0 = lane * 8;
	int16_t *yo = yb + (size_t)img * y_stride;
	int16_t *kmo = kmb + (size_t)img * km_stride;
	uint8_t *soo = sob + (size_t)img * so_stride;
	uint32_t mw = lane < 16 ? reinterpret_cast<const uint32_t *>(soo)[lane] : 0u;
	unsigned long long words = __ballot(mw != 0);
	if (!words) return;
	for (int k = lane; k < 2 * (W + 8); k += 64) (&s_km[0][0])[k] = 0;
	MarkState ks = { 0, 0, 0, 0, 0, 0 };
	auto load = [&](int r) {
		*reinterpret_cast<uint4 *>(&s_km[r & 1][c0]) = *reinterpret_cast<const uint4 *>(kmo + (size_t)r * W + c0);
		*reinterpret_cast<uint4 *>(&s_y[r & 1][c0]) = *reinterpret_cast<const uint4 *>(yo + (size_t)r * W + c0);
		*reinterpret_cast<uint2 *>(&s_so[r & 1][c0]) = *reinterpret_cast<const uint2 *>(soo + (size_t)r * W + c0);
	};
	auto store = [&](int r) {
		*reinterpret_cast<uint4 *>(yo + (size_t)r * W + c0) = *reinterpret_cast<const uint4 *>(&s_y[r & 1][c0]);
		*reinterpret_cast<uint4 *>(kmo + (size_t)r * W + c0) = *reinterpret_cast<const uint4 *>(&s_km[r & 1][c0]);
		*reinterpret_cast<uint2 *>(soo + (size_t)r * W + c0) = *reinterpret_cast<const uint2 *>(&s_so[r & 1][c0]);
	};
	int prev = -1;                                                     /* the row in the other slot */
	while (words) {
		const int w = __builtin_ctzll(words);
		words &= words - 1;
		uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)mw, w);
		while (bits) {
			const int r = 32 * w + __builtin_ctz(bits);
			bits &= bits - 1;
			__syncthreads();
			if (r >= 2 && prev != r - 1) load(r - 1);
			load(r);
			__syncthreads();
			{
				uint32_t cs = 0, cwk = 0, cl = 0, cm = 0;
				const uint4 kw = *reinterpret_cast<const uint4 *>(&s_km[r & 1][c0]);
				const uint32_t w4[4] = { kw.x, kw.y, kw.z, kw.w };
				for (int e = 0; e < 8; e++) {
					bool a, b2, c2, d2;
					c_classify(pp, (int16_t)((e & 1) ? w4[e >> 1] >> 16 : w4[e >> 1] & 0xFFFF), a, b2, c2, d2);
					cs |= (uint32_t)a << e; cwk |= (uint32_t)b2 << e; cl |= (uint32_t)c2 << e; cm |= (uint32_t)d2 << e;
				}
				s_cmask[0][lane] = (uint8_t)cs; s_cmask[1][lane] = (uint8_t)cwk; s_cmask[2][lane] = (uint8_t)cl; s_cmask[3][lane] = (uint8_t)cm;
			}
			__syncthreads();
			if (lane == 0) {
				MachFx fx = { s_km[r & 1], s_km[(r - 1) & 1], s_y[r & 1], s_y[(r - 1) & 1], s_so[r & 1], s_so[(r - 1) & 1], &s_cmask[0][0], pp };
				CWalk cw = { 2, 0, 0, 0 };
				while (cw.v <= W - 3) {
					const int wb = cw.v < 40 ? 0 : ((cw.v - 8) & ~31);
					CMasks cm;
					cm.strong = window64(s_cmask[0], wb); cm.weak = window64(s_cmask[1], wb); cm.small = window64(s_cmask[2], wb); cm.marker = window64(s_cmask[3], wb);
					c_walk_window<true>(cw, ks, pp, cm, wb, wb + 57 < W - 3 ? wb + 57 : W - 3, r >= 2, fx);
				}
			}
			__syncthreads();
			if (r >= 2) store(r - 1);
			store(r);
			prev = r;
		}
	}
}

/* Passes C and D for the rows pass C can take independently of each other -- every row without a marker (k_low_machine lists the others
 * and has walked their pass C itself) -- a lane per row, 191 rows (+ the one below them) to a workgroup, in windows of 64 columns.
 *
 * Pass C (:1994-2310) without markers keeps nothing from row to row: its cursor dance starts afresh in every row; it reads the contrast
 * map of its row and of the row above and ADDS to the picture in both (and sets their flags to 1): additions commute, so rows need no
 * order.  Both walks only ask which of a few magnitude classes a map cell is in, so a lane turns the 64 cells of its row's window into bit
 * masks once and then walks on registers:
 *   * pass C's dance is periodic while nothing fires: from (idle, retry, fresh) = 0 at cursor v it looks at the pairs starting at
 *     v-1, v+1, v, v+3, v+5, v+4 (and v+2 if one of the cells v+1, v+4 is small) and is back in the same state at v+8 (:2279-2308).  A block
 *     without a "strong next to weak" pair among those is skipped with one mask test; only blocks with one are walked visit by visit;
 *   * pass D (:2312-2420) slides by one wherever none of its five pair rules applies: the cursor jumps to the next pair where one does.
 * What the walks add goes to byte planes of the lane's own (own row / the row above), the flags they raise to bit masks; nothing else is
 * written during the walks, so lanes never meet.  Pass D of a row needs the row's flags as pass C of this row and of the row below leave
 * them -- final a few columns behind both cursors -- so it follows inside the same window loop.  Then the window's sums go to the picture
 * (and its flags to the flag plane) where there are any: the picture is not read at all elsewhere.
 * (Round 2 ran pass C on the chain lane of the pre-filter kernel, 30 % of its 271 ms, and pass D as a kernel over the three planes: 21 GB;
 * the first row-parallel form -- LDS loads and compares at every visit -- took 15.5 ms per batch at quality 10.) */
#define MK_R 192                                                       /* lanes: 191 rows + the row below them */
#define MK_A 48                                                        /* columns a window advances */
#define MK_BP 68                                                       /* pitch of the byte planes */
namespace {
struct MkMasks { uint64_t strong, weak, small, ja, g56, g160, jas, big; };
DEVI uint64_t bits_from(uint64_t m, int b) { return b >= 64 ? 0 : b <= 0 ? m : (m >> b); }   /* m >> b with the window's edges */
/* the picture side of pass C in k_low_marks: the lane's byte planes and bit masks */
struct MkFx {
	const int16_t *kr, *ku;
	int8_t *ow, *uw;
	int wb;
	uint64_t own_flag, own_dirty, up_flag, up_dirty;
	DEVM int k(int col) const { return kr[col]; }
	DEVM int kup(int col) const { return ku[col]; }
	DEVM void own(int col, int d) { ow[col] = (int8_t)(ow[col] + d); own_flag |= 1ull << (col - wb); own_dirty |= 1ull << (col - wb); }
	DEVM void up(int col, int d) { uw[col] = (int8_t)(uw[col] + d); up_flag |= 1ull << (col - wb); up_dirty |= 1ull << (col - wb); }
	DEVM void resolve(int, int) {}
};
}
__global__ __launch_bounds__(MK_R) void k_low_marks(int16_t *__restrict__ yb, size_t y_stride, const int16_t *__restrict__ kmb, size_t km_stride,
                                                    uint8_t *__restrict__ sob, size_t so_stride, int q, int dbg)
{
	__shared__ __attribute__((aligned(16))) int8_t own[MK_R * MK_BP], upd[MK_R * MK_BP];   /* what a lane adds to its row / to the row above */
	__shared__ uint64_t s_upflag[MK_R + 1], s_updirty[MK_R + 1];          /* of the lane's additions to the row above: flags raised, cells touched */
	__shared__ uint32_t s_mask[16];
	const int tid = threadIdx.x, img = blockIdx.y;
	const int R0 = 1 + (MK_R - 1) * blockIdx.x;                        /* rows R0 .. R0 + 190 are this workgroup's; the last lane walks pass C of the row below them for what it adds to the last one */
	const int r = R0 + tid;
	const PfP pp = pf_params(q);
	const int sharp = pp.sharp, s2 = pp.s2, half = pp.half;
	const int16_t *km = kmb + (size_t)img * km_stride;
	int16_t *y = yb + (size_t)img * y_stride;
	uint8_t *so = sob + (size_t)img * so_stride;
	if (tid < 16) s_mask[tid] = reinterpret_cast<const uint32_t *>(so)[tid];
	for (int k = tid; k < MK_R * MK_BP / 4; k += MK_R) { reinterpret_cast<uint32_t *>(own)[k] = 0; reinterpret_cast<uint32_t *>(upd)[k] = 0; }
	if (tid == 0) { s_upflag[MK_R] = 0; s_updirty[MK_R] = 0; }
	__syncthreads();
	const bool in_pic = r <= W - 2;
	const bool walk_c = in_pic && !((s_mask[(r >> 5) & 15] >> (r & 31)) & 1) && !(dbg & 4);
	const bool own_row = tid < MK_R - 1 && in_pic;
	const bool have_up = r >= 2;
	CWalk cw = { 2, 0, 0, 0 };                                         /* pass C */
	MarkState no_marks = { 0, 0, 0, 0, 0, 0 };
	int p = 1;                                                         /* pass D: first pixel of the pair */

	/* The windows advance by 48 columns = 96 bytes: read window by window, three in four straddle two 128-byte lines of the contrast map and
	 * of the flag plane, and with a lane per row nothing holds a line from one window to the next (2.4 - 2.7 fetches a line).  So
	 * the row is read and sorted in line-aligned blocks of 64 cells, every block once, and a window's masks are shifted out of the two
	 * blocks it lies in. */
	const int16_t *kr = km + (size_t)r * W, *ku = km + (size_t)(r - 1) * W;            /* indexed by column: the few cells a fired rule looks at come from the plane */
	auto sort_block = [&](int bi, MkMasks &mk, uint64_t &f_lo, uint64_t &f_hi) {
		mk = MkMasks{ 0, 0, 0, 0, 0, 0, 0, 0 };
		f_lo = 0; f_hi = 0;
		if (bi >= W / 64) return;
		const int cb = 64 * bi;
		if (own_row)                                                   /* the flags of my row as passes B and C have left them so far (2 bits a cell: two masks) */
			for (int d = 0; d < 4; d++) {
				const uint4 w4 = *reinterpret_cast<const uint4 *>(so + (size_t)r * W + cb + 16 * d);
				const uint32_t w[4] = { w4.x, w4.y, w4.z, w4.w };
				for (int e = 0; e < 16; e++) { f_lo |= (uint64_t)((w[e >> 2] >> (8 * (e & 3))) & 1) << (16 * d + e); f_hi |= (uint64_t)((w[e >> 2] >> (8 * (e & 3) + 1)) & 1) << (16 * d + e); }
			}
		if (in_pic) {
			uint4 blk[8];                                                              /* the block's 64 cells: all eight loads go out together */
#pragma unroll
			for (int g = 0; g < 8; g++) blk[g] = *reinterpret_cast<const uint4 *>(kr + cb + 8 * g);
#pragma unroll
			for (int g = 0; g < 8; g++) {
				const uint4 q4 = blk[g];
				const uint32_t w4[4] = { q4.x, q4.y, q4.z, q4.w };
#pragma unroll
				for (int e = 0; e < 8; e++) {
					const int a = iabs_((int)(int16_t)((e & 1) ? w4[e >> 1] >> 16 : w4[e >> 1] & 0xFFFF));
					const uint64_t bit = 1ull << (8 * g + e);
					if (a > sharp + 20) mk.strong |= bit;
					if (a > half && a <= s2) mk.weak |= bit;
					if (a <= s2) mk.small |= bit;
					if (a > sharp && a <= sharp + 20) mk.ja |= bit;
					if (a > sharp + 56) mk.g56 |= bit;
					if (a > sharp + 160) mk.g160 |= bit;
					if (a > s2 && a <= s2 + 20) mk.jas |= bit;
					if (a > 4000) mk.big |= bit;
				}
			}
		}
	};
	MkMasks bA, bB;
	uint64_t fA_lo, fA_hi, fB_lo, fB_hi;
	int blockA = 0;
	sort_block(0, bA, fA_lo, fA_hi);
	sort_block(1, bB, fB_lo, fB_hi);
	for (int wb = 0; wb < W - 16; wb += MK_A) {                        /* windows of columns wb .. wb + 63: 0, 48, .., 480 */
		const bool last = wb + MK_A >= W - 16;
		if ((wb >> 6) != blockA) { blockA = wb >> 6; bA = bB; fA_lo = fB_lo; fA_hi = fB_hi; sort_block(blockA + 1, bB, fB_lo, fB_hi); }
		const int sh = wb & 63;
		auto win = [&](uint64_t a, uint64_t b) { return sh ? (a >> sh) | (b << (64 - sh)) : a; };
		const uint64_t f_lo = win(fA_lo, fB_lo), f_hi = win(fA_hi, fB_hi);
		int8_t *ow = own + tid * MK_BP - wb, *uw = upd + tid * MK_BP - wb;
		const MkMasks mk = { win(bA.strong, bB.strong), win(bA.weak, bB.weak), win(bA.small, bB.small), win(bA.ja, bB.ja), win(bA.g56, bB.g56), win(bA.g160, bB.g160), win(bA.jas, bB.jas), win(bA.big, bB.big) };
		MkFx fx = { kr, ku, ow, uw, wb, 0, 0, 0, 0 };
		if (walk_c) {
			CMasks cm = { mk.strong, mk.weak, mk.small, 0 };
			c_walk_window<false>(cw, no_marks, pp, cm, wb, last ? W - 3 : wb + 57, have_up, fx);   /* a block that starts at v looks at cells up to v + 6 */
		}
		uint64_t own_dirty = fx.own_dirty;
		const uint64_t own_flag = fx.own_flag, up_flag = fx.up_flag, up_dirty = fx.up_dirty;
		s_upflag[tid] = up_flag; s_updirty[tid] = up_dirty;
		__syncthreads();
		const uint64_t below_flag = s_upflag[tid + 1], below_dirty = s_updirty[tid + 1];   /* what the row below adds to mine */
		if (own_row && !(dbg & 8)) {                                   /* pass D behind pass C: cells up to wb + 53 have their final flags */
			const int pe = last ? W - 2 : wb + 53;                     /* pairs p < pe */
			const uint64_t raised = own_flag | below_flag;
			const uint64_t is1 = (f_lo & ~f_hi) | raised, is2 = f_hi & ~f_lo & ~raised, is3 = f_hi & f_lo & ~raised;
			/* bit i: one of the rules of :2322-2414 applies to the pair (wb + i, wb + i + 1); everywhere else the walk slides by one */
			const uint64_t rule = mk.big | (mk.big >> 1) | (mk.ja & (mk.ja >> 1)) | (mk.g56 & (mk.g56 >> 1)) | (mk.g160 & (mk.jas >> 1)) | ((mk.g160 >> 1) & mk.jas);
			const int16_t *krr = kr;
			while (p < pe) {
				const uint64_t ahead = bits_from(rule, p - wb);
				if (!ahead) { p = pe; break; }
				p += __builtin_ctzll(ahead);
				if (p >= pe) break;
				const int b = p - wb;
				const int f0 = ((is1 >> b) & 1) ? 1 : ((is2 >> b) & 1) ? 2 : ((is3 >> b) & 1) ? 3 : 0;
				const int f1 = ((is1 >> (b + 1)) & 1) ? 1 : ((is2 >> (b + 1)) & 1) ? 2 : ((is3 >> (b + 1)) & 1) ? 3 : 0;
				const int p0 = p;
				int d0 = 0, d1 = 0;
				p = final_pair(pp, krr, f0, f1, p0, d0, d1);
				if (d0) { ow[p0] = (int8_t)(ow[p0] + d0); own_dirty |= 1ull << b; }            /* pass D does not raise flags */
				if (d1) { ow[p0 + 1] = (int8_t)(ow[p0 + 1] + d1); own_dirty |= 1ull << (b + 1); }
			}
			if (p > pe && !last) { /* a pair that started below pe may end at pe + 1: fine, the cursor is what is carried */ }
		}
		__syncthreads();
		if (own_row) {                                                 /* the window's sums and flags go out where there are any */
			uint64_t dirty = own_dirty | below_dirty;
			const uint64_t raised = own_flag | below_flag;
			const int8_t *un = upd + (tid + 1) * MK_BP - wb;
			while (dirty) {
				const int b = __builtin_ctzll(dirty);
				dirty &= dirty - 1;
				const int col = wb + b;
				const int add = (int)ow[col] + ((below_dirty >> b) & 1 ? (int)un[col] : 0);
				const size_t at = (size_t)r * W + col;
				if (add) y[at] = (int16_t)(y[at] + add);
				ow[col] = 0;
			}
			uint64_t rs = raised;
			while (rs) { const int b = __builtin_ctzll(rs); rs &= rs - 1; so[(size_t)r * W + wb + b] = 1; }
			/* ... and into the blocks' copies of the flags, which were read before this window wrote (a raised flag reads as 1) */
			fA_lo |= raised << sh; fA_hi &= ~(raised << sh);
			if (sh) { fB_lo |= raised >> (64 - sh); fB_hi &= ~(raised >> (64 - sh)); }
		}
		__syncthreads();
		{ uint64_t ud = up_dirty; while (ud) { const int b = __builtin_ctzll(ud); ud &= ud - 1; uw[wb + b] = 0; } }   /* my additions to the row above have been taken */
		__syncthreads();
	}
}

/* pre_processing_UV (:2428-2464), pointwise on a copy: 8-neighbour Laplacian of the 256 x 256 chroma plane, one or two steps back.
 * src: the 4:2:0 byte plane; dst: the int16 plane the filterbank starts from. */
__global__ __launch_bounds__(256) void k_low_prefilter_chroma(const uint8_t *__restrict__ srcb, size_t src_stride, int16_t *__restrict__ dstb, size_t dst_stride, int q)
{
	const uint8_t *s = srcb + (size_t)blockIdx.y * src_stride;
	int16_t *d = dstb + (size_t)blockIdx.y * dst_stride;
	const int g = blockIdx.x * 256 + threadIdx.x, r = g >> 6, c4 = (g & 63) * 4;        /* four pixels of one row */
	const uint8_t *m = s + r * H + c4;
	const uint32_t wm = *reinterpret_cast<const uint32_t *>(m);
	int o[4] = { (int)(wm & 255), (int)((wm >> 8) & 255), (int)((wm >> 16) & 255), (int)(wm >> 24) };
	if (r >= 1 && r < H - 1) {
		int U[6], M[6], D[6];                                      /* columns c4-1 .. c4+4 of the rows above, at and below */
		const uint32_t wu = *reinterpret_cast<const uint32_t *>(m - H), wd = *reinterpret_cast<const uint32_t *>(m + H);
		for (int e = 0; e < 4; e++) { U[e + 1] = (wu >> (8 * e)) & 255; M[e + 1] = o[e]; D[e + 1] = (wd >> (8 * e)) & 255; }
		U[0] = c4 ? m[-H - 1] : 0; M[0] = c4 ? m[-1] : 0; D[0] = c4 ? m[H - 1] : 0;
		U[5] = c4 < H - 4 ? m[-H + 4] : 0; M[5] = c4 < H - 4 ? m[4] : 0; D[5] = c4 < H - 4 ? m[H + 4] : 0;
		for (int e = 0; e < 4; e++) {
			const int c = c4 + e;
			if (c < 1 || c > H - 2) continue;
			const int lap = (M[e + 1] << 3) - M[e] - M[e + 2] - U[e] - U[e + 1] - U[e + 2] - D[e] - D[e + 1] - D[e + 2];
			if (q < 14) {
				if (iabs_(lap) >= 14) o[e] += lap > 0 ? -2 : 2;
				else if (iabs_(lap) > 5) o[e] += lap > 0 ? -1 : 1;
			} else {
				if (lap > 5) o[e]--; else if (lap < -5) o[e]++;
			}
		}
	}
	*reinterpret_cast<uint2 *>(d + r * H + c4) = make_uint2((uint32_t)(uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16), (uint32_t)(uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16));
}

/* ------------------------------------------------------------------------------------------------ Y11 + Y12 (nhw_encoder.c:285-621) */
namespace {
__device__ static const uint8_t k_ll2_thr[13][7] = {          /* wvlt_thrx1..7 by quality (:313-381) */
	{ 0 }, { 11, 15, 10, 15, 36, 20, 21 }, { 11, 15, 10, 15, 36, 19, 20 }, { 11, 15, 10, 15, 36, 18, 18 },
	{ 11, 15, 10, 15, 36, 17, 17 }, { 11, 15, 10, 15, 36, 17, 17 }, { 11, 15, 10, 15, 36, 17, 17 },
	{ 10, 15, 9, 14, 36, 17, 17 }, { 8, 13, 6, 11, 34, 15, 15 }, { 8, 13, 6, 11, 34, 15, 15 },
	{ 8, 13, 6, 11, 34, 15, 15 }, { 8, 13, 6, 11, 34, 15, 15 }, { 8, 13, 6, 11, 34, 14, 0 } };
DEVI void zero_below(int16_t *v, int lim) { if (iabs_(*v) < lim) *v = 0; }
}
#define LS (H / 2)          /* LL2 is 128 x 128 */
#define LP (LS + 4)         /* LDS row pitch in shorts.  66 dwords: the skewed walks' lanes (row r0 + l at column t - 2 l: 65 dwords apart) each have a bank of their own, the lane-per-row walks share one between two.  At LS + 2 (65 dwords) it was the other way round, and every read of the skewed walks -- 60 % of the kernel -- was a 32-way conflict */
/* Hits are kept as four bitmaps over the LL2 cells: the level-1 children under HH1 get the largest of the limits 32 / 34 / 36 that hit the
 * cell (thrx5 is 34 or 36; the other two bands always get thrx6 and thrx6 + 6), and -- q <= 11 -- its level-2 siblings go too.  Zeroing
 * below a limit is idempotent and monotone, and the walks never read what they zero, so the walks only record hits and all lanes clear
 * the children afterwards.
 * The first walk (five cells in a row, :383-486) and the last (three flat cells, :585-620) stay inside a row: a lane per row.  The two in
 * between (:488-583) write the cell diagonally below and read it back in the next row: one raster walk each, on the scalar unit, every
 * cell's eight samples fetched in one go. */
#define HITBIT(map, cell) atomicOr(&(map)[(cell) >> 5], 1u << ((cell) & 31))
namespace {
/* The walks' hits of ONE row of the 128-cell band, kept in the registers of the lane that makes them (every walk's lane hits cells of one
 * row only) and OR-ed into the row's four map words when the lane is through: an LDS atomic a hit bit was most of a walk step's time. */
struct RowBits { uint64_t lo, hi; };
DEVI void row_set(RowBits &m, int pos, int n)                    /* cells pos .. pos + n - 1 (n <= 3, inside the row) */
{
	const uint64_t f = (1ull << n) - 1;
	if (pos < 64) { m.lo |= f << pos; if (pos + n > 64) m.hi |= f >> (64 - pos); } else m.hi |= f << (pos - 64);
}
DEVI void row_flush(uint32_t *map, int row, const RowBits &m)   /* the row's words belong to this lane until the next barrier */
{
	uint32_t *w = map + row * 4;
	if (m.lo) { w[0] |= (uint32_t)m.lo; w[1] |= (uint32_t)(m.lo >> 32); }
	if (m.hi) { w[2] |= (uint32_t)m.hi; w[3] |= (uint32_t)(m.hi >> 32); }
}
}
#define LL2_NT 256
__global__ __launch_bounds__(LL2_NT) void k_low_ll2(int16_t *__restrict__ procb, size_t plane_stride, int q)
{
	/* LDS sized at launch: the buffer and the hit maps -- three of them where thrx5 is 34 (quality 8 .. 12: the first walk's hits and the last
	 * walk's share a limit and a map), four where it is 36: 39.9 KB, four workgroups to a CU, instead of 42 KB and three */
	extern __shared__ __attribute__((aligned(16))) int16_t ll2_lds[];
	int16_t *ll = ll2_lds;
	uint32_t *h32 = reinterpret_cast<uint32_t *>(ll2_lds + LS * LP + 8), *h34 = h32 + LS * LS / 32, *hsib = h34 + LS * LS / 32;
	uint32_t *h36 = k_ll2_thr[q <= 12 ? q : 12][4] == 36 ? hsib + LS * LS / 32 : h34;
	__shared__ int stale_hits;
	__shared__ int walk_red[4][3];
	int16_t *p = procb + (size_t)blockIdx.x * plane_stride;
	const int tid = threadIdx.x, lane = tid & 63;                  /* four wavefronts: the row walks take a thread per row (128 rows), the clearing of children all 256; the two skewed raster walks stay one wavefront */

	if (q <= 11) {
		/* Y11 (:285-309): rows are independent, a thread walks one of them -- through LDS, 64 columns of all 128 rows at a time (a lane on
		 * "its" row of the plane reads one cell of a different line at every step); what the walk carries from tile to tile is the
		 * updated left neighbour. */
		const int lim = q > 6 ? 10 : 11;
		enum { TP = 68 };                                          /* tile pitch: columns c0-2 .. c0+65 as 34 dwords */
		int16_t *tile = ll;                                        /* 128 x 68 shorts: the LL2 buffer is not in use yet */
		int left = 0;
		for (int c0 = 0; c0 < H; c0 += 64) {
			for (int k = tid; k < (H / 2) * (TP / 2); k += LL2_NT) {
				const int rr = k / (TP / 2), d = k % (TP / 2);
				reinterpret_cast<uint32_t *>(tile + rr * TP)[d] = reinterpret_cast<const uint32_t *>(p + (size_t)(H / 2 + rr) * W + c0 - 2)[d];
			}
			__syncthreads();
			if (tid < H / 2) {
				int16_t *row = tile + tid * TP + 2 - c0;               /* row[j]: cell of column j */
				if (!c0) left = row[-1];                               /* the cell before the row in memory (never written by this pass) */
				int cur = row[c0];
				for (int j = c0; j < c0 + 64; j++) {
					const int nxt = row[j + 1];
					const int m = iabs_(cur);
					int out = cur;
					if (m >= DEADZONE && m < lim) {
						const bool ql = iabs_(left) < DEADZONE, qr = iabs_(nxt) < DEADZONE;
						if ((ql && qr) || (m == DEADZONE && (ql || qr))) out = 0;
					}
					if (out != cur) row[j] = (int16_t)out;
					left = out; cur = nxt;
				}
			}
			__syncthreads();
			for (int k = tid; k < (H / 2) * 32; k += LL2_NT) {
				const int rr = k >> 5, d = 1 + (k & 31);
				reinterpret_cast<uint32_t *>(p + (size_t)(H / 2 + rr) * W + c0 - 2)[d] = reinterpret_cast<const uint32_t *>(tile + rr * TP)[d];
			}
			__syncthreads();
		}
	}
	if (q > 12) return;
	for (int k = tid; k < LS * LS / 2; k += LL2_NT) {
		const int r = k >> 6, c2 = (k & 63) * 2;
		*reinterpret_cast<uint32_t *>(&ll[r * LP + c2]) = *reinterpret_cast<const uint32_t *>(p + (size_t)r * W + c2);
	}
	for (int k = tid; k < LS * LS / 32; k += LL2_NT) { h32[k] = 0; h34[k] = 0; h36[k] = 0; hsib[k] = 0; }
	if (tid == 0) stale_hits = 0;
	__syncthreads();

	const uint8_t *t = k_ll2_thr[q];
	const int t1 = t[0], t2 = t[1], t3 = t[2], t4 = t[3], t5 = t[4], t6 = t[5], t7 = t[6];
	const bool deep = q <= 11;
	uint32_t *h5 = t5 == 36 ? h36 : h34;
	/* `last`: the reference's `count` variable as the third walk finds it (:571-579 use it without having set it when the inner test fails):
	 * -1 = still IM_SIZE (no hit so far), otherwise an LL2 cell index */
	int any1 = 0;
	for (int r = tid; r < LS; r += LL2_NT) {                       /* five cells in a row (:383-486), in place along the row */
		int16_t *row = ll + r * LP;
		int v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
		RowBits hits = { 0, 0 };
		for (int j = 0; j < LS - 4; j++) {
			const int v4 = row[j + 4];
			bool h = false;
			if (iabs_(v4 - v0) < t1 && iabs_(v4 - v3) < t1 && iabs_(v1 - v0) < t1 && iabs_(v3 - v1) < t1 && iabs_(v3 - v2) < t2 - 2) {
				int n2;
				if ((v3 - v1) > 5 && (v2 - v3) >= 0) n2 = v3;
				else if ((v1 - v3) > 5 && (v2 - v3) <= 0) n2 = v3;
				else if ((v1 - v3) > 5 && (v2 - v1) >= 0) n2 = v1;
				else if ((v3 - v1) > 5 && (v2 - v1) <= 0) n2 = v1;
				else if ((v3 - v2) > 0 && (v2 - v1) > 0) n2 = v2;
				else if ((v1 - v2) > 0 && (v2 - v3) > 0) n2 = v2;
				else n2 = (v3 + v1) >> 1;
				if (n2 != v2) { v2 = n2; row[j + 2] = (int16_t)n2; }
				h = true;
			}
			else if (iabs_(v4 - v0) < t2 + 1 && iabs_(v4 - v3) < t2 + 1 && iabs_(v1 - v0) < t2 + 1) {
				if (iabs_(v3 - v1) < t2 + 6 && iabs_(v3 - v2) < t2 + 6) {
					const int a = v3 - v2, bq = v2 - v1;
					if ((a >= 0 && bq >= 0) || (a <= 0 && bq <= 0)) h = true;
				}
			}
			if (h) { row_set(hits, j + 1, 3); any1 = 1; }
			v0 = v1; v1 = v2; v2 = v3; v3 = v4;
		}
		row_flush(h5, r, hits);
		if (deep) row_flush(hsib, r, hits);
	}
	int last = __syncthreads_or(any1) ? 4 : -1;                   /* the inner loop counter's exit value: row 0, column 4 */
	/* plus shape (+2, :488-533), then flat corner (+1, :535-583): two raster walks in which a hit at (r, j) rewrites cell (r+1, j+1).
	 * Cell (r, j) reads rows r .. r+2 at columns j .. j+2: of what the walk has written before it gets there it sees (r+1, j) -- from its
	 * own left neighbour -- and row r, written by row r-1's cells up to column j+1.  So the rows run as a skewed wavefront: lane l takes
	 * row r0 + l and is at column t - 2 l in step t (its upper neighbour is two columns ahead, and reads (r+2, j+1) two steps before the
	 * row below rewrites it); 64 rows per block, 252 steps per block instead of 8000 cells one after the other.
	 * The reference's stale `count` (`last`): every hit marks the siblings of its own target (the inner test sits inside the outer one);
	 * a cell that passes only the outer test marks the target of the hit before it -- marked already -- or, before the first hit of the
	 * second walk, what the first walk left: its last hit's target, else the 4 / IM_SIZE of the row pass above. */
	int last0 = last;                                              /* `count` as the second walk finds it */
	{
		/* Both walks at once, on all four wavefronts (until round 5: one wavefront, the walks and their two blocks of rows one after the other --
		 * 1000 steps of a lone wavefront at some 1 300 cycles each, 60 % of the kernel): wavefront w takes walk w >> 1 and rows 64 (w & 1) + lane;
		 * the row of absolute index r of walk p is at column T - 2 r - LAG p in step T, a workgroup barrier a step.  The first walk's skew
		 * carries over the block seam unchanged.  The second walk may follow the first LAG = 4 steps behind: at (r, j) it reads rows r .. r + 2
		 * up to column j + 2, whose last write by the first walk -- cell (r + 2, j + 1), from (r + 1, j) -- lies two steps back; and what it
		 * writes, (r + 1, j + 1), the first walk has read for the last time -- as (r + 1, j)'s upper middle cell -- two steps before. */
		constexpr int LAG = 4;
		const int wv = tid >> 6, pass = wv >> 1, r = 64 * (wv & 1) + lane;
		const bool row_ok = r < LS - 2;
		int hit_max = -1, hit_min = 1 << 30, outer_min = 1 << 30;    /* visiting positions r * LS + j */
		/* the window's cells travel in registers: of rows r and r + 1 only the cell two columns ahead is new at a step (the row above wrote
		 * it one step ago, and writes nothing behind it), of row r + 2 the one cell looked at; what this lane's own hit writes into row r + 1 it
		 * keeps (three LDS reads a step where there were seven) */
		int a0 = 0, a1 = 0, b0 = 0, b1 = 0;
		RowBits t32 = { 0, 0 }, tsib = { 0, 0 };                      /* hits on the cells of row r + 1 */
		for (int T = 0; T < (LS - 2) + 2 * (LS - 3) + LAG; T++) {
			const int j = T - 2 * r - LAG * pass;
			if (row_ok && j >= 0 && j < LS - 2) {
				const int16_t *v = ll + r * LP + j;
				if (j == 0) { a0 = v[0]; a1 = v[1]; b0 = v[LP]; b1 = v[LP + 1]; }
				const int a2 = v[2], b2 = v[LP + 2], c1 = v[2 * LP + 1];
				bool outer, hit;
				if (!pass) { outer = false; hit = iabs_(a1 - c1) < t3 && iabs_(b0 - b2) < t3 && iabs_(b1 - b0) < t4 - 1 && iabs_(a1 - b1) < t4; }
				else {
					outer = iabs_(a2 - a1) < t3 && iabs_(a1 - a0) < t3 && iabs_(a0 - b0) < t3 && iabs_(a2 - b2) < t3;
					hit = outer && iabs_(c1 - b0) < t3 && iabs_(b0 - b1) < t4;
				}
				const int pos = r * LS + j;
				if (outer && pos < outer_min) outer_min = pos;
				if (hit) {
					const int e = (a1 + c1 + b0 + b2 + (pass ? 1 : 2)) >> 2;
					if (iabs_(e - b0) < 5 || iabs_(e - b2) < 5) { ll[(r + 1) * LP + j + 1] = (int16_t)e; b1 = (int16_t)e; }
					row_set(t32, j + 1, 1);                              /* the target (r + 1, j + 1) */
					if (deep) row_set(tsib, j, 3);
					if (pos > hit_max) hit_max = pos;
					if (pos < hit_min) hit_min = pos;
				}
				a0 = a1; a1 = a2; b0 = b1; b1 = b2;
			}
			__syncthreads();
		}
		if (row_ok) {                                              /* (both walks hit row r + 1: atomics here, one a word and lane) */
			uint32_t *w32 = h32 + (r + 1) * 4, *wsb = hsib + (r + 1) * 4;
			const uint32_t m32[4] = { (uint32_t)t32.lo, (uint32_t)(t32.lo >> 32), (uint32_t)t32.hi, (uint32_t)(t32.hi >> 32) };
			const uint32_t msb[4] = { (uint32_t)tsib.lo, (uint32_t)(tsib.lo >> 32), (uint32_t)tsib.hi, (uint32_t)(tsib.hi >> 32) };
			for (int k = 0; k < 4; k++) { if (m32[k]) atomicOr(&w32[k], m32[k]); if (deep && msb[k]) atomicOr(&wsb[k], msb[k]); }
		}
		for (int d = 32; d; d >>= 1) {                               /* over a wavefront's rows */
			const int a = __shfl_xor(hit_max, d), bq = __shfl_xor(hit_min, d), cq = __shfl_xor(outer_min, d);
			hit_max = a > hit_max ? a : hit_max; hit_min = bq < hit_min ? bq : hit_min; outer_min = cq < outer_min ? cq : outer_min;
		}
		if (lane == 0) { walk_red[wv][0] = hit_max; walk_red[wv][1] = hit_min; walk_red[wv][2] = outer_min; }
		__syncthreads();
		const int hm0 = walk_red[0][0] > walk_red[1][0] ? walk_red[0][0] : walk_red[1][0];
		const int hmin1 = walk_red[2][1] < walk_red[3][1] ? walk_red[2][1] : walk_red[3][1], omin1 = walk_red[2][2] < walk_red[3][2] ? walk_red[2][2] : walk_red[3][2];
		if (hm0 >= 0) last0 = hm0 + LS + 1;                           /* target of the first walk's last hit: (r + 1) * LS + j + 1 */
		if (deep && omin1 < hmin1) {                                 /* cells that pass the outer test before any hit of the second walk */
			if (last0 < 0) { if (tid == 0) stale_hits = 1; }
			else if (tid < 3) HITBIT(hsib, last0 - 1 + tid);
		}
	}
	__syncthreads();
	if (deep)
		for (int r = tid; r < LS; r += LL2_NT) {                   /* three flat cells in a row (:585-620): reads only */
			const int16_t *row = ll + r * LP;
			RowBits hits = { 0, 0 };
			int x0 = row[0], x1 = row[1];
			for (int j = 0; j < LS - 2; j++) {
				const int x2 = row[j + 2];
				if (iabs_(x2 - x1) < t7 && iabs_(x2 - x0) < t7 && iabs_(x1 - x0) < t7) row_set(hits, j + 1, 1);
				x0 = x1; x1 = x2;
			}
			row_flush(h34, r, hits); row_flush(hsib, r, hits);
		}
	__syncthreads();
	for (int k = tid; k < LS * LS / 2; k += LL2_NT) {             /* the smoothed LL2 band goes back */
		const int r = k >> 6, c2 = (k & 63) * 2;
		*reinterpret_cast<uint32_t *>(p + (size_t)r * W + c2) = *reinterpret_cast<const uint32_t *>(&ll[r * LP + c2]);
	}
	/* the children / siblings of the cells that were hit are cleared where they are small.  All loads of a cell's up to fifteen targets
	 * are issued before the first store (one memory round trip per cell instead of fifteen in a row: this loop was most of the kernel). */
	for (int cell = tid; cell < LS * LS; cell += LL2_NT) {
		const int w = cell >> 5;
		const uint32_t bit = 1u << (cell & 31);
		const int limc = (t5 == 36 && (h36[w] & bit)) ? 36 : (h34[w] & bit) ? 34 : (h32[w] & bit) ? 32 : 0;   /* (thrx5 34: h36 is h34) */
		const bool sib = hsib[w] & bit;
		if (!limc && !sib) continue;
		const int r = cell >> 7, j = cell & 127, flat = r * W + j;
		uint32_t cw[6] = { 0, 0, 0, 0, 0, 0 };
		int sv[3] = { 0, 0, 0 };
		int16_t *sp[3] = { p + flat + H / 2, p + flat + Q, p + flat + Q + H / 2 };
		const int band[3] = { H, 2 * Q, 2 * Q + H };
		if (limc)
			for (int bnd = 0; bnd < 3; bnd++) {
				const uint32_t *v = reinterpret_cast<const uint32_t *>(p + (flat << 1) + band[bnd]);
				cw[2 * bnd] = v[0]; cw[2 * bnd + 1] = v[W / 2];
			}
		if (sib) for (int k = 0; k < 3; k++) sv[k] = *sp[k];
		if (limc) {
			const int lim[3] = { t6, t6 + 6, limc };
			for (int bnd = 0; bnd < 3; bnd++)
				for (int h = 0; h < 2; h++) {
					const uint32_t x = cw[2 * bnd + h];
					uint32_t y = x;
					if (iabs_((int16_t)(x & 0xFFFF)) < lim[bnd]) y &= 0xFFFF0000u;
					if (iabs_((int16_t)(x >> 16)) < lim[bnd]) y &= 0x0000FFFFu;
					if (y != x) reinterpret_cast<uint32_t *>(p + (flat << 1) + band[bnd])[h ? W / 2 : 0] = y;
				}
		}
		if (sib) { const int sl[3] = { 11, 12, 13 }; for (int k = 0; k < 3; k++) if (sv[k] && iabs_(sv[k]) < sl[k]) *sp[k] = 0; }
	}
	if (stale_hits && tid < 3) {                                  /* `count` still IM_SIZE: the "siblings" of plane cells 65535..65537 */
		const int flat = Q - 1 + tid;
		zero_below(p + flat + H / 2, 11); zero_below(p + flat + Q, 12); zero_below(p + flat + Q + H / 2, 13);
	}
}
#undef HITBIT
#undef LP
#undef LS

/* level-1 chroma detail below 24 / 32 / 48 goes (nhw_encoder.c:2277-2308, :2590-2621; q <= 16), pointwise on the 256 x 256 coefficient plane */
__global__ __launch_bounds__(256) void k_low_chroma_thin(int16_t *__restrict__ planeb, size_t plane_stride)
{
	int16_t *p = planeb + (size_t)blockIdx.y * plane_stride;
	const int g = blockIdx.x * 256 + threadIdx.x;                  /* 8 cells */
	const int r = g >> 5, c0 = (g & 31) * 8;
	if (r < H / 2 && c0 < H / 2) return;
	const int lim = r < H / 2 ? 24 : (c0 < H / 2 ? 32 : 48);
	uint4 w = *reinterpret_cast<uint4 *>(p + r * H + c0);
	uint32_t ww[4] = { w.x, w.y, w.z, w.w };
	for (int e = 0; e < 4; e++) {
		const int lo = (int16_t)(ww[e] & 0xFFFF), hi = (int16_t)(ww[e] >> 16);
		if (iabs_(lo) >= DEADZONE && iabs_(lo) < lim) ww[e] &= 0xFFFF0000u;
		if (iabs_(hi) >= DEADZONE && iabs_(hi) < lim) ww[e] &= 0x0000FFFFu;
	}
	*reinterpret_cast<uint4 *>(p + r * H + c0) = make_uint4(ww[0], ww[1], ww[2], ww[3]);
}
void nhw_launch_low_chroma_thin(int16_t *plane, size_t plane_stride, int n, hipStream_t s)
{
	k_low_chroma_thin<<<dim3(Q / 8 / 256, n), 256, 0, s>>>(plane, plane_stride);
}

void nhw_launch_low_ll2(int16_t *proc, size_t plane_stride, int q, int n, hipStream_t s)
{
	const int maps = (q <= 7) ? 4 : 3;                               /* k_ll2_thr[q][4] == 36 up to quality 7 */
	k_low_ll2<<<n, LL2_NT, (size_t)(128 * 132 + 8) * 2 + (size_t)maps * (128 * 128 / 32) * 4, s>>>(proc, plane_stride, q);   /* 128 rows at the kernel's pitch LP + the hit maps */
}
/* The pre-filter of a batch.  parts > 1: the batch is cut into sub-batches whose sequences run on streams of their own (aux), the next one's
 * pass A starting when the one before has finished its own -- the chain (k_low_chain) is two wavefronts a picture that live on the scalar
 * unit and on look-ups, and leaves the vector units and the memory system to the streaming kernels of the sub-batches before and behind it
 * (pass A, the answers' application, passes C and D).  ev: 1 + 2 * parts events.  Returns a hipError_t. */
int nhw_launch_low_prefilter(const int16_t *src, size_t src_stride, int16_t *y, size_t y_stride, int16_t *km, size_t km_stride, uint8_t *so, size_t so_stride,
                             uint8_t *chain /* CH_BYTES an image: the machine's answers */, size_t chain_stride,
                             uint16_t *tab /* MASK_ROW * W bytes an image: pass A's candidate masks */, size_t tab_stride, int q, int n, hipStream_t s, int force /* 32: the bands of pass A go back three rows for their entry state (tests) */,
                             int parts, hipStream_t *aux, hipEvent_t *ev)
{
	static int dbg = 0;
#ifdef NHW_DEV   /* developer builds: bits that switch passes of the pre-filter off for timing */
	{ const char *e = getenv("NHW_LOW_DBG"); dbg = e ? atoi(e) : 0; }
#endif
	auto head = [&](int i0, int m, hipStream_t st) {                    /* pass A of images i0 .. i0 + m - 1 */
		k_low_pre<<<dim3(PRE_NB, m), 64, 0, st>>>(src + (size_t)i0 * src_stride, src_stride, km + (size_t)i0 * km_stride, km_stride, reinterpret_cast<uint8_t *>(tab) + (size_t)i0 * tab_stride, tab_stride, q, dbg | force);
		k_low_mapfix<<<m, 64, 0, st>>>(km + (size_t)i0 * km_stride, km_stride, reinterpret_cast<const uint8_t *>(tab) + (size_t)i0 * tab_stride, tab_stride, so + (size_t)i0 * so_stride, so_stride, q, dbg);
	};
	auto machine = [&](int i0, int m, hipStream_t st) {                 /* pass B's machine */
		k_low_chain<<<m, 128, 0, st>>>(km + (size_t)i0 * km_stride, km_stride, chain + (size_t)i0 * chain_stride, chain_stride, q, dbg);
	};
	auto rest = [&](int i0, int m, hipStream_t st) {                    /* pass B's picture side, passes C and D */
		k_low_apply<<<dim3(PRE_NB, m), 64, 0, st>>>(src + (size_t)i0 * src_stride, src_stride, y + (size_t)i0 * y_stride, y_stride, km + (size_t)i0 * km_stride, km_stride, so + (size_t)i0 * so_stride, so_stride,
		                                           chain + (size_t)i0 * chain_stride, chain_stride, q, dbg);
		k_low_markrows<<<m, 64, 0, st>>>(y + (size_t)i0 * y_stride, y_stride, km + (size_t)i0 * km_stride, km_stride, so + (size_t)i0 * so_stride, so_stride, q);
		k_low_marks<<<dim3((W - 2 + MK_R - 2) / (MK_R - 1), m), MK_R, 0, st>>>(y + (size_t)i0 * y_stride, y_stride, km + (size_t)i0 * km_stride, km_stride, so + (size_t)i0 * so_stride, so_stride, q, dbg);
	};
	if (parts <= 1 || !aux || !ev) { head(0, n, s); machine(0, n, s); rest(0, n, s); return (int)hipGetLastError(); }
#define LOWCHK(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
	LOWCHK(hipEventRecord(ev[0], s));
	for (int p = 0; p < parts; p++) {
		const int i0 = (int)((long long)n * p / parts), i1 = (int)((long long)n * (p + 1) / parts);
		LOWCHK(hipStreamWaitEvent(aux[p], ev[0], 0));
		if (p) LOWCHK(hipStreamWaitEvent(aux[p], ev[p], 0));              /* the sub-batch before is through its pass A */
		head(i0, i1 - i0, aux[p]);
		LOWCHK(hipEventRecord(ev[1 + p], aux[p]));
		machine(i0, i1 - i0, aux[p]);
		rest(i0, i1 - i0, aux[p]);
		LOWCHK(hipEventRecord(ev[1 + parts + p], aux[p]));
	}
	for (int p = 0; p < parts; p++) LOWCHK(hipStreamWaitEvent(s, ev[1 + parts + p], 0));
#undef LOWCHK
	return (int)hipGetLastError();
}
/* Compatibility mode (NHW_COMPAT_GLIBC_ONESHOT) only, quality <= 16: the contrast-map cells whose memory the stock binary's malloc hands
 * out again -- res256's slack (row 128, columns 0..3), tree1 (map bytes from 262176 on: rows 272..280 hold what is read before it is
 * written) and resIII's slack (row 256, columns 8..11) -- copied out of the map plane before the band kernel takes the plane over.
 * out: [0..3] row 128, 9 rows of 512, [4 + 9*512 ..+3] row 256 (the first two as k_front_stale lays them out for quality 17..21). */
__global__ void k_low_stale(const int16_t *__restrict__ kmb, size_t km_stride, int16_t *__restrict__ stale, size_t stale_stride)
{
	const int16_t *km = kmb + (size_t)blockIdx.x * km_stride;
	int16_t *out = (int16_t *)((uint8_t *)stale + (size_t)blockIdx.x * stale_stride);
	const int t = threadIdx.x;
	for (int i = t; i < 9 * W / 4; i += 256) reinterpret_cast<uint2 *>(out + 4)[i] = reinterpret_cast<const uint2 *>(km + (size_t)272 * W)[i];
	if (t < 4) { out[t] = km[(size_t)128 * W + t]; out[4 + 9 * W + t] = km[(size_t)256 * W + 8 + t]; }
}
void nhw_launch_low_stale(const int16_t *km, size_t km_stride, int16_t *stale, size_t stale_stride, int n, hipStream_t s)
{
	k_low_stale<<<n, 256, 0, s>>>(km, km_stride, stale, stale_stride);
}
void nhw_launch_low_prefilter_chroma(const uint8_t *src, size_t src_stride, int16_t *dst, size_t dst_stride, int q, int n, hipStream_t s)
{
	k_low_prefilter_chroma<<<dim3(Q / 4 / 256, n), 256, 0, s>>>(src, src_stride, dst, dst_stride, q);
}
