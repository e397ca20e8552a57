#include <type_traits>
/*
 * nhw_tail_wave.h -- one wavefront per image: the raster-serial passes as row-sequential, column-parallel walks.
 *
 * The reference walks these bands cell by cell in raster order, and a cell's action depends on what the walk
 * did just before it (skip the partner of a marked pair, a sample bumped by its left neighbour, ...) and on what
 * the row above wrote into this row.  Two observations make them data-parallel along a row:
 *   * within a row the walk's memory is tiny (skip the next cell / this cell was bumped) and the marks it leaves
 *     never overlap the cells a later step of the same row reads, so "would fire if visited" is a pure function
 *     of the row's values before the walk: one bit per cell, gathered with __ballot into a 256-bit row mask;
 *   * "visited" then follows from the fire bits alone: in a run of consecutive fire bits every second cell is
 *     visited (the closed form alt_runs(), an add-with-carry over the mask), or, for mixed skip lengths, a
 *     scalar loop over the set bits.
 * The row masks are wave-uniform, so all of that is scalar-unit work; the lanes only classify their own cells
 * (lane l owns columns l, l+64, l+128, l+192 of the row: bit l of mask word k) and apply the marks.  Rows stay
 * sequential -- a row writes into the next one -- but a step is a few hundred instructions on registers, with
 * the next rows already in flight, instead of one memory round trip per cell.  Four images share a 256-thread
 * workgroup, one per wavefront; there are no workgroup barriers in here.
 */
#ifndef NHW_TAIL_WAVE_H
#define NHW_TAIL_WAVE_H

#include "nhw_tail_dev.h"

namespace nhw {

struct M4 { uint64_t w[4]; };
DEV M4 operator&(M4 a, M4 b) { return M4{ { a.w[0] & b.w[0], a.w[1] & b.w[1], a.w[2] & b.w[2], a.w[3] & b.w[3] } }; }
DEV M4 operator|(M4 a, M4 b) { return M4{ { a.w[0] | b.w[0], a.w[1] | b.w[1], a.w[2] | b.w[2], a.w[3] | b.w[3] } }; }
DEV M4 operator~(M4 a) { return M4{ { ~a.w[0], ~a.w[1], ~a.w[2], ~a.w[3] } }; }
DEV M4 m4_zero() { return M4{ { 0, 0, 0, 0 } }; }
/* up(m)[j] = m[j-1] (the bit moves to the next column), dn(m)[j] = m[j+1] */
DEV M4 up1(M4 a) { return M4{ { a.w[0] << 1, (a.w[1] << 1) | (a.w[0] >> 63), (a.w[2] << 1) | (a.w[1] >> 63), (a.w[3] << 1) | (a.w[2] >> 63) } }; }
DEV M4 dn1(M4 a) { return M4{ { (a.w[0] >> 1) | (a.w[1] << 63), (a.w[1] >> 1) | (a.w[2] << 63), (a.w[2] >> 1) | (a.w[3] << 63), a.w[3] >> 1 } }; }
DEV M4 dn2(M4 a) { return M4{ { (a.w[0] >> 2) | (a.w[1] << 62), (a.w[1] >> 2) | (a.w[2] << 62), (a.w[2] >> 2) | (a.w[3] << 62), a.w[3] >> 2 } }; }
DEV M4 dn3(M4 a) { return M4{ { (a.w[0] >> 3) | (a.w[1] << 61), (a.w[1] >> 3) | (a.w[2] << 61), (a.w[2] >> 3) | (a.w[3] << 61), a.w[3] >> 3 } }; }
DEV uint64_t low_bits(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1)); }
DEV M4 col_range(int lo, int hi)                                /* columns lo..hi inclusive */
{
	M4 r;
	r.w[0] = low_bits(hi + 1) & ~low_bits(lo);
	r.w[1] = low_bits(hi + 1 - 64) & ~low_bits(lo - 64);
	r.w[2] = low_bits(hi + 1 - 128) & ~low_bits(lo - 128);
	r.w[3] = low_bits(hi + 1 - 192) & ~low_bits(lo - 192);
	return r;
}
/* bit `lane` of a wave-uniform mask word as a per-lane condition: the mask itself becomes the select operand (v_cndmask with an SGPR pair) */
#define TB(m, k) ((int)__builtin_amdgcn_inverse_ballot_w64((m).w[k]))
#define BALLOT4(m, arr, expr) do { { const int x = arr[0]; (m).w[0] = __ballot(expr); } { const int x = arr[1]; (m).w[1] = __ballot(expr); } \
	{ const int x = arr[2]; (m).w[2] = __ballot(expr); } { const int x = arr[3]; (m).w[3] = __ballot(expr); } } while (0)

/* Bit-sliced rows.  The scalar unit is one per CU: a row mask algebra of a few hundred scalar instructions per step, times 16 resident
 * wavefronts, is what bounded these kernels.  So a lane also keeps its cells' booleans as a small bit field (bit k = cell lane + 64k):
 * one vector instruction combines a whole row, a shift by one column is a wave-wide DPP move plus the word seam, and the scalar unit only
 * sees the ballots that feed a run resolution (alt_runs) and what comes back from it. */
template <int N> DEV unsigned bs_up(unsigned b, int lane, unsigned in0 = 0)          /* cell j takes the bit of cell j - 1 (in0: the bit left of column 0) */
{
	const unsigned seam = (((unsigned)__builtin_amdgcn_readlane((int)b, 63) << 1) | in0) & ((1u << N) - 1);
	(void)lane;
	return (unsigned)__builtin_amdgcn_update_dpp((int)seam, (int)b, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);   /* lane 0 has no source and keeps the seam */
}
DEV unsigned bs_dn(unsigned b, int lane)                                             /* cell j takes the bit of cell j + 1 (0 behind the last column) */
{
	const unsigned seam = (unsigned)__builtin_amdgcn_readlane((int)b, 0) >> 1;
	(void)lane;
	return (unsigned)__builtin_amdgcn_update_dpp((int)seam, (int)b, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);   /* lane 63 has no source and keeps the seam */
}
#define BS_PREDK(b, arr, k0, expr) do { (b) = 0; for (int k_ = (k0); k_ < 4; k_++) { const int x = (arr)[k_]; (b) |= ((expr) ? 1u : 0u) << k_; } } while (0)
#define BS_PRED(b, arr, n, expr) do { (b) = 0; for (int k_ = 0; k_ < (n); k_++) { const int x = (arr)[k_]; (b) |= ((expr) ? 1u : 0u) << k_; } } while (0)
DEV M4 bs_ballot4(unsigned b) { return M4{ { __ballot(b & 1), __ballot(b & 2), __ballot(b & 4), __ballot(b & 8) } }; }
DEV unsigned bs_from4(M4 m) { unsigned b = 0; for (int k = 0; k < 4; k++) b |= (unsigned)__builtin_amdgcn_inverse_ballot_w64(m.w[k]) << k; return b; }

/* A walk that skips the cell after every cell where it fires: given "fires if visited" per cell, the cells where
 * it does fire.  Inside a run of consecutive fire bits those are the cells at even distance from the run's
 * first cell.  Adding a 1 at every run start that sits on an even column ripples through exactly those runs. */
DEV M4 alt_runs(M4 f)
{
	const uint64_t even = 0x5555555555555555ull;
	const M4 pf = up1(f);
	uint64_t sum[4];
	unsigned carry = 0;
	for (int k = 0; k < 4; k++) {
		const uint64_t se = f.w[k] & ~pf.w[k] & even;
		const uint64_t t = f.w[k] + se;
		const unsigned c1 = t < se;
		sum[k] = t + carry;
		carry = c1 | (sum[k] < t);
	}
	M4 r;
	for (int k = 0; k < 4; k++) {
		const uint64_t re = f.w[k] & ~sum[k];                       /* cells of runs that start on an even column */
		r.w[k] = (re & even) | (f.w[k] & ~re & ~even);
	}
	return r;
}

/* value of the cell d columns to the right (d = 1..3) for every cell of a row held as v[k] = column lane + 64k */
DEV int right_of(const int *v, int k, int nk, int d, int lane)
{
	const int a = __shfl(v[k], (lane + d) & 63);
	const int b = k + 1 < nk ? __shfl(v[k + 1], (lane + d) & 63) : 0;
	return lane + d < 64 ? a : b;
}

DEV int ll2_round(int v) { return (v > 0 && v < 256) ? (v & 0xFFFE) : v; }

/* ------------------------------------------------------------------------------------------------------------
 * LL2 part of offsetY_recons256 (image_processing.c:2609-2735): tag runs of four odd samples, the walk that
 * bumps the next sample / the sample below, and the rounded copy into the reconstruction plane.
 *
 * The +16000 tags are kept as masks (values stay untagged in registers): a tag matters only as "tagged cells do
 * not fire, and in the first loop skip their right neighbour", in the |s - s2| > 1 test (true against a tagged
 * partner) and in the "< 10000" test of the sample below.  A row's final value is known when its own step
 * ends (the row above has bumped it, its own walk has bumped it), so the untagging / rounding pass (:2697-2735)
 * is folded into the step: p = value, jp = tagged ? value : rounded value, in both loops.
 * ------------------------------------------------------------------------------------------------------------ */
DEV void ll2_load_row(const int16_t *p, int r, int lane, int *v, M4 *tag, int q, int part, M4 *starts = nullptr, int ps = W /* row pitch of the plane the rows come from */)
{
	if (starts) *starts = m4_zero();
	if (r >= H / 2) { v[0] = v[1] = v[2] = 0; *tag = m4_zero(); return; }
	for (int k = 0; k < 3; k++) v[k] = p[r * ps + lane + 64 * k];
	M4 t = m4_zero();
	if (q > 17) {                                                  /* :2609-2640 */
		int v4[4] = { v[0], v[1], v[2], 0 };
		M4 o, d3;
		BALLOT4(o, v4, x & 1);
		int far[4];
		for (int k = 0; k < 2; k++) far[k] = right_of(v, k, 3, 3, lane);
		d3.w[0] = __ballot(iabs(v[0] - far[0]) > 1); d3.w[1] = __ballot(iabs(v[1] - far[1]) > 1); d3.w[2] = d3.w[3] = 0;
		const M4 f = o & dn1(o) & dn2(o) & dn3(o) & d3 & col_range(0, H / 2 - 4);
		unsigned __int128 m = ((unsigned __int128)f.w[1] << 64) | f.w[0], tg = 0;
		const unsigned __int128 pat = part ? 5 : 15;
		unsigned __int128 st = 0;
		while (m) {
			const uint64_t lo = (uint64_t)m;
			const int j = lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll((uint64_t)(m >> 64));
			tg |= pat << j; st |= (unsigned __int128)1 << j;
			m &= ~((unsigned __int128)15 << j);
		}
		t.w[0] = (uint64_t)tg; t.w[1] = (uint64_t)(tg >> 64);
		if (starts) { starts->w[0] = (uint64_t)st; starts->w[1] = (uint64_t)(st >> 64); }
	}
	*tag = t;
}

DEV void wave_ll2(Ctx *c, int part, int lane, bool keep_p, const int16_t *src, int ss /* where the rows are read: the work plane (pitch W), or -- second closed loop, production -- the level-2 block's copy l2save (pitch H): Y17 (nhw_encoder.c:749-755) restored the block into the work plane for this reader alone */)
{
	int16_t *p = c->proc, *jp = c->jpeg;
	const int q = c->q;
	int v0[3], v1[3], v2[3], v3[3];
	M4 t0, t1, t2, t3;
	ll2_load_row(src, 0, lane, v0, &t0, q, part, nullptr, ss);
	ll2_load_row(src, 1, lane, v1, &t1, q, part, nullptr, ss);
	ll2_load_row(src, 2, lane, v2, &t2, q, part, nullptr, ss);
	ll2_load_row(src, 3, lane, v3, &t3, q, part, nullptr, ss);
	const M4 ll = col_range(0, H / 2 - 1);
	for (int r = 0; r < H / 2; r++) {
		int vn[3]; M4 tn;
		ll2_load_row(src, r + 4, lane, vn, &tn, q, part, nullptr, ss);
		if (q > 17) {
			int a0[4] = { v0[0], v0[1], v0[2], 0 }, a1[4] = { v1[0], v1[1], v1[2], 0 }, a2[4] = { v2[0], v2[1], 0, 0 }, a3[4] = { v3[0], v3[1], 0, 0 };
			M4 o, o1, o2, o3, d2;
			BALLOT4(o, a0, x & 1); BALLOT4(o1, a1, x & 1); BALLOT4(o2, a2, x & 1); BALLOT4(o3, a3, x & 1);
			const int f0 = right_of(v0, 0, 3, 2, lane), f1 = right_of(v0, 1, 3, 2, lane);
			d2.w[0] = __ballot(iabs(v0[0] - f0) > 1); d2.w[1] = __ballot(iabs(v0[1] - f1) > 1); d2.w[2] = d2.w[3] = 0;
			d2 = d2 | dn2(t0);                                         /* a tagged partner is 16000 away */
			const M4 cond1 = o & dn1(o) & col_range(1, H / 2 - 1);
			const M4 hbr = cond1 & dn2(o) & col_range(0, H / 2 - 3);
			const M4 skip = part ? up1(t0) : m4_zero();
			const M4 act = ll & ~t0 & ~skip;
			const M4 fired = alt_runs(hbr & d2 & act);                 /* bumps the next sample */
			const M4 bumped = up1(fired);
			M4 vf = m4_zero();
			if (r <= H / 2 - 2) vf = cond1 & ~hbr & o1 & dn1(o1) & ~dn2(o1);
			if (r >= 1 && r <= H / 2 - 4) vf = vf | (~cond1 & o & o1 & dn1(o1) & o2 & ~o3);
			vf = vf & act & ~bumped & ~t1;                             /* bumps the sample below (if it is not tagged) */
			for (int k = 0; k < 2; k++) { v0[k] += TB(bumped, k); v1[k] += TB(vf, k); }
		}
		for (int k = 0; k < 2; k++) {
			const int at = r * W + lane + 64 * k;
			if (keep_p || !part) p[at] = (int16_t)v0[k];              /* (second loop: the verbatim samples below read them back) */
			jp[at] = (int16_t)(TB(t0, k) ? v0[k] : ll2_round(v0[k]));
		}
		for (int k = 0; k < 3; k++) { v0[k] = v1[k]; v1[k] = v2[k]; v2[k] = v3[k]; v3[k] = vn[k]; }
		t0 = t1; t1 = t2; t2 = t3; t3 = tn;
	}
	if (!part && !c->defer_verbatim) {                             /* samples the LL coder sent verbatim keep their exact value (:2728-2735); production: the synthesis behind this pass does it (k_dwt_syn), the coder runs beside this kernel */
		__threadfence_block();
		const int nm = c->m->ll_mem_len;
		for (int i = lane; i < nm; i += 64) {
			const int idx = c->ll_mem[i], pos = ((idx >> 7) << 9) + (idx & 127);
			jp[pos] = p[pos];
		}
	}
	__threadfence_block();
}

/* Y14 + Y15 (nhw_encoder.c:640-741): tag the runs of four odd LL2 samples (their first columns go to the res4 list),
 * the same bump walk as above, and the emission of the samples as bytes: a sample outside 0..255 goes to the
 * exception list and repeats the byte before it in the stream.  The band is left zero. */
DEV void wave_emit_ll2(Ctx *c, int lane)
{
	int16_t *p = c->proc;
	const int q = c->q;
	int v0[3], v1[3], v2[3], v3[3];
	M4 t0, t1, t2, t3, s0, s1, s2, s3;
	ll2_load_row(p, 0, lane, v0, &t0, q, 0, &s0);
	ll2_load_row(p, 1, lane, v1, &t1, q, 0, &s1);
	ll2_load_row(p, 2, lane, v2, &t2, q, 0, &s2);
	ll2_load_row(p, 3, lane, v3, &t3, q, 0, &s3);
	const M4 ll = col_range(0, H / 2 - 1);
	int n4 = 0, e = 0, carry = 0;
	for (int r = 0; r < H / 2; r++) {
		int vn[3]; M4 tn, sn;
		ll2_load_row(p, r + 4, lane, vn, &tn, q, 0, &sn);
		if (q > 17) {
			int a0[4] = { v0[0], v0[1], v0[2], 0 }, a1[4] = { v1[0], v1[1], v1[2], 0 }, a2[4] = { v2[0], v2[1], 0, 0 }, a3[4] = { v3[0], v3[1], 0, 0 };
			M4 o, o1, o2, o3, d2;
			BALLOT4(o, a0, x & 1); BALLOT4(o1, a1, x & 1); BALLOT4(o2, a2, x & 1); BALLOT4(o3, a3, x & 1);
			const int f0 = right_of(v0, 0, 3, 2, lane), f1 = right_of(v0, 1, 3, 2, lane);
			d2.w[0] = __ballot(iabs(v0[0] - f0) > 1); d2.w[1] = __ballot(iabs(v0[1] - f1) > 1); d2.w[2] = d2.w[3] = 0;
			d2 = d2 | dn2(t0);
			const M4 cond1 = o & dn1(o) & col_range(1, H / 2 - 1);
			const M4 hbr = cond1 & dn2(o) & col_range(0, H / 2 - 3);
			const M4 act = ll & ~t0;
			const M4 bumped = up1(alt_runs(hbr & d2 & act));
			M4 vf = m4_zero();
			if (r <= H / 2 - 2) vf = cond1 & ~hbr & o1 & dn1(o1) & ~dn2(o1);
			if (r >= 1 && r <= H / 2 - 4) vf = vf | (~cond1 & o & o1 & dn1(o1) & o2 & ~o3);
			vf = vf & act & ~bumped & ~t1;
			/* the emitted value of a cell is the one the walk finds there: bumps by its left neighbour / the row above are in, its own firing is not */
			for (int k = 0; k < 2; k++) v1[k] += TB(vf, k);
			for (int k = 0; k < 2; k++) v0[k] += TB(bumped, k);
		}
		uint64_t x[2], g[2];
		int byte[2];
		for (int k = 0; k < 2; k++) {
			const int s = v0[k];
			x[k] = __ballot(s > 255 || s < 0);
			byte[k] = s > 255 ? 255 : (s < 0 ? 0 : s);
		}
		if (r == 0) x[0] &= ~1ull;
		g[0] = ~x[0]; g[1] = ~x[1];
		for (int k = 0; k < 2; k++) {
			const int col = lane + 64 * k, a = r * (H / 2) + col;
			const bool exc = (x[k] >> lane) & 1;
			const uint64_t hi = k ? (g[1] & low_bits(lane)) : 0, lo = k ? g[0] : (g[0] & low_bits(lane));
			const int src = hi ? 64 + 63 - __builtin_clzll(hi) : (lo ? 63 - __builtin_clzll(lo) : -1);   /* nearest in-range sample before me in this row */
			const int b0 = __shfl(byte[0], src & 63), b1 = __shfl(byte[1], src & 63);
			const int prev = src < 0 ? carry : (((src >> 6) ? b1 : b0) & 254);
			if (exc) {
				const int s = v0[k], mag = s > 255 ? s - 255 : -s;
				const int rank = __popcll(x[0] & (k ? ~0ull : low_bits(lane))) + (k ? __popcll(x[1] & low_bits(lane)) : 0);
				uint8_t *ex = c->exw + e + 3 * rank;
				ex[0] = (uint8_t)r; ex[1] = (uint8_t)(col + (s > 255 ? 128 : 0)); ex[2] = (uint8_t)(mag > 255 ? 255 : mag);
				c->ll_bytes[a] = (uint8_t)prev; c->ll_full[a] = (uint8_t)prev;
			} else { c->ll_full[a] = (uint8_t)byte[k]; c->ll_bytes[a] = (uint8_t)(byte[k] & 254); }
			p[r * W + col] = 0;
		}
		e += 3 * (__popcll(x[0]) + __popcll(x[1]));
		if (g[1]) carry = __shfl(byte[1], 63 - __builtin_clzll(g[1])) & 254;
		else if (g[0]) carry = __shfl(byte[0], 63 - __builtin_clzll(g[0])) & 254;
		if (q > 17) {                                              /* res4: first column + 1 of every tagged run, the row's last entry flagged (a lone flag if the row has none) */
			const int cnt = __popcll(s0.w[0]) + __popcll(s0.w[1]);
			if (!cnt) { if (lane == 0) c->res4[n4] = 128; n4++; }
			else {
				for (int k = 0; k < 2; k++)
					if ((s0.w[k] >> lane) & 1) {
						const int rank = __popcll(s0.w[0] & (k ? ~0ull : low_bits(lane))) + (k ? __popcll(s0.w[1] & low_bits(lane)) : 0);
						c->res4[n4 + rank] = (uint8_t)(lane + 64 * k + 1 + (rank == cnt - 1 ? 128 : 0));
					}
				n4 += cnt;
			}
		}
		for (int k = 0; k < 3; k++) { v0[k] = v1[k]; v1[k] = v2[k]; v2[k] = v3[k]; v3[k] = vn[k]; }
		t0 = t1; t1 = t2; t2 = t3; t3 = tn;
		s0 = s1; s1 = s2; s2 = s3; s3 = sn;
	}
	if (lane == 0) { c->m->res4_len = q > 17 ? n4 : 0; c->m->exw_len = e; }
}

/* ------------------------------------------------------------------------------------------------------------
 * detail bands of offsetY_recons256 (image_processing.c:2759-3124): triple / vertical-pair marking (writes into
 * the next row), equal-sign 5..7 pairs, and the per-row dequantiser with its next-cell fix-ups.
 * ------------------------------------------------------------------------------------------------------------ */
DEV void det_load_row(const int16_t *p, int r, int lane, int *v, int ps = W)
{
	if (r >= H) { v[0] = v[1] = v[2] = v[3] = 0; return; }
	const bool top = r < H / 2;                                    /* rows of the LL2 | HL2 half: only HL2 (columns >= 128) belongs to this pass */
	v[0] = top ? 0 : p[r * ps + lane];
	v[1] = top ? 0 : p[r * ps + lane + 64];
	v[2] = p[r * ps + lane + 128];
	v[3] = p[r * ps + lane + 192];
}

/* columns lo..hi of a 256-column row as a lane's bit field (bit k = column lane + 64k) */
DEV unsigned bs_range(int lo, int hi, int lane)
{
	unsigned b = 0;
	for (int k = 0; k < 4; k++) { const int col = lane + 64 * k; b |= (col >= lo && col <= hi ? 1u : 0u) << k; }
	return b;
}

/* Above quality 16 the walk's questions about a cell -- is it 8, 7, -7, a loud x6 / x7, one of the six marks of the passes before -- and the
 * value the walk leaves for it come out of one table word per value: -DQ_LIM .. DQ_LIM, then the marks 15300 .. 15800 (every fourth
 * value from 15300: the six marks and BIG between them).  Low half: dequant_value(floored cell), or what a visited mark stands for; then a
 * bit a question.  A larger value or an unknown code sends its 64 cells down the comparisons (DQ_BIG). */
#define DQ_LIM 512
#define DQ_MARK0 (2 * DQ_LIM + 1)
#define DQ_WORDS (DQ_MARK0 + 126)
enum : unsigned { DQ_E8 = 1u << 16, DQ_E7 = 1u << 17, DQ_EM7 = 1u << 18, DQ_DC = 1u << 19, DQ_AC = 1u << 20, DQ_BIG = 1u << 21, DQ_CODE = 1u << 22, DQ_K2 = 1u << 23, DQ_K1 = 1u << 24, DQ_PRP = 1u << 25, DQ_PRN = 1u << 26 };
DEV int dq_index(int x)
{
	const int d = x - 15300;
	const int cl = x < -DQ_LIM ? -DQ_LIM : x > DQ_LIM ? DQ_LIM : x;
	return ((unsigned)d <= 500u && !(d & 3)) ? DQ_MARK0 + (d >> 2) : cl + DQ_LIM;
}
DEV unsigned dq_entry(int i)
{
	if (i >= DQ_MARK0) {
		const int d = 4 * (i - DQ_MARK0);
		if (d % 100) return DQ_BIG;
		const int x = 15300 + d;                                   /* 15300 / 15500 -> 5, 15400 / 15600 -> -5, 15700 -> 6, 15800 -> -6 (:2909-3124) */
		const int v = (x == 15300 || x == 15500) ? 5 : (x == 15400 || x == 15600) ? -5 : x == 15700 ? 6 : -6;
		return (unsigned)(uint16_t)(int16_t)v | DQ_CODE | (x <= 15400 ? DQ_K2 : DQ_K1) | (x == 15700 ? DQ_PRP : 0u) | (x == 15800 ? DQ_PRN : 0u);
	}
	const int x = i - DQ_LIM;
	if (x >= DQ_LIM || x <= -DQ_LIM) return DQ_BIG;
	int a = x;
	if (a < 0) { a = -a; if ((a & 7) < 7) a &= 0xFFF8; a = -a; }
	unsigned e = (unsigned)(uint16_t)(int16_t)dequant_value(a);
	if (x == 8) e |= DQ_E8;
	if (x == 7) e |= DQ_E7;
	if (x == -7) e |= DQ_EM7;
	if (x > 12 && (x & 7) >= 6) e |= DQ_DC;
	if (x < -12 && ((-x) & 7) == 6) e |= DQ_AC;
	return e;
}
DEV void wave_dequant_details(Ctx *c, int part, int lane, const uint32_t *lut /* DQ_WORDS, filled by the workgroup */, bool keep_p, const int16_t *src, int ss /* see wave_ll2 */)
{
	int16_t *p = c->proc, *jp = c->jpeg;
	const bool hq = c->q > 16;                                     /* quality 1..16: no triple / pair marking, and negative magnitudes keep their low bits on a ration (:2938-2989) */
	int cur[4], nxt[4], pend_v[4] = { 0, 0, 0, 0 };
	unsigned pend = 0;                                             /* jp cells of the current row the row above has set (bit-sliced, like every row mask in here) */
	int q0[4], q1[4], q2[4];                                       /* rows r+2 .. r+4, already on their way (a row step is shorter than a memory round trip) */
	det_load_row(src, 0, lane, cur, ss);
	det_load_row(src, 1, lane, nxt, ss);
	det_load_row(src, 2, lane, q0, ss); det_load_row(src, 3, lane, q1, ss); det_load_row(src, 4, lane, q2, ss);
#define UP(b) bs_up<4>((b), lane)
#define DN(b) bs_dn((b), lane)
#define BIT(b, k) (((b) >> (k)) & 1u)
	/* one row; K0 = 2 in the upper half, where the first two words of a row (columns 0..127: they read as zero, det_load_row) take no part:
	 * their predicates and updates are not evaluated at all (the kernel is bound by vector work) */
	auto row_step = [&](auto k0c, const int r) {
		constexpr int K0 = decltype(k0c)::value;
		int far[4];
		det_load_row(src, r + 5, lane, far, ss);
		const bool top = r < H / 2;
		const int col0 = top ? H / 2 : 0;
		int jv[4] = { pend_v[0], pend_v[1], pend_v[2], pend_v[3] };
		unsigned je = pend;
		pend = 0;
		if (hq && r < H - 1) {                                     /* :2759-2853 (quality 17 and up) */
			unsigned P, N, PN, NN;
			BS_PREDK(P, cur, K0, x > 3 && x < 8); BS_PREDK(N, cur, K0, x < -3 && x > -8);
			BS_PREDK(PN, nxt, K0, x > 3 && x < 8); BS_PREDK(NN, nxt, K0, x < -3 && x > -8);
			const unsigned rg = bs_range(col0 + 1, H - 2, lane);
			const unsigned dP = DN(P), dN = DN(N);
			const unsigned pp = P & UP(P), nn = N & UP(N);
			const unsigned tp = pp & dP, tn = nn & dN;
			const unsigned vp = pp & ~dP & UP(PN) & PN, vn = nn & ~dN & UP(NN) & NN;
			const unsigned cand = (tp | vp | tn | vn) & rg;
			unsigned fired = 0;
			if (__any(cand != 0)) fired = bs_from4(alt_runs(bs_ballot4(cand)));
			if (__any(fired != 0)) {
				const unsigned ftp = fired & tp, ftn = fired & tn, fvp = fired & vp, fvn = fired & vn;
				const unsigned ft = ftp | ftn, fv = fvp | fvn;
				const unsigned ftp_l = DN(ftp), ftn_l = DN(ftn), fvp_l = DN(fvp), fvn_l = DN(fvn);   /* the cell left of a firing cell */
				const unsigned ftp_r = UP(ftp), ftn_r = UP(ftn);
				const unsigned touched = fired | ftp_l | ftn_l | fvp_l | fvn_l | ftp_r | ftn_r;   /* a firing cell, the cell on its left, the cell on the right of a triple: few cells of a row, and the nine assignments below are skipped for a word that holds none */
				for (int k = K0; k < 4; k++) {
					if (!__ballot(BIT(touched, k))) continue;
					if (BIT(ft, k)) cur[k] = 0;
					if (BIT(ftp_l, k)) cur[k] = 15300; if (BIT(ftn_l, k)) cur[k] = 15400;
					if (BIT(fvp_l, k)) { cur[k] = 15500; nxt[k] = 15500; }
					if (BIT(fvn_l, k)) { cur[k] = 15600; nxt[k] = 15600; }
					if (BIT(fv, k)) nxt[k] = 0;
					if (BIT(ftp, k) || BIT(fvp, k) || BIT(ftp_r, k)) jv[k] = 5;
					if (BIT(ftn, k)) jv[k] = -6;
					if (BIT(fvn, k) || BIT(ftn_r, k)) jv[k] = -5;
					pend_v[k] = BIT(fvp, k) ? 5 : -5;
				}
				je |= fired | ftp_r | ftn_r;
				pend = fv;
			}
		}
		if (!part && hq) {                                         /* :2857-2905 (quality 17 and up) */
			unsigned A, B;
			BS_PREDK(A, cur, K0, x >= 5 && x <= 7); BS_PREDK(B, cur, K0, x <= -5 && x >= -7);
			const unsigned cand = ((A & DN(A)) | (B & DN(B))) & bs_range(col0, H - 2, lane);
			if (__any(cand != 0)) {
				const unsigned fired = bs_from4(alt_runs(bs_ballot4(cand)));
				const unsigned fa = fired & A, fb = fired & B;
				for (int k = K0; k < 4; k++) { if (BIT(fa, k)) cur[k] = 15700; if (BIT(fb, k)) cur[k] = 15800; }
			}
		}
		{                                                          /* :2909-3124 */
			unsigned code = 0, k1 = 0, k2 = 0, pr_p = 0, pr_n = 0, wr = 0;
			unsigned e8 = 0, e7 = 0, em7 = 0, dc = 0, ac = 0;
			unsigned direct = hq ? 0u : 0xFu;                       /* words whose cells are sorted and dequantised by comparisons: all of them at quality 1..16 (no marks there, rationed low bits), above it those with a value beyond the table */
			int jq[4] = { 0, 0, 0, 0 };                             /* from the table: what the walk leaves for the cell */
			if (hq)
				for (int k = K0; k < 4; k++) {
					const unsigned e = lut[dq_index(cur[k])];
					if (__ballot(e & DQ_BIG)) { direct |= 1u << k; continue; }
#define DQB(dst, bit) (dst) |= (((e) & (bit)) ? 1u : 0u) << k
					DQB(code, DQ_CODE); DQB(k2, DQ_K2); DQB(k1, DQ_K1); DQB(pr_p, DQ_PRP); DQB(pr_n, DQ_PRN);
					DQB(e8, DQ_E8); DQB(e7, DQ_E7); DQB(em7, DQ_EM7); DQB(dc, DQ_DC); DQB(ac, DQ_AC);
#undef DQB
					jq[k] = (int)(int16_t)(e & 0xFFFFu);
				}
			if (direct)
				for (int k = K0; k < 4; k++) {
					if (!((direct >> k) & 1u)) continue;
					const int x = cur[k];
					e8 |= (x == 8 ? 1u : 0u) << k; e7 |= (x == 7 ? 1u : 0u) << k; em7 |= (x == -7 ? 1u : 0u) << k;
					dc |= (x > 12 && x < 15000 && (x & 7) >= 6 ? 1u : 0u) << k; ac |= (x < -12 && ((-x) & 7) == 6 ? 1u : 0u) << k;
					if (hq) {                                           /* (no marks below quality 17) */
						code |= (x > 15000 ? 1u : 0u) << k;
						k2 |= (x == 15300 || x == 15400 ? 1u : 0u) << k; k1 |= (x == 15500 || x == 15600 || x == 15700 || x == 15800 ? 1u : 0u) << k;
						pr_p |= (x == 15700 ? 1u : 0u) << k; pr_n |= (x == 15800 ? 1u : 0u) << k;
						wr |= (x > 15000 && !(x == 15300 || x == 15400 || x == 15500 || x == 15600 || x == 15700 || x == 15800) ? 1u : 0u) << k;   /* code cells with another value (none are produced) write nothing */
					}
				}
			const bool any_code = __any(code != 0);                 /* most rows of a calm picture carry no mark at all */
			const unsigned rd = bs_range(col0, H - 1, lane);
			unsigned skipped = 0;
			if (__any(((k1 | k2) & rd) != 0)) {                    /* which cells the walk steps over: a visited code cell hides the next one (two for a triple) */
				const M4 k12 = bs_ballot4((k1 | k2) & rd), k2m = bs_ballot4(k2);
				M4 sk4;
				uint64_t carry = 0;
				for (int k = 0; k < 4; k++) {
					uint64_t sk = carry, m = k12.w[k];
					carry = 0;
					while (m) {
						const int j = __builtin_ctzll(m);
						m &= m - 1;
						if (!((sk >> j) & 1)) {
							const uint64_t pat = ((k2m.w[k] >> j) & 1) ? 6 : 2;
							sk |= pat << j;
							if (j > 60) carry |= pat >> (64 - j);
						}
					}
					sk4.w[k] = sk;
				}
				skipped = bs_from4(sk4);
			}
			const unsigned vis = rd & ~skipped, vc = vis & code, vnc = vis & ~code;
			const unsigned ml = bs_range(0, H - 2, lane);
			const unsigned dm = part ? 0u : (vnc & ml & dc);
			const unsigned udm = UP(dm);
			const unsigned is8 = vnc & (e8 | (e7 & udm));           /* the walk sees an 8 here (a 7 the cell before has raised counts) */
			const unsigned to_m8 = em7 & UP((vnc & ml & ac) | (is8 & ml));
			const unsigned to_8 = e7 & udm;
			const unsigned self_m8 = vnc & ml & em7 & ~to_m8 & DN(e8);
			const unsigned m8 = to_m8 | self_m8;
			/* pr_p / pr_n: visited 15700 / 15800: the partner takes the same +-6 */
			const unsigned part_p = UP(pr_p & vc), part_n = UP(pr_n & vc);
			for (int k = K0; k < 4; k++) {
				if (BIT(m8, k)) cur[k] = -8;
				if (BIT(to_8, k)) cur[k] = 8;
			}
			/* quality 1..16: of the -15 the walk meets in a row every sixth is floored to -8 (the first, the seventh ..), of the -x7 below -22
			 * every fourth; all other negative values are floored (the reference's counters start at 0 in every row).  The count a cell
			 * finds is its rank among the row's cells of its kind: ballots and popcounts. */
			unsigned keep_low = 0;                                  /* visited negative cells that keep their low bits */
			if (!hq) {
				unsigned b15, bx7;
				BS_PREDK(b15, cur, K0, x == -15); BS_PREDK(bx7, cur, K0, x < -22 && ((-x) & 7) == 7);
				b15 &= vnc; bx7 &= vnc;
				if (__any((b15 | bx7) != 0)) {
					const M4 m15 = bs_ballot4(b15), mx7 = bs_ballot4(bx7);
					const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
					int base15 = 0, basex7 = 0;
					for (int k = 0; k < 4; k++) {
						const int r15 = base15 + __builtin_popcountll(m15.w[k] & below), rx7 = basex7 + __builtin_popcountll(mx7.w[k] & below);
						if (BIT(b15, k) && r15 % 6 != 0) keep_low |= 1u << k;
						if (BIT(bx7, k) && (rx7 & 3) != 0) keep_low |= 1u << k;
						base15 += __builtin_popcountll(m15.w[k]); basex7 += __builtin_popcountll(mx7.w[k]);
					}
				}
			}
			for (int k = K0; k < 4; k++) {
				if (BIT(part_p, k)) jv[k] = 6;
				if (BIT(part_n, k)) jv[k] = -6;
				if (!((direct >> k) & 1u)) {                            /* from the table; a -7 / 7 a neighbour has turned into -8 / 8 is the one thing it does not know */
					if (BIT(vis, k)) jv[k] = BIT(m8, k) ? dequant_value(-8) : BIT(to_8, k) ? dequant_value(8) : jq[k];
					continue;
				}
				if (BIT(vc, k)) {
					const int a = cur[k];
					if (a == 15300 || a == 15500) jv[k] = 5;
					else if (a == 15400 || a == 15600) jv[k] = -5;
					else if (a == 15700) jv[k] = 6;
					else if (a == 15800) jv[k] = -6;
				}
				if (BIT(vnc, k)) {
					int a = cur[k];
					if (a < 0) { a = -a; if (hq ? (a & 7) < 7 : !BIT(keep_low, k)) a &= 0xFFF8; a = -a; }
					jv[k] = dequant_value(a);
				}
			}
			je |= part_p | part_n | (vis & ~wr);
		}
		for (int k = K0; k < 4; k++) {
			const int at = r * W + lane + 64 * k;
			if (keep_p) p[at] = (int16_t)cur[k];
			if (BIT(je, k)) jp[at] = (int16_t)jv[k];
		}
		for (int k = 0; k < 4; k++) { cur[k] = nxt[k]; nxt[k] = q0[k]; q0[k] = q1[k]; q1[k] = q2[k]; q2[k] = far[k]; }
	};
	for (int r = 0; r < H / 2; r++) row_step(std::integral_constant<int, 2>{}, r);
	for (int r = H / 2; r < H; r++) row_step(std::integral_constant<int, 0>{}, r);
#undef UP
#undef DN
#undef BIT
	__threadfence_block();
}

/* isolated coefficient shrink (image_processing.c:3154-3188): a reconstructed detail >= 8 in magnitude with no
 * such neighbour moves one step towards zero.  Only isolated cells move, and a cell next to one is small, so the
 * decisions are those of the untouched plane; a three-row window of ">= 8" masks carries them. */
DEV void wave_shrink(Ctx *c, int lane)
{
	int16_t *jp = c->jpeg;
	const int diag = c->q <= 16 ? 16 : 8;                          /* quality 1..16 lets diagonal neighbours up to 15 pass (:3135-3188) */
	int jc[4], jn[4];
	unsigned bp, bc, bn;                                           /* ">= 8" of rows r-1, r, r+1, bit-sliced (bit k = column lane + 64k) */
	unsigned dp, dn_;                                              /* ">= diag" of rows r-1, r+1 */
	for (int k = 0; k < 4; k++) { jn[k] = jp[lane + 64 * k]; jc[k] = jp[W + lane + 64 * k]; }
	BS_PRED(bp, jn, 4, iabs(x) >= 8); BS_PRED(dp, jn, 4, iabs(x) >= diag);
	BS_PRED(bc, jc, 4, iabs(x) >= 8);
	unsigned dc_;                                                  /* ">= diag" of row r (becomes dp of the next step) */
	BS_PRED(dc_, jc, 4, iabs(x) >= diag);
	for (int k = 0; k < 4; k++) jn[k] = jp[2 * W + lane + 64 * k];
	BS_PRED(bn, jn, 4, iabs(x) >= 8); BS_PRED(dn_, jn, 4, iabs(x) >= diag);
	const unsigned inner = bs_range(1, H - 2, lane), right = bs_range(H / 2, H - 2, lane);
	int g0[4], g1[4];                                              /* rows r+2, r+3 in flight */
	for (int k = 0; k < 4; k++) { g0[k] = jp[3 * W + lane + 64 * k]; g1[k] = jp[4 * W + lane + 64 * k]; }
	for (int r = 1; r < H - 1; r++) {
		int jf[4] = { 0, 0, 0, 0 };
		if (r + 4 < H) for (int k = 0; k < 4; k++) jf[k] = jp[(r + 4) * W + lane + 64 * k];
		const unsigned side = bc | dp | dn_;                          /* a neighbour column: its cell of this row at 8 or more, or one of its two diagonal cells at `diag` or more */
		const unsigned near = bs_up<4>(side, lane) | bs_dn(side, lane) | bp | bn;
		const unsigned hit = bc & ~near & (r >= H / 2 ? inner : right);
		for (int k = 0; k < 4; k++)
			if ((hit >> k) & 1u) jp[r * W + lane + 64 * k] = (int16_t)(jc[k] > 0 ? jc[k] - 1 : jc[k] + 1);
		bp = bc; bc = bn; dp = dc_; dc_ = dn_;
		for (int k = 0; k < 4; k++) { jc[k] = jn[k]; jn[k] = g0[k]; g0[k] = g1[k]; g1[k] = jf[k]; }
		BS_PRED(bn, jn, 4, iabs(x) >= 8); BS_PRED(dn_, jn, 4, iabs(x) >= diag);
	}
}

/* ------------------------------------------------------------------------------------------------------------
 * Y28, the quantiser offsetY (image_processing.c:185-521, q>16 branches), on whole 512-column rows: 8 mask words,
 * lane l owns columns l + 64k.
 *   loop 1 (:195-238)  paired multiples of 8 in the level-1 detail bands: a cell may decrement itself or its right
 *                      neighbour, a decremented cell stops being a multiple of 8 and does nothing -> skip walk
 *   loop 2 (:241-284)  triples / vertical pairs of 4..7 in rows 0..255, columns 1..254 (writes the next row)
 *   loop 3 (:286-311)  equal-sign 5..7 pairs, rows 0..255
 *   loop 4 (:314-519)  the symbol of every cell; a cell may rewrite its right neighbour's +-7 first, which only
 *                      depends on the cell's own value: a stencil.  Its one unguarded look at the next cell at
 *                      column 511 sees the next row's first cell after loops 1-3, so a row is coded one step late.
 * ------------------------------------------------------------------------------------------------------------ */
struct M8 { uint64_t w[8]; };
DEV M8 operator&(M8 a, M8 b) { M8 r; for (int k = 0; k < 8; k++) r.w[k] = a.w[k] & b.w[k]; return r; }
DEV M8 operator|(M8 a, M8 b) { M8 r; for (int k = 0; k < 8; k++) r.w[k] = a.w[k] | b.w[k]; return r; }
DEV M8 operator~(M8 a) { M8 r; for (int k = 0; k < 8; k++) r.w[k] = ~a.w[k]; return r; }
DEV M8 up1(M8 a, unsigned in = 0) { M8 r; r.w[0] = (a.w[0] << 1) | in; for (int k = 1; k < 8; k++) r.w[k] = (a.w[k] << 1) | (a.w[k - 1] >> 63); return r; }
DEV M8 dn1(M8 a) { M8 r; for (int k = 0; k < 7; k++) r.w[k] = (a.w[k] >> 1) | (a.w[k + 1] << 63); r.w[7] = a.w[7] >> 1; return r; }
DEV M8 dn2(M8 a) { M8 r; for (int k = 0; k < 7; k++) r.w[k] = (a.w[k] >> 2) | (a.w[k + 1] << 62); r.w[7] = a.w[7] >> 2; return r; }
DEV M8 alt_runs(M8 f)
{
	const uint64_t even = 0x5555555555555555ull;
	const M8 pf = up1(f);
	M8 r;
	unsigned carry = 0;
	for (int k = 0; k < 8; k++) {
		const uint64_t se = f.w[k] & ~pf.w[k] & even;
		const uint64_t t = f.w[k] + se;
		const unsigned c1 = t < se;
		const uint64_t sum = t + carry;
		carry = c1 | (sum < t);
		const uint64_t re = f.w[k] & ~sum;
		r.w[k] = (re & even) | (f.w[k] & ~re & ~even);
	}
	return r;
}
#define BALLOT8(m, arr, expr) do { for (int k_ = 0; k_ < 8; k_++) { const int x = arr[k_]; (m).w[k_] = __ballot(expr); } } while (0)

DEV M8 bs_ballot8(unsigned b) { M8 m; for (int k = 0; k < 8; k++) m.w[k] = __ballot(b & (1u << k)); return m; }
DEV unsigned bs_from8(M8 m) { unsigned b = 0; for (int k = 0; k < 8; k++) b |= (unsigned)__builtin_amdgcn_inverse_ballot_w64(m.w[k]) << k; return b; }
/* multiples of 8 from `from` on (x = from, from + 8, ...) with one comparison */
DEV bool mult8_from(int x, int from) { return ((unsigned)(x - from) & 0x80000007u) == 0; }

/* the value one column to the left / right of every cell of a row held as v[k] = column lane + 64k (wave-wide DPP shift, the
 * word seam through a readlane); `edge` stands in where the row ends */
DEV int left_of_dpp(const int *v, int k, int lane, int edge)
{
	const int seam = k ? __builtin_amdgcn_readlane(v[k - 1], 63) : edge;
	(void)lane;
	return __builtin_amdgcn_update_dpp(seam, v[k], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);   /* lane 0 has no source and keeps the seam */
}
DEV int right_of_dpp(const int *v, int k, int nk, int lane, int edge)
{
	const int seam = k + 1 < nk ? __builtin_amdgcn_readlane(v[k + 1], 0) : edge;
	(void)lane;
	return __builtin_amdgcn_update_dpp(seam, v[k], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);   /* lane 63 has no source and keeps the seam */
}

/* ll: the level-2 block (rows and columns below 256) comes from its copy l2save instead of the work plane -- Y26 (nhw_encoder.c:1893-1910) put
 * it back there with the tags of its LL2 quarter cleared (everything up to 8000 in rows and columns below 128 becomes 0), for this reader
 * alone: 256 KB of copy per image that this load does itself */
DEV void quant_load_row(const int16_t *p, int r, int lane, int *v, const int16_t *ll = nullptr)
{
	if (ll && r < H) {
		for (int k = 0; k < 4; k++) { const int x = ll[r * H + lane + 64 * k]; v[k] = (r < H / 2 && k < 2 && x <= 8000) ? 0 : x; }
		for (int k = 4; k < 8; k++) v[k] = p[r * W + lane + 64 * k];
		return;
	}
	for (int k = 0; k < 8; k++) v[k] = r < W ? p[r * W + lane + 64 * k] : 0;       /* the cell behind the plane reads as 0 (zero guard) */
}


/* The symbols go straight into the stream in its serpentine order (Y30, nhw_encoder.c:2108-2132: 128 strips of 4 columns,
 * within a strip row after row, odd rows right to left): 16 rows are parked as bytes in a wave-private LDS block and
 * leave as one 64-byte run per strip.  The int16 plane is only written when Y29 needs it (q > 21). */
#define QROW 516
/* Loop 4 above quality 16 (image_processing.c:314-519, the q > 16 branches), per value of a cell clamped to -128 .. 128: the symbol it
 * takes when its neighbours do nothing (low byte), and what it is to its neighbours and they to it.  Three rules look sideways, and each
 * moves the symbol by one step of 8:
 *   a -7 behind a loud negative x6 (:375), behind an 8 (:389) or in front of one (:378) becomes -9 / -8: symbol 120 instead of 128;
 *   a 7 behind a loud positive x6 / x7 becomes 9 (:390): 136 instead of 128;
 *   a negative 15, 23, .. (kept whole by :396) in front of a 1 .. 7 loses 2 first (:381) and is floored after all: one step up.
 * A wavefront keeps the 257 words in LDS and a cell's step is a lookup, two lane shifts and a dozen bit operations (it was fifty
 * compare-and-selects for five of six words of a row: nearly every 64 cells hold a +-7). */
enum : unsigned { QE_AC = 1u << 8, QE_EQ8 = 1u << 9, QE_DC = 1u << 10, QE_R17 = 1u << 11, QE_M7 = 1u << 12, QE_P7 = 1u << 13, QE_N157 = 1u << 14, QE_BIG = 1u << 15, QE_LOUD = 1u << 16 };
#define QLUT 257
DEV int clamp128(int x) { return x < -128 ? -128 : x > 128 ? 128 : x; }   /* (one v_med3_i32) */
DEV unsigned quant_entry(int x)
{
	const bool neg = x < 0;
	const int m = neg ? -x : x;
	const int mf = (neg && (m & 7) < 7) ? (m & 504) : m;            /* :396: negative values are floored to a multiple of 8 unless they end in 7 */
	const int v = neg ? -mf : mf;
	unsigned e = (unsigned)(v + 7) < 15u ? 128u : (unsigned)((v + 128) & 248);
	if (x <= -13 && x >= -127 && (m & 7) == 6) e |= QE_AC;
	if (x == 8) e |= QE_EQ8;
	if (x >= 13 && x <= 127 && (x & 7) >= 6) e |= QE_DC;
	if (x >= 1 && x <= 7) e |= QE_R17;
	if (x == -7) e |= QE_M7;
	if (x == 7) e |= QE_P7;
	if (x <= -15 && x >= -127 && (m & 7) == 7) e |= QE_N157;
	if (m > 127) e |= QE_BIG;
	if (m >= 7) e |= QE_LOUD;
	return e;
}
/* The stream leaves as a LIST (round 5): a q20 image holds some 6 000 symbols that are not the zero symbol 128 among its 262 144, and the
 * dense 256 KB went out here, came back three times in Y31 and three times in the packetiser.  Per flush of 16 rows and strip -- a slice of
 * 64 consecutive stream symbols -- the lane that owns the strip leaves one 64-bit word (bit k: symbol k of the slice is not 128; nzq[flush][strip],
 * 1 KB a flush in two coalesced stores) and appends the symbols themselves to `vals`: flush after flush, inside a flush strip after
 * strip (a wave prefix sum of the popcounts), inside a slice in stream order.  fbase[f] is where flush f starts in `vals`.  Zero-run lengths
 * are gaps between set bits; Y31 (scan_rewrite_list_par) turns the map into stream order for the packetiser.  `dense`: the byte stream as
 * well (stage checks). */
DEV void wave_quantise_luma(Ctx *c, int lane, uint8_t *park /* 16 x QROW bytes of this wavefront */, uint32_t *lut /* QLUT words of this wavefront */, bool write_plane, bool dense, bool ll_from_save)
{
	const int16_t *const llsrc = ll_from_save ? c->l2save : nullptr;
	uint64_t *const nzq = c->nzq;
	uint8_t *const vals = c->vals;
	unsigned vtotal = 0;                                            /* values written so far (wave-uniform) */
	for (int i = lane; i < QLUT; i += 64) lut[i] = quant_entry(i - 128);
	__threadfence_block();
	int16_t *p = c->proc;
	uint8_t *stream = c->scan;
	int prev[8], cur[8], nxt[8];
	int q0[8];                                                     /* row r+2 in flight (and r+3: `far`; a row's step is 3.5 us, a memory round trip shorter) */
	quant_load_row(p, 0, lane, cur, llsrc);
	quant_load_row(p, 1, lane, nxt, llsrc);
	quant_load_row(p, 2, lane, q0, llsrc);
	for (int k = 0; k < 8; k++) prev[k] = 0;
	const bool low = c->q <= 16;                                   /* quality 1..16 (image_processing.c:357-410, :427-510): no loops 2 and 3; rationed low bits; the `quant4` pushes */
	int q4_turn = 0, q4_carry = 0;                                 /* quant4: its every-third-pair counter runs through the whole plane; a push out of column 511 lands in the next row's first cell */
	unsigned last_le0 = 0;                                         /* the last cell of the row above is <= 0 (loop 1 looks at it from column 0) */
	for (int r = 0; r <= W; r++) {                                 /* step r: loops 1-3 on row r, loop 4 on row r - 1 */
		int far[8];
		quant_load_row(p, r + 3, lane, far, llsrc);
		if (r < W) {
			if (r < H) {                                           /* loop 1, upper half: only columns 256..511 (words 4..7) take part */
				/* both rules start from two neighbours on multiples of 8 (from 8 up): a row without such a pair -- most rows -- is done after that test */
				const int *c4 = cur + 4;
				unsigned g8;
				BS_PRED(g8, c4, 4, mult8_from(x, 8));
				const unsigned both = g8 & bs_dn(g8, lane) & (lane == 63 ? 0x7u : 0xFu);                       /* columns <= 510 */
				last_le0 = __builtin_amdgcn_readlane(cur[7], 63) <= 0;
				if (__any(both != 0)) {
					unsigned g16, le0;
					BS_PRED(g16, c4, 4, mult8_from(x, 16)); BS_PRED(le0, c4, 4, x <= 0);
					const unsigned ple = bs_up<4>(le0, lane, __builtin_amdgcn_readlane(cur[3], 63) <= 0);
					const unsigned cself = both & g16 & ple;
					const unsigned cnext = both & ((g16 & ~ple) | (g8 & ~g16)) & bs_dn(g16, lane) & bs_dn(bs_dn(le0, lane), lane) & (lane >= 62 ? 0x7u : 0xFu);   /* columns <= 509 */
					const unsigned hit = __any(cnext != 0) ? bs_from4(up1(alt_runs(bs_ballot4(cnext)))) : 0u;      /* cells decremented by their left neighbour (a row with no such pair skips the run resolution) */
					const unsigned dec = hit | (cself & ~hit);
					for (int k = 0; k < 4; k++) cur[4 + k] -= (dec >> k) & 1;
				}
			} else {                                               /* loop 1, lower half: whole rows */
				unsigned g8;
				BS_PRED(g8, cur, 8, mult8_from(x, 8));
				const unsigned both = g8 & bs_dn(g8, lane) & (lane == 63 ? 0x7Fu : 0xFFu);
				const unsigned prev_le0 = last_le0;
				last_le0 = __builtin_amdgcn_readlane(cur[7], 63) <= 0;
				if (__any(both != 0)) {
					unsigned g16, le0;
					BS_PRED(g16, cur, 8, mult8_from(x, 16)); BS_PRED(le0, cur, 8, x <= 0);
					const unsigned ple = bs_up<8>(le0, lane, prev_le0);
					const unsigned cself = both & g16 & ple;
					const unsigned cnext = both & ((g16 & ~ple) | (g8 & ~g16)) & bs_dn(g16, lane) & bs_dn(bs_dn(le0, lane), lane) & (lane >= 62 ? 0x7Fu : 0xFFu);
					const unsigned hit = __any(cnext != 0) ? bs_from8(up1(alt_runs(bs_ballot8(cnext)))) : 0u;
					const unsigned dec = hit | (cself & ~hit);
					for (int k = 0; k < 8; k++) cur[k] -= (dec >> k) & 1;
				}
			}
			if (r < H && !low) {
				{                                                  /* loop 2 */
					unsigned P, N, PN, NN;
					BS_PRED(P, cur, 4, (unsigned)(x - 4) < 4u); BS_PRED(N, cur, 4, (unsigned)(x + 7) < 4u);
					BS_PRED(PN, nxt, 4, (unsigned)(x - 4) < 4u); BS_PRED(NN, nxt, 4, (unsigned)(x + 7) < 4u);
					const unsigned rg = (lane == 0 ? 0xEu : 0xFu) & (lane == 63 ? 0x7u : 0xFu);              /* columns 1..254 */
					const unsigned pdn = bs_dn(P, lane), ndn = bs_dn(N, lane);
					const unsigned pp = P & bs_up<4>(P, lane), nn = N & bs_up<4>(N, lane);
					const unsigned tp = pp & pdn, tn = nn & ndn;
					const unsigned vp = pp & ~pdn & bs_up<4>(PN, lane) & PN, vn = nn & ~ndn & bs_up<4>(NN, lane) & NN;
					const unsigned cand2 = (tp | vp | tn | vn) & rg;
					const unsigned fired = __any(cand2 != 0) ? bs_from4(alt_runs(bs_ballot4(cand2))) : 0u;
					const unsigned ftp = fired & tp, ftn = fired & tn, fvp = fired & vp, fvn = fired & vn, fv = fvp | fvn;
					const unsigned ft_l = bs_dn(ftp | ftn, lane), fvp_l = bs_dn(fvp, lane), fvn_l = bs_dn(fvn, lane), fv_l = fvp_l | fvn_l;
					const unsigned touched = fired | ft_l | fv_l;       /* a firing cell or the cell on its left: a word without one skips the assignments */
					if (__any(fired != 0))
					for (int k = 0; k < 4; k++) {
						if (!__ballot((touched >> k) & 1u)) continue;
						if ((ftp >> k) & 1) cur[k] = 12700; if ((ftn >> k) & 1) cur[k] = 12900;
						if ((ft_l >> k) & 1) cur[k] = 10100;
						if ((fv >> k) & 1) { cur[k] = 10100; nxt[k] = 10100; }
						if ((fvp_l >> k) & 1) cur[k] = 12100; if ((fvn_l >> k) & 1) cur[k] = 12200;
						if ((fv_l >> k) & 1) nxt[k] = 10100;
					}
				}
				{                                                  /* loop 3 */
					unsigned A, B;
					BS_PRED(A, cur, 4, (unsigned)(x - 5) < 3u); BS_PRED(B, cur, 4, (unsigned)(x + 7) < 3u);
					const unsigned cand3 = ((A & bs_dn(A, lane)) | (B & bs_dn(B, lane))) & (lane == 63 ? 0x7u : 0xFu);   /* columns 0..254 */
					const unsigned fired = __any(cand3 != 0) ? bs_from4(alt_runs(bs_ballot4(cand3))) : 0u;
					const unsigned fa = fired & A, fb = fired & B;
					for (int k = 0; k < 4; k++) { if ((fa >> k) & 1) cur[k] = 10300; if ((fb >> k) & 1) cur[k] = 10204; }
				}
			}
		}
		if (r >= 1 && low) {                                       /* loop 4 of quality 1..16 on row r - 1 */
			const int rr = r - 1;
			const int first_next = __builtin_amdgcn_readlane(cur[0], 0);   /* the row below, through loop 1, before the push this row may hand it */
			if (q4_carry) { if (lane == 0) prev[0] += q4_carry; q4_carry = 0; }   /* the push the row above handed down (:427-510 leaves it for the next row's first cell) */
			unsigned pusher = 0;                                        /* my cells that pushed to the left: their += 2 comes behind the "above 127" test of :314 */
			/* quant4 (:427-510): of the pairs of neighbours that both sit on x6 / x7 (>= 14) in a detail band every third one -- counted through
			 * the whole plane -- is pushed apart by 2.  A pushed cell stops being a candidate; candidates are few (two large coefficients of
			 * the right residues side by side): the walk over them runs on the scalar unit, in column order. */
			{
				unsigned cb = 0;
				for (int k = (rr < H ? 4 : 0); k < 8; k++) {               /* detail cells: rows from 256 on, or columns from 256 on */
					const int a = prev[k], nx = right_of_dpp(prev, k, 8, lane, first_next);
					cb |= (a >= 14 && a <= 127 && nx >= 14 && (a & 6) == 6 && (nx & 6) == 6 && ((a | nx) & 1) ? 1u : 0u) << k;   /* q4_cand; a value above 127 has left the walk with its code (:314) */
				}
				if (__any(cb != 0)) {
					const M8 cm = bs_ballot8(cb);
					int killed = -1;                                    /* the column a push has just changed */
					int dl[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
					for (int k = 0; k < 8; k++) {
						uint64_t m = cm.w[k];
						while (m) {
							const int bit = __builtin_ctzll(m);
							m &= m - 1;
							const int col = 64 * k + bit;
							if (col == killed) continue;
							if (!q4_turn) {
								const int a = __builtin_amdgcn_readlane(prev[k], bit);
								const int c1 = col + 1, c2 = col + 2;
								int nx = first_next, r2 = 0;
								if (c1 < W) { nx = 0; for (int kk = 0; kk < 8; kk++) if (kk == (c1 >> 6)) nx = __builtin_amdgcn_readlane(prev[kk], c1 & 63); }
								if (c2 < W) for (int kk = 0; kk < 8; kk++) if (kk == (c2 >> 6)) r2 = __builtin_amdgcn_readlane(prev[kk], c2 & 63);
								const bool left = (a & 504) == (nx & 504) ? a >= nx : a <= nx;
								const bool veto = col > 0 && col < W - 2 && ((r2 < -2 && r2 > -8) || (r2 < -7 && ((-r2) & 7) >= 6));
								const int push = left ? -2 : (veto ? 0 : 2);
								if (push) {
									for (int kk = 0; kk < 8; kk++) {
										if (push < 0 && kk == k && lane == bit) { dl[kk] += 2; pusher |= 1u << kk; }
										if (c1 < W && kk == (c1 >> 6) && lane == (c1 & 63)) dl[kk] += push;
									}
									if (c1 >= W) q4_carry = push;
									killed = c1;
								}
							}
							q4_turn = q4_turn == 2 ? 0 : q4_turn + 1;
						}
					}
					for (int k = 0; k < 8; k++) prev[k] += dl[k];
				}
			}
			/* the symbols: the same stencil on (left, cell, right) as above quality 16; what differs is which negative values keep their low
			 * bits: of the 15s a row's walk meets every sixth is floored to 8, of the x7 above 22 every fourth, all others are floored */
			int mv[8];
			unsigned negm = 0, b15 = 0, bx7 = 0;
			unsigned loud = 0;                                          /* words with a cell at +-7 or beyond: in the others every cell is inside the dead zone whatever its neighbours do (the fix-ups only touch +-7), as above quality 16 */
			for (int k = 0; k < 8; k++) loud |= (__ballot(iabs(prev[k]) >= 7) ? 1u : 0u) << k;
			for (int k = 0; k < 8; k++) {
				mv[k] = 0;
				if (!((loud >> k) & 1)) continue;
				const int raw = prev[k];
				const int lf = left_of_dpp(prev, k, lane, 0), rt = right_of_dpp(prev, k, 8, lane, first_next);
				const bool last = k == 7 && lane == 63;
				const bool m7 = raw == -7;
				const bool ac = (unsigned)(-lf - 13) < 115u && ((-lf) & 7) == 6;
				const bool to8 = lf == 8 || (rt == 8 && !last);
				const bool dc = (unsigned)(lf - 13) < 115u && (lf & 7) >= 6;
				int a = raw;
				a = (m7 && ac) ? -9 : a;
				a = (m7 && !ac && to8) ? -8 : a;
				a = (raw == 7 && dc) ? 9 : a;
				const bool neg = a < 0 && a >= -127;
				int m = a < 0 ? -a : a;
				m = (neg && m > 14 && (m & 7) == 7 && (unsigned)(rt - 1) < 7u) ? m - 2 : m;
				mv[k] = a < 0 ? -m : m;
				negm |= (neg ? 1u : 0u) << k;
				b15 |= (neg && m == 15 ? 1u : 0u) << k;
				bx7 |= (neg && m > 22 && (m & 7) == 7 ? 1u : 0u) << k;
			}
			unsigned keep_low = 0;
			if (__any((b15 | bx7) != 0)) {
				const M8 m15 = bs_ballot8(b15), mx7 = bs_ballot8(bx7);
				const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
				int base15 = 0, basex7 = 0;
				for (int k = 0; k < 8; k++) {
					const int r15 = base15 + __builtin_popcountll(m15.w[k] & below), rx7 = basex7 + __builtin_popcountll(mx7.w[k] & below);
					if (((b15 >> k) & 1) && r15 % 6 != 0) keep_low |= 1u << k;
					if (((bx7 >> k) & 1) && (rx7 & 3) != 0) keep_low |= 1u << k;
					base15 += __builtin_popcountll(m15.w[k]); basex7 += __builtin_popcountll(mx7.w[k]);
				}
			}
			for (int k = 0; k < 8; k++) {
				if (!((loud >> k) & 1)) {
					if (write_plane) p[rr * W + lane + 64 * k] = 128;
					park[(rr & 15) * QROW + lane + 64 * k] = 128;
					continue;
				}
				int v = mv[k];
				if (((negm >> k) & 1) && !((keep_low >> k) & 1)) v = -((-v) & 504);
				int sym = (unsigned)(v + 7) < 15u ? 128 : ((v + 128) & 248);
				const int raw = prev[k];
				if (raw > 127 && !((pusher >> k) & 1 && raw <= 129)) sym = big_code(raw, k_big_pos);
				else if (raw < -127) sym = big_code(-raw, k_big_neg);
				if (write_plane) p[rr * W + lane + 64 * k] = (int16_t)sym;
				park[(rr & 15) * QROW + lane + 64 * k] = (uint8_t)sym;
			}
		}
		if (r >= 1 && !low) {                                      /* loop 4 on row r - 1: a stencil on (left, cell, right), through the class table (quant_entry) */
			unsigned e[8];
			for (int k = 0; k < 8; k++) e[k] = lut[clamp128(prev[k]) + 128];
			const unsigned e_next = lut[clamp128(__builtin_amdgcn_readlane(cur[0], 0)) + 128];   /* the row below, already through loops 1-3 */
			for (int k = 0; k < 8; k++) {
				if (!__ballot(e[k] & QE_LOUD)) {                        /* 64 quiet cells: every one is inside the dead zone whatever its neighbours do (the fix-ups only touch +-7) */
					if (write_plane) p[(r - 1) * W + lane + 64 * k] = 128;
					park[((r - 1) & 15) * QROW + lane + 64 * k] = 128;
					continue;
				}
				const unsigned el = (unsigned)left_of_dpp(reinterpret_cast<const int *>(e), k, lane, 0), er = (unsigned)right_of_dpp(reinterpret_cast<const int *>(e), k, 8, lane, (int)e_next);
				const unsigned er8 = (k == 7 && lane == 63) ? 0u : er;   /* column 511: the fix-ups do not reach across the row end (:378), the look at the next cell (:381) does */
				const unsigned up = e[k] & (((er << 3) & QE_N157) | ((el << 3) & QE_P7));            /* a 15, 23, .. below zero in front of 1..7 is floored after all (:381); a 7 behind a loud x6 / x7 is raised to 9 (:390) */
				const unsigned dn = e[k] & QE_M7 & ((el << 4) | ((el | er8) << 3));                    /* a -7 behind a loud negative x6 (:375) or beside an 8 (:378, :389) leaves the dead zone */
				int sym = (int)(e[k] & 255u) + (up ? 8 : 0) - (dn ? 8 : 0);
				if (__ballot(e[k] & QE_BIG)) {                          /* marks of loops 2-3 and values beyond +-127: one word in six on busy pictures, and mostly for a large value, so the marks' seven comparisons wait behind a ballot of their own */
					const int raw = prev[k];
					bool mark = false;
					if (__ballot(raw > 10000)) {
						mark = raw > 10000 && (raw == 10100 || raw == 12700 || raw == 12900 || raw == 10204 || raw == 10300 || raw == 12100 || raw == 12200);
						if (mark) sym = raw == 10100 ? 128 : raw == 12700 ? 127 : raw == 12900 ? 129 : raw == 10204 ? 125 : raw == 10300 ? 126 : raw == 12100 ? 121 : 122;
					}
					if (!mark && (raw > 127 || raw < -127)) {                 /* big_code of either sign in one go */
						const int a = raw < 0 ? -raw : raw;
						int kk = ((a & 0xFFF8) - 128) >> 3;
						kk = kk > 18 ? 18 : kk;
						sym = raw > 0 ? 10 + 2 * kk + 2 * ((kk * 11) >> 5) : 60 + 2 * kk + 2 * (((kk + 1) * 11) >> 5);
					}
				}
				if (write_plane) p[(r - 1) * W + lane + 64 * k] = (int16_t)sym;
				park[((r - 1) & 15) * QROW + lane + 64 * k] = (uint8_t)sym;
			}
		}
		if (r >= 1) {
			if (((r - 1) & 15) == 15) {                                /* 16 rows complete: strips lane and lane + 64 */
				__threadfence_block();
				const int rb = r - 16, f = rb >> 4;
				if (lane == 0) c->fbase[f] = vtotal;
				for (int h = 0; h < 2; h++) {
					const int strip = lane + 64 * h;
					uint32_t w[16];
					for (int i = 0; i < 16; i++) {
						const uint32_t x = *reinterpret_cast<const uint32_t *>(park + i * QROW + 4 * strip);
						w[i] = ((rb + i) & 1) ? __builtin_bswap32(x) : x;
					}
					if (dense) {
						uint4 *dst = reinterpret_cast<uint4 *>(stream + strip * (4 * W) + 4 * rb);
						for (int i = 0; i < 4; i++) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
					}
					const uint64_t M = (uint64_t)ne_mask32(w, 0x80808080u) | (uint64_t)ne_mask32(w + 8, 0x80808080u) << 32;
					nzq[f * 128 + strip] = M;
					const unsigned cnt = (unsigned)__builtin_popcountll(M);
					unsigned incl = cnt;
					for (int o = 1; o < 64; o <<= 1) { const unsigned t_ = (unsigned)__shfl_up((int)incl, o); if (lane >= o) incl += t_; }
					unsigned at = vtotal + incl - cnt;
					vtotal += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
					for (uint64_t m = M; m; m &= m - 1) {                  /* the slice's symbols in stream order, back out of the parked rows (symbol 4 i + k is byte k of row i, odd rows mirrored) */
						const int bit = __builtin_ctzll(m), i = bit >> 2, k = bit & 3;
						vals[at++] = park[i * QROW + 4 * strip + (((rb + i) & 1) ? 3 - k : k)];
					}
				}
				__threadfence_block();
			}
		}
		for (int k = 0; k < 8; k++) { prev[k] = cur[k]; cur[k] = nxt[k]; nxt[k] = q0[k]; q0[k] = far[k]; }
	}
	if (lane == 0) c->fbase[32] = vtotal;
}

/* offsetY_recons256 (image_processing.c:2600-3190), one wavefront per image */
/* keep_p: the marked coefficients go back into the work plane as the reference's in-place pass leaves them.  Nothing reads them there -- the
 * synthesis that follows reads the dequantised plane, and the next analysis (first loop) or that synthesis (second loop) rewrites the whole
 * block -- so production leaves these 128 KB per image and loop out; the stage checks compare the plane and keep them. */
DEV void wave_dequant_sim_luma(Ctx *c, int part, int lane, const uint32_t *lut, bool from_save, bool keep_p)
{
	PROF_BEGIN();
	const int16_t *src = from_save ? c->l2save : c->proc;
	const int ss = from_save ? H : W;
	wave_ll2(c, part, lane, keep_p, src, ss);
	wave_dequant_details(c, part, lane, lut, keep_p, src, ss);
	if (!part) wave_shrink(c, lane);
	if (!lane) PROF(c, part ? 1 : 7);
}

}  /* namespace nhw */
#endif
