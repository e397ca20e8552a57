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
#include "nhw_tail_wave.h"

namespace nhw {

#define NT 256
#define BARRIER() __syncthreads()
#include "nhw_residual_rules.h"   /* Y21 / Y22 / Y23: value codes, kinds, actions, class tables, the steps as pure functions (also compiled for the host by tests/test_residual_rules.py) */

/* ---------------------------------------------------------------- P passes */

/* Y5 (nhw_encoder.c:144-177): reads proc, writes only its own ll1 cell */
DEV void tag_l2_details_par(Ctx *c, int tid)
{
	const int16_t *p = c->proc;
	for (int g = tid; g < Q / 8; g += NT) {                        /* 8 cells of one row per item, 16-byte loads */
		const int r = g >> 5, j0 = (g & 31) * 8;
		if (r < H / 2 && j0 < H / 2) continue;
		const int at0 = r * W + j0;
		const uint4 pv = *reinterpret_cast<const uint4 *>(p + at0);
		uint4 cv = *reinterpret_cast<const uint4 *>(c->ll1 + r * H + j0);
		/* the two diagonal neighbours a 2..4 sample looks at (linear indexing, as the reference's): cells at0 - W - 1 .. of the row above and
		 * at0 + W + 1 .. of the row below, requested with the item's own loads instead of one by one when a sample asks (a dependent round
		 * trip per asking sample was most of this pass's time) */
		uint4 uv = make_uint4(0, 0, 0, 0), dv = uv;
		int u0 = 0, d7 = 0;
		if (r > 0) { uv = *reinterpret_cast<const uint4 *>(p + at0 - W); u0 = p[at0 - W - 1]; }
		dv = *reinterpret_cast<const uint4 *>(p + at0 + W); d7 = p[at0 + W + 8];
		const uint32_t uw[4] = { uv.x, uv.y, uv.z, uv.w }, dw[4] = { dv.x, dv.y, dv.z, dv.w };
		uint32_t pw[4] = { pv.x, pv.y, pv.z, pv.w }, cw[4] = { cv.x, cv.y, cv.z, cv.w }, aw[4] = { 0, 0, 0, 0 };
#pragma unroll
		for (int e = 0; e < 8; e++) {
			const int s = (int16_t)(pw[e >> 1] >> (16 * (e & 1))), at = at0 + e;
			const int up = e ? (int)(int16_t)(uw[(e - 1) >> 1] >> (16 * ((e - 1) & 1))) : u0;
			const int dn = e < 7 ? (int)(int16_t)(dw[(e + 1) >> 1] >> (16 * ((e + 1) & 1))) : d7;
			/* (no branches: eight cells a thread took every arm of the chain in every item, and the pass was bound by the scalar unit's mask
			 * bookkeeping -- 2 scalar instructions for every vector one) */
			const bool neg_big = s < -7 && mult8_or_7(-s), neg_mid = s < -4 && s >= -7;
			const bool p24 = s >= 2 && s < 5, p24_hit = p24 && at >= W + 1 && at < 2 * Q - W - 1 && (up != 0 || dn != 0);
			const bool pos = s >= 0 && !p24, p01 = pos && (s & 7) < 2, p57 = pos && !p01 && s > 4 && s <= 7;
			const int add = (neg_big || p57) ? 16000 : (neg_mid || p24_hit || p01) ? 12000 : 0;
			aw[e >> 1] |= (uint32_t)add << (16 * (e & 1));
		}
		typedef unsigned short u16x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
		for (int i = 0; i < 4; i++) cw[i] = __builtin_bit_cast(uint32_t, (u16x2_)(__builtin_bit_cast(u16x2_, cw[i]) + __builtin_bit_cast(u16x2_, aw[i])));   /* each cell in its own half (v_pk_add_u16) */
		*reinterpret_cast<uint4 *>(c->ll1 + r * H + j0) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
	}
}

/* Y8 (:183-216): every tagged L2 coefficient (r, j) of the transposed coefficient plane nudges the recon sample
 * that sits under it in the natural-orientation plane: (2(j-128)+1, 2r), (2j, 2(r-128)+1) or (2(j-128)+1,
 * 2(r-128)+1).  That is a transposition; done naively it is 49k scattered 2-byte read-modify-writes per image
 * (measured: 133 GB fetched per batch).  Here it goes through LDS in 64x64 target tiles: the three 32x32 blocks
 * of tags that land in a tile are read (and cleaned) row-wise, their +-1 steps are parked in LDS, and the
 * target rows are updated with coalesced dword read-modify-writes. */
DEV void apply_tags_par(Ctx *c, int tid, int16_t *lds)
{
	int8_t *steps = reinterpret_cast<int8_t *>(lds);              /* [3][32][33] */
	int16_t *p = c->proc, *o = c->ll1;
	/* A tile is two short phases with a memory round trip in front of each (the tags; the target dwords): both are requested a tile ahead.
	 * Tiles touch disjoint tags and disjoint target cells, so reading the next tile's before this one's are written back changes nothing. */
	auto tag_cell = [&](int tile, int k) {
		const int ty = tile >> 2, tx = tile & 3, type = k >> 10, rl = (k >> 5) & 31, jl = k & 31;
		const int r = 32 * tx + rl + (type >= 1 ? H / 2 : 0), j = 32 * ty + jl + (type != 1 ? H / 2 : 0);   /* 0: (r<128, j>=128), 1: (r>=128, j<128), 2: both >= 128 */
		return o + r * H + j;
	};
	auto target = [&](int tile, int k) {                            /* one dword = target cells (yy, xx), (yy, xx+1), xx even */
		const int ty = tile >> 2, tx = tile & 3, yl = k >> 5, xl = 2 * (k & 31);
		return reinterpret_cast<uint32_t *>(p + (64 * ty + yl) * W + 64 * tx + xl);
	};
	int16_t tg[12]; uint32_t tv[8];
#pragma unroll
	for (int u = 0; u < 12; u++) tg[u] = *tag_cell(0, tid + u * NT);
#pragma unroll
	for (int u = 0; u < 8; u++) tv[u] = *target(0, tid + u * NT);
	for (int tile = 0; tile < 16; tile++) {
#pragma unroll
		for (int u = 0; u < 12; u++) {
			const int k = tid + u * NT, type = k >> 10, rl = (k >> 5) & 31, jl = k & 31;
			int step = 0;
			if (tg[u] > 14000) { *tag_cell(tile, k) = (int16_t)(tg[u] - 16000); step = 1; }
			else if (tg[u] > 10000) { *tag_cell(tile, k) = (int16_t)(tg[u] - 12000); step = -1; }
			steps[(type * 32 + rl) * 33 + jl] = (int8_t)step;
		}
		uint32_t cur[8];
#pragma unroll
		for (int u = 0; u < 8; u++) cur[u] = tv[u];
		if (tile + 1 < 16) {
#pragma unroll
			for (int u = 0; u < 12; u++) tg[u] = *tag_cell(tile + 1, tid + u * NT);
#pragma unroll
			for (int u = 0; u < 8; u++) tv[u] = *target(tile + 1, tid + u * NT);
		}
		BARRIER();
		const int ty = tile >> 2;
#pragma unroll
		for (int u = 0; u < 8; u++) {
			const int k = tid + u * NT, yl = k >> 5, xl = 2 * (k & 31), yy = 64 * ty + yl;
			const int rl = xl >> 1, jl = yl >> 1;
			int s0 = 0, s1 = 0;
			if (yy & 1) { s0 = steps[(0 * 32 + rl) * 33 + jl]; s1 = steps[(2 * 32 + rl) * 33 + jl]; }   /* odd row: even col <- type 0, odd col <- type 2 */
			else s1 = steps[(1 * 32 + rl) * 33 + jl];                                                      /* even row: odd col <- type 1 */
			if (s0 | s1) {
				const uint32_t v = cur[u];
				*target(tile, k) = (uint32_t)(uint16_t)((int16_t)(v & 0xFFFF) + s0) | ((uint32_t)(uint16_t)((int16_t)(v >> 16) + s1) << 16);
			}
		}
		BARRIER();
	}
}

/* rows x cols block copy between two planes */
/* (16-byte pieces, four in flight per thread: with one 8-byte piece a turn -- load, then a store that may alias the next load -- the copy of a
 * 128 KB block was 64 memory round trips one after the other; every block copied here is a multiple of 8 columns wide on 16-byte rows) */
DEV void copy_block_par(const int16_t *__restrict__ src, int src_row, int16_t *__restrict__ dst, int dst_row, int rows, int cols, int tid)
{
	const int per = cols >> 3, n = rows * per;       /* 16-byte pieces */
	for (int i0 = tid; i0 < n; i0 += 4 * NT) {
		uint4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) { const int idx = i0 + u * NT; if (idx < n) v[u] = reinterpret_cast<const uint4 *>(src + (idx / per) * src_row)[idx % per]; }
#pragma unroll
		for (int u = 0; u < 4; u++) { const int idx = i0 + u * NT; if (idx < n) reinterpret_cast<uint4 *>(dst + (idx / per) * dst_row)[idx % per] = v[u]; }
	}
}

/* ---------------------------------------------------------------- LDS column tiles for the row-serial passes */
/* A thread that walks "its" row of a plane touches one 2-byte cell of a different cache line than its
 * neighbours at every step: measured 10-30x HBM read amplification (profiles/round1_pmc_front.json).  The
 * row-serial passes therefore run on tiles: 64 columns (+2 halo columns on each side) of all the rows of the
 * pass are staged in LDS with row-contiguous (coalesced) dword loads, every thread walks its row inside LDS,
 * and the tile goes back with coalesced stores.  Tile rows are TLS shorts apart.  Halo columns are addressed linearly, so "one cell past the end of the row" is the
 * first cell of the next row, exactly as in the reference's linear indexing. */
#define TLC 32      /* columns per tile: 258 rows x 36 shorts = 18.6 KB keeps 6+ workgroups on a CU (a 64-column tile halves that and is slower overall) */
#define TLS 36
DEV void tile_load(int16_t *lds, const int16_t *plane, int rs, int nrows, int c0, int tid)
{
	for (int idx = tid; idx < nrows * (TLS / 2); idx += NT) {
		const int r = idx / (TLS / 2), d = idx % (TLS / 2);
		reinterpret_cast<uint32_t *>(lds + r * TLS)[d] = reinterpret_cast<const uint32_t *>(plane + (size_t)r * rs + c0 - 2)[d];
	}
}
DEV void tile_store(const int16_t *lds, int16_t *plane, int rs, int nrows, int c0, int ncols /* even, <= TLC + 2 */, int tid, int first = 2 /* 2: from column c0, 0: from c0-2 */)
{
	const int d0 = first >> 1, nd = (ncols + (2 - first)) >> 1;
	for (int idx = tid; idx < nrows * nd; idx += NT) {
		const int r = idx / nd, d = idx % nd + d0;
		reinterpret_cast<uint32_t *>(plane + (size_t)r * rs + c0 - 2)[d] = reinterpret_cast<const uint32_t *>(lds + r * TLS)[d];
	}
}

/* run(row, r, j, j1, state): process columns [j, j1) of processing row r (row[jj] is the cell of absolute column jj,
 * valid for c0-2 <= jj < c0+66; row[jj +- TLS] are the plane rows below / above), return the next column to visit
 * (>= j1; skips may overshoot into the next tile).  The tile holds `nrows` plane rows starting at `plane`; thread t
 * (t < nproc) owns tile row t + roff.  Cells the pass may write: own row, columns c0-2 .. c0+65. */
template <class F>
DEV void row_pass_tiled(int16_t *plane, int rs, int row_end, int nrows, int roff, int nproc, int jb, int je, int16_t *lds, int tid, F f, typename F::State *out = nullptr)
{
	int jnext = jb;
	typename F::State st = f.init(tid);
	for (int c0 = (jb / TLC) * TLC; c0 < je; c0 += TLC) {
		tile_load(lds, plane, rs, nrows, c0, tid);
		BARRIER();
		if (tid < nproc) {
			const int j1 = c0 + TLC < je ? c0 + TLC : je;
			if (jnext < j1) jnext = f.run(lds + (tid + roff) * TLS + 2 - c0, tid, jnext, j1, st);
		}
		BARRIER();
		int nst = TLC + 2;
		if (row_end - c0 < nst) nst = row_end - c0;
		tile_store(lds + roff * TLS, plane + (size_t)roff * rs, rs, nproc, c0, nst, tid, c0 > 0 ? 0 : 2);
		BARRIER();
	}
	if (out) *out = st;
}

/* ---------------------------------------------------------------- Y9 (R) */
/* (:218-279) left neighbour is read after its own update, right neighbour before: serial along a row; rows do
 * not interact (column 0 reads proc[r][-1] = the LH1 cell (r-1, 511), which this pass never writes). */
DEV void unpack4(uint2 w, int v[4]) { v[0] = (int16_t)(w.x & 0xFFFF); v[1] = (int16_t)(w.x >> 16); v[2] = (int16_t)(w.y & 0xFFFF); v[3] = (int16_t)(w.y >> 16); }
/* the step of a cell with difference d, given the difference on its right as the walk found it and the one on its left as the walk left it
 * (:225-279): what it sees of the two is their sum a, the right one first pulled in by its own large step */
DEV int precomp_right(int dnext) { return iabs(dnext) > 4 ? dnext + big_step(dnext) : dnext; }
DEV int precomp_pick(int d, int a)
{
	int step = big_step(d);
	if (!step && iabs(d) > 1) {
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
	return step;
}
DEV int precomp_step(int d, int dnext, int prev) { return precomp_pick(d, precomp_right(dnext) + prev); }
/* precomp_pick only tells d apart inside -12 .. 12 and a inside -5 .. 5: as a table (k_l2_recon) */
#define PRECOMP_TAB (25 * 11)
DEV int precomp_index(int d, int a) { return __mul24((d < -12 ? -12 : d > 12 ? 12 : d) + 12, 11) + (a < -5 ? -5 : a > 5 ? 5 : a) + 5; }
DEV void precompensate_ll1_par(Ctx *c, int tid, int16_t *lds)
{
	/* The walk only looks at differences d = recon - ll1 (its own, its right neighbour's original one, its left neighbour's
	 * updated one) and moves recon and the jpeg copy of ll1 by the same step.  A wavefront takes a row, a lane four cells: what travels
	 * along the row is the updated difference of the cell on the left; the lanes start from the original one and hand theirs on until
	 * nothing moves.  The next row is on its way meanwhile. */
	int16_t *p = c->proc, *o = c->ll1, *jp = c->jpeg;
	const int lane = tid & 63, wv = tid >> 6, c0 = 4 * lane;
	(void)lds;
	uint2 pc = make_uint2(0, 0), oc = pc; int edge = 0;
#define PRE_LOAD(r) do { pc = *reinterpret_cast<const uint2 *>(p + (size_t)(r) * W + c0); oc = *reinterpret_cast<const uint2 *>(o + (size_t)(r) * H + c0); \
		if (lane == 0) edge = p[(size_t)(r) * W - 1] - o[(size_t)(r) * H - 1];       /* left neighbour of column 0: the cells before the row in memory, never updated */ \
		if (lane == 63) edge = p[(size_t)(r) * W + H] - o[(size_t)(r) * H + H]; } while (0)
	int r = wv;
	PRE_LOAD(r);
	for (; r < H; r += NT / 64) {
		int pv[4], ov[4], d[4], st[4];
		unpack4(pc, pv); unpack4(oc, ov);
		const int my_edge = edge, row = r;
#pragma unroll
		for (int k = 0; k < 4; k++) d[k] = (int16_t)(pv[k] - ov[k]);
		if (r + NT / 64 < H) PRE_LOAD(r + NT / 64);
		const int sd = __shfl_down(d[0], 1), su = __shfl_up(d[3], 1);
		const int dn4 = lane < 63 ? sd : my_edge;                  /* the difference on the right of my last cell, as it was */
		const int first = lane ? su : my_edge;
		int prev_in = first, prev_out;
		for (;;) {
			int prev = prev_in;
#pragma unroll
			for (int k = 0; k < 4; k++) { st[k] = precomp_step(d[k], k < 3 ? d[k + 1] : dn4, prev); prev = d[k] + st[k]; }
			prev_out = prev;
			int np = __shfl_up(prev_out, 1);
			if (!lane) np = first;
			if (!__any(np != prev_in)) break;
			prev_in = np;
		}
		uint2 w;
		w.x = (uint32_t)(uint16_t)(pv[0] + st[0]) | ((uint32_t)(uint16_t)(pv[1] + st[1]) << 16); w.y = (uint32_t)(uint16_t)(pv[2] + st[2]) | ((uint32_t)(uint16_t)(pv[3] + st[3]) << 16);
		*reinterpret_cast<uint2 *>(p + (size_t)row * W + c0) = w;
		w.x = (uint32_t)(uint16_t)(ov[0] + st[0]) | ((uint32_t)(uint16_t)(ov[1] + st[1]) << 16); w.y = (uint32_t)(uint16_t)(ov[2] + st[2]) | ((uint32_t)(uint16_t)(ov[3] + st[3]) << 16);
		*reinterpret_cast<uint2 *>(jp + (size_t)row * W + c0) = w;
	}
#undef PRE_LOAD
	BARRIER();
}

/* ---------------------------------------------------------------- a8 dequantisation simulation */
/* (the dequantiser simulation itself is a wavefront-per-image kernel for every quality: nhw_tail_wave.h) */
/* ---------------------------------------------------------------- Y21 (R) */
/* (:970-1073) only same-row neighbours are read or written (the vertical branches are unreachable) */
/* A cell may overwrite its two neighbours: the one on the left has been visited (nobody looks at it again: that is just its final value),
 * the one on the right has not, so what travels along the row is (value of the cell as its visit left it, value forced on the next cell).
 * A wavefront takes a row, a lane four cells; the lanes start from "nothing forced, left neighbour as it was" and hand their results to
 * the right until nothing moves.  (Every value a cell can be set to fails all of the tests, so the chain dies out after a cell or two.) */
/* The cell rule (:970-1073) only asks where a value lies: 4 .. 7 (bit 0: a neighbour of a triple), 5 .. 7 (bit 1: its centre), 6, 7 (bit 2:
 * what turns an 8 into a 10), 8 (bit 3), with its sign (bit 4) -- one code a value, from a table in a constant:
 *   x in +-5..7 between two neighbours in +-4..7 of its sign: x = 12700 / 12900, the next cell is forced to 10100, a triple;
 *   x = +-8 next to a +-6, 7 of its sign: 10 / -9; else (LH1 only) with a +-8 of its sign on its right: 9 / -9, and the next cell forced to the same.
 * Written without branches: 64 lanes x 4 cells found every branch of the chain in every row. */
/* (tag_code, tag_rule, tag_fires4: nhw_residual_rules.h) */
template <int PASS>
DEV void tag_rows_wave(int16_t *p, int r_first, int r_last, int c_base, int jb, int je, int tid)
{
	const int lane = tid & 63, wv = tid >> 6, c0 = c_base + 4 * lane;
	/* four rows of the wavefront's in flight while it works on the four before: a row is 512 bytes, and with one row ahead a CU's 32 wavefronts
	 * had 16 KB on their way -- the pass waited for memory latency (2.5 us a row) */
	constexpr int TD = 4, RS = NT / 64;
	uint2 nxt[TD];
#pragma unroll
	for (int u = 0; u < TD; u++) { const int rr = r_first + wv + u * RS; nxt[u] = rr <= r_last ? *reinterpret_cast<const uint2 *>(p + (size_t)rr * W + c0) : make_uint2(0, 0); }
	bool act[4];
#pragma unroll
	for (int k = 0; k < 4; k++) act[k] = c0 + k >= jb && c0 + k < je;
	const uint32_t act_bytes = (act[0] ? 1u : 0u) | (act[1] ? 1u << 8 : 0u) | (act[2] ? 1u << 16 : 0u) | (act[3] ? 1u << 24 : 0u);
	for (int rb = r_first + wv; rb <= r_last; rb += TD * RS) {
	uint2 curs[TD];
#pragma unroll
	for (int u = 0; u < TD; u++) curs[u] = nxt[u];
#pragma unroll
	for (int u = 0; u < TD; u++) { const int rr = rb + (TD + u) * RS; if (rr <= r_last) nxt[u] = *reinterpret_cast<const uint2 *>(p + (size_t)rr * W + c0); }
#pragma unroll
	for (int u = 0; u < TD; u++) {
		const int r = rb + u * RS;
		if (r > r_last) continue;
		int o[4], e[4];
		unpack4(curs[u], o);
		const int row = r;
#pragma unroll
		for (int k = 0; k < 4; k++) e[k] = tag_code(o[k]);
		const int e_right = __shfl_down(e[0], 1), e_left = __shfl_up(e[3], 1);   /* (lane 63's right and lane 0's left neighbour only meet cells outside the pass) */
		/* Does any cell of the row fire with its neighbours as they are?  All four cells of a lane at once, a code a byte: if none does, the
		 * walk changes nothing (a cell only sees another left neighbour behind a cell that fired) -- nearly every row: the walk itself
		 * was most of the pass's instructions. */
		if (!__any(tag_fires4(PASS, e, e_left, e_right, act_bytes) != 0)) continue;
		const int sd0 = __shfl_down(o[0], 1);                      /* the cell on the right of my last one, as it was */
		const int left0 = __shfl_up(o[3], 1);
		int lv_in = left0, force_in = 0, own[4], lv_out, force_out;
		bool trip[4];
		for (;;) {
			int el = tag_code(lv_in), force = force_in;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const bool forced = force != 0;
				const int x = forced ? force : o[k], ex = forced ? 0 : e[k], er = k < 3 ? e[k < 3 ? k + 1 : k] : e_right;
				int ow; bool t3;
				tag_rule(PASS, act[k], x, ex, el, er, ow, force, t3);
				trip[k] = t3; own[k] = ow;
				el = ow != x ? 0 : ex;                              /* (every value a cell can be set or forced to has code 0) */
			}
			lv_out = own[3]; force_out = force;
			int nl = __shfl_up(lv_out, 1), nf = __shfl_up(force_out, 1);
			if (!lane) { nl = left0; nf = 0; }
			/* what arrives matters to a lane only through its first cell: a forced value, or another left neighbour for a cell that looks at it
			 * (+-5..7 or +-8, not forced, inside the pass) -- anything else would repeat the evaluation to the same result (half the rows did) */
			if (!__any(nf != force_in || (nl != lv_in && act[0] && !force_in && (e[0] & 10)))) break;
			lv_in = nl; force_in = nf;
		}
		const int tnext = __shfl_down((int)trip[0], 1);            /* a triple marks the cell before it as well */
		int fin[4];
#pragma unroll
		for (int k = 0; k < 4; k++) fin[k] = (k < 3 ? trip[k < 3 ? k + 1 : k] : (lane < 63 && tnext)) ? 10100 : own[k];
		uint2 w;
		w.x = (uint32_t)(uint16_t)fin[0] | ((uint32_t)(uint16_t)fin[1] << 16); w.y = (uint32_t)(uint16_t)fin[2] | ((uint32_t)(uint16_t)fin[3] << 16);
		if (fin[0] != o[0] || fin[1] != o[1] || fin[2] != o[2] || fin[3] != o[3]) *reinterpret_cast<uint2 *>(p + (size_t)row * W + c0) = w;   /* (most pieces stay as they are: 1 GB a batch went back unchanged) */
	}
	}
}
DEV void tag_small_runs_par(Ctx *c, int tid, int16_t *lds)
{
	(void)lds;
	tag_rows_wave<0>(c->proc, 1, H - 2, H, H + 1, W - 1, tid);                   /* rows 1..254, LH1 columns 257..510 */
	tag_rows_wave<1>(c->proc, H + 1, W - 2, 0, 1, H - 1, tid);                   /* rows 257..510, HL1 columns 1..254 */
	BARRIER();
}

/* Y22 (:1077-1325).  The reference walks column after column; column j only reads column j+1 (not yet visited, i.e. its original values)
 * and the recon sample (j, 255) of the last column.  Columns 0..254 therefore run in parallel, one thread each; a column's walk is a chain
 * (a step rewrites the two samples below it and reads what the step before left).  The planes come as rows through LDS tiles, a chunk of
 * CR rows at a time; the LH1 coefficients -- row j of the plane for column j -- through a transposing piece tile.  Until round 5 Y22 and
 * Y23 were two sweeps (column 255 in between, on a packed copy); now one: residuals_fused_par. */
#ifndef CR
#define CR 4      /* rows per chunk (measured on the two-sweep form, ms per 4096-image batch: CR 2: 3.78, 4: 3.40, 8: 3.71, 16: 4.00) */
#endif
/* The LH1 coefficients of column j are row j of the plane, LW of them at a time for all 256 rows: whole 64-byte pieces of every row
 * (a piece per chunk of rows is a few bytes of a line that has left the L2 again when the next chunk asks for its neighbour -- measured
 * 4x the band's bytes in either direction with 16-byte pieces).  LP: LDS pitch of a column's piece, an odd number of dwords. */
#ifndef LW
#define LW 16     /* shorts of an LH1 line a tile holds (32 bytes; measured again in round 5 with 64 and 128 bytes: DESIGN 4.3) */
#endif
#define LP (LW + 2)
DEV void lh_tile_load(int16_t *lt, const int16_t *p, int r0, int tid)
{
	for (int v = tid; v < H * (LW / 8); v += NT) {
		const int jj = v / (LW / 8), h = v % (LW / 8);
		const uint4 x = *reinterpret_cast<const uint4 *>(p + jj * W + H + r0 + 8 * h);
		uint32_t *d = reinterpret_cast<uint32_t *>(lt + jj * LP + 8 * h);
		d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
	}
}
DEV void lh_tile_store(const int16_t *lt, int16_t *p, int r0, int tid)
{
	for (int v = tid; v < H * (LW / 8); v += NT) {
		const int jj = v / (LW / 8), h = v % (LW / 8);
		const uint32_t *d = reinterpret_cast<const uint32_t *>(lt + jj * LP + 8 * h);
		*reinterpret_cast<uint4 *>(p + jj * W + H + r0 + 8 * h) = make_uint4(d[0], d[1], d[2], d[3]);
	}
}
/* ---------------------------------------------------------------- Y22 + Y23 as ONE sweep (round 5)
 * Y23's step at (r, j) reads what Y22's step at (r, j) has just left -- the recon sample of row r (final since step r - 1), the LL1 cell of row
 * r (the step's code, or the cell as the step before left it) and the LH1 coefficient (j, 256 + r) -- and nothing that a LATER Y22 step of any
 * column reads: Y22 looks at its own column's rows r .. r + 2 (in registers here: ColState), at the row above only through the sign of its
 * difference (ColState::dm1, as Y22 left it), at the coefficient before (lhm1: as Y22 left it, a register; Y23 keeps its own, vm1) and at the
 * ORIGINAL differences of the column on its right (the tile dt, never written).  So a column thread does both steps of a row one after the
 * other and the planes are read once: the recon plane's LL quadrant and the LL1 plane as rows (coalesced, a chunk of CR rows ahead in registers),
 * the LH1 band through the transposing piece tile; the LL1 plane (now the code plane) and the band go back once.  The recon plane's LL
 * quadrant is NOT written back: nothing reads it behind Y23 (Y26 or the quantiser take the level-2 block from l2save) -- except row 256, the
 * first row of HL1, which Y22's last step reaches.
 *
 * What the reference's column order adds: column 255 is visited last by Y22, and (a) every column's Y22 starts from the recon sample
 * (j, 255) as it was BEFORE that visit, every column's Y23 from the sample AFTER it; (b) column 255's "column on the right" is the LH1
 * coefficient (x, 256) as the Y22 step 0 of column x left it minus the LL1 cell (x + 1, 0) as column 0's whole Y22 walk left it.  Hence a
 * prologue on packed LDS copies: step 0 of every column (only the coefficient is kept), then one thread walks column 0 and column 255
 * (2 x 255 steps); column 255's results stay in LDS (oc, rc: its cells and residuals as Y22 left them) for thread 255, which only does Y23. */
struct ColState { int v0, o0, v1, o1, dm1; };                     /* recon sample and LL1 cell of rows r, r + 1 as the walk left them; (sample - cell) of row r - 1 */
template <class NBF>
__device__ __forceinline__ void classify_step_reg(const uint8_t *tab, int q, int r, ColState &cs, int v2, int o2, int &lv, int lhm1, NBF nb, int &o_fin)
{
	const int res = cs.v0 - cs.o0, a = cs.v1 - cs.o1, d2 = v2 - o2;
	const int v0 = cs.v0, v1 = cs.v1;
	int n1, below = cs.o1, n2;
	o_fin = cs.o0;
	int kind = classify_lookup(tab, res, a, d2);
	if (kind >= CK_NBP && kind <= CK_PREVLE0) {                    /* the four kinds that look further: two cells of the column on the right and their sign, or the row above */
		const int sg = (kind == CK_NBP || kind == CK_PREVGE0) ? 1 : -1;
		bool ok;
		if (kind <= CK_NBN) { const int x0 = sg * nb(0), x1 = sg * nb(1), x2 = sg * nb(2); ok = (x0 & ~1) == 2 && (x1 & ~1) == 2 && x2 > 0; }
		else ok = r > 0 && sg * cs.dm1 >= 0;
		kind = ok ? (sg > 0 ? CK_MARKP : CK_MARKN) : CK_NONE;
	}
	{                                                              /* (CK_NONE's word does nothing: no branch around this) */
		const uint32_t aw = reinterpret_cast<const uint32_t *>(tab + CK_ACT_OFF)[kind];
		const int code = (int)(aw & 0xFFFF);
		o_fin = code ? code : o_fin;
		if (q == 18) { const int nx = (int)(aw >> 23) & 3; below = nx ? (nx == 1 ? 14100 : 14000) : below; }
		n1 = (int16_t)(((aw >> 22) & 1) ? below : v1 + (int)((aw >> 16) & 7) - 2); n2 = (int16_t)(v2 + (int)((aw >> 19) & 7) - 2);
		const int rule = (int)(aw >> 25);
		const int bc = (lhm1 < -9 ? -9 : lhm1 > 8 ? 8 : lhm1) + 9;
		lv = (int16_t)(lv + reinterpret_cast<const int8_t *>(tab + CK_LHT_OFF)[__mul24(__mul24(rule, 9) + lh_class(lv), 18) + bc]);
	}
	cs = ColState{ n1, below, n2, o2, v0 - o_fin };
}
/* LDS: the row tiles ot (LL1 cells) and dt (recon - LL1, as loaded) of rows r0 .. r0 + CR + 1, the cells a chunk's steps leave (bytes), the LH1
 * piece tile, the tables, column 255 as Y22 left it: 20 220 bytes -- EIGHT workgroups a CU, i.e. the 16 images a CU gets of a 4096-image batch
 * in two rounds (seven were 7 + 7 + 2).
 * Cells as bytes: an LL1 cell is either a code of Y22 (12100 .. 14900, a multiple of 100) or a sample of the level-1 LL band of an 8-bit
 * picture, which is far below 12000 in magnitude (|LL1| <= (10 x (10 x 255 + 2 x 255) + ...) / 64 < 500: nhw_front_image.h) -- so what Y23
 * leaves in a cell is 0 or a code's hundredth, and "from 12000 on" means "a code". */
#define RF_LDS_BYTES (2 * (CR + 2) * H * 2 + CR * H + H * LP * 2 + CK_TABLE_BYTES + Y23_TAB_BYTES + 2 * H)
DEV void residuals_fused_par(Ctx *c, int res_setting, int tid, int16_t *lds)
{
	PROF_BEGIN();
	static_assert(NT == H && CR == 4 && LW % CR == 0, "a thread a column; a chunk's four new rows are one 8-byte item a thread");
	int16_t *p = c->proc, *o = c->ll1;
	const int q = c->q, j = tid;
	int16_t *ot = lds, *dt = lds + (CR + 2) * H, *lt = lds + 2 * (CR + 2) * H + CR * H / 2;
	uint8_t *cb = reinterpret_cast<uint8_t *>(lds + 2 * (CR + 2) * H);   /* [CR][H]: what the chunk's steps leave in the cells */
	uint8_t *ktab = reinterpret_cast<uint8_t *>(lt + H * LP), *ytab = ktab + CK_TABLE_BYTES;
	uint8_t *oc = ytab + Y23_TAB_BYTES;                                 /* [H]: column 255's LL1 cells as Y22 left them: a code's hundredth, or 0 */
	int8_t *rc = reinterpret_cast<int8_t *>(oc + H);                    /* [H]: its residuals (recon - cell) as Y22 left them, held to -128 .. 127 (Y23 compares with 0 .. 8) */
	classify_table_fill(ktab, q, res_setting, tid, NT);
	code_table_fill(ytab, q, res_setting, tid, NT);
	int lhm1, vm1, hl0_255 = 0;
	{                                                              /* ---- prologue, on packed copies in the tiles' space (5184 of its 3584 + 4608 shorts: it reaches into the piece tile) */
		int16_t *pc0 = lds, *oc0 = lds + 260, *d1 = lds + 520, *lc0 = lds + 780, *pc = lds + 1036, *ocs = lds + 1296, *lc = lds + 1556, *l0 = lds + 1812;
		int16_t *pt3 = lds + 2072, *ot3 = pt3 + 3 * H, *dt3 = ot3 + 3 * H;
		for (int t = tid; t < H + 2; t += NT) {                     /* columns 0, 1 and 255 of both planes, rows 0 .. 257 (256, 257 of ll1: its zero guard) */
			const int a0 = p[t * W], b0 = o[t * H];
			pc0[t] = (int16_t)a0; oc0[t] = (int16_t)b0; d1[t] = (int16_t)(p[t * W + 1] - o[t * H + 1]);
			pc[t] = p[t * W + H - 1]; ocs[t] = o[t * H + H - 1];
		}
		lc0[tid] = p[H + tid]; lc[tid] = p[(H - 1) * W + H + tid];  /* the LH1 coefficients of columns 0 and 255: rows 0 and 255 of the band */
		l0[tid] = p[tid * W + H];                                   /* (x, 256): the coefficient of column x's row 0 */
		if (tid == 0) l0[H] = p[H * W + H];
		if (tid < 3 * (H / 8)) {
			const int i = tid / (H / 8), c8 = 8 * (tid % (H / 8));
			const uint4 pv = *reinterpret_cast<const uint4 *>(p + i * W + c8), ov = *reinterpret_cast<const uint4 *>(o + i * H + c8);
			*reinterpret_cast<uint4 *>(pt3 + i * H + c8) = pv; *reinterpret_cast<uint4 *>(ot3 + i * H + c8) = ov;
			const uint32_t a[4] = { pv.x, pv.y, pv.z, pv.w }, b[4] = { ov.x, ov.y, ov.z, ov.w };
			uint32_t d[4];
			for (int e = 0; e < 4; e++) d[e] = ((a[e] - b[e]) & 0xFFFF) | (((a[e] >> 16) - (b[e] >> 16)) << 16);
			*reinterpret_cast<uint4 *>(dt3 + i * H + c8) = make_uint4(d[0], d[1], d[2], d[3]);
		}
		lhm1 = p[j * W + H - 1];                                    /* (j, 255): column 255 has not been visited */
		/* The two walks are serial, but most of their steps change nothing, and a step that starts from untouched values (its three rows, the row
		 * above, the coefficient before) can be evaluated by anybody: a thread a row does that first (em: the rows whose step would change
		 * something), and the walk goes from such a row on until every value the next step looks at is as it was -- then it jumps to the next row
		 * of em.  (One thread doing all 2 x 255 steps was a fifth of the kernel's time.) */
		uint64_t *em = reinterpret_cast<uint64_t *>(lds + 4376);    /* [2][4] */
		int16_t *oc0f = lds + 4408, *pcf = lds + 4668, *lcf = lds + 4928;   /* what the walks leave: column 0's cells, column 255's samples and coefficients (the packed copies stay as they were: a walk compares with them) */
		auto next_set = [&](const uint64_t *m, int r) {
			for (int w = r >> 6; w < 4; w++) { uint64_t x = m[w]; if (w == (r >> 6)) x &= ~0ull << (r & 63); if (x) return 64 * w + (int)__builtin_ctzll(x); }
			return H;
		};
		auto held = [](int d) { return (int8_t)(d < -128 ? -128 : d > 127 ? 127 : d); };
		auto hundredth = [](int cell) { return (uint8_t)(cell >= 12000 ? cell / 100 : 0); };
		BARRIER();
		if (!tid) PROF(c, 52);
		if (j < H - 1) {                                            /* step 0 of every column: column 255 reads the coefficient it leaves */
			ColState cs{ pt3[j], ot3[j], pt3[H + j], ot3[H + j], 0 };
			int lv = l0[j], of;
			classify_step_reg(ktab, q, 0, cs, pt3[2 * H + j], ot3[2 * H + j], lv, lhm1, [&](int dr) { return (int)dt3[dr * H + j + 1]; }, of);
			l0[j] = (int16_t)lv;
		}
		{                                                           /* column 0: the rows whose step changes something */
			bool eff = false;
			if (tid < H - 1) {
				const int t = tid, u = t ? t - 1 : 0;
				ColState cs{ pc0[t], oc0[t], pc0[t + 1], oc0[t + 1], t ? pc0[u] - oc0[u] : 0 };
				const int lv0 = lc0[t];
				int lv = lv0, of;
				classify_step_reg(ktab, q, t, cs, pc0[t + 2], oc0[t + 2], lv, t ? (int)lc0[u] : (int)pc[0], [&](int dr) { return (int)d1[t + dr]; }, of);
				eff = of != oc0[t] || cs.v0 != pc0[t + 1] || cs.o0 != oc0[t + 1] || cs.v1 != pc0[t + 2] || lv != lv0;
			}
			const uint64_t m = __ballot(eff);
			if ((tid & 63) == 0) em[tid >> 6] = m;
		}
		oc[tid] = 0; rc[tid] = held(pc[tid] - ocs[tid]);           /* column 255 where its walk changes nothing */
		for (int t = tid; t < H + 2; t += NT) { oc0f[t] = oc0[t]; pcf[t] = pc[t]; }
		lcf[tid] = lc[tid];
		BARRIER();
		if (!tid) PROF(c, 53);
		if (tid == 0) {                                             /* column 0: column 255 reads its LL1 cells (oc0f) */
			ColState cs{ 0, 0, 0, 0, 0 };
			int prev = 0, r = 0;
			bool untouched = true;
			while (r < H - 1) {
				if (untouched) {
					r = next_set(em, r);
					if (r >= H - 1) break;
					const int u = r ? r - 1 : 0;
					cs = ColState{ pc0[r], oc0[r], pc0[r + 1], oc0[r + 1], r ? pc0[u] - oc0[u] : 0 };
					prev = r ? (int)lc0[u] : (int)pc[0];
				}
				int lv = lc0[r], of;
				classify_step_reg(ktab, q, r, cs, pc0[r + 2], oc0[r + 2], lv, prev, [&](int dr) { return (int)d1[r + dr]; }, of);
				oc0f[r] = (int16_t)of; oc0f[r + 1] = (int16_t)cs.o0;
				untouched = cs.v0 == pc0[r + 1] && cs.o0 == oc0[r + 1] && cs.v1 == pc0[r + 2] && cs.dm1 == pc0[r] - oc0[r] && lv == lc0[r];   /* everything the next step looks at */
				prev = lv; r++;
#ifdef NHW_PROFILE
				reinterpret_cast<unsigned long long *>(c->prof)[57] += 1000;
#endif
			}
		}
		if (!tid) PROF(c, 54);
		BARRIER();
		{                                                           /* column 255: its neighbour is (x, 256) - (x + 1, 0), both as left above */
			bool eff = false;
			if (tid < H - 1) {
				const int t = tid, u = t ? t - 1 : 0;
				ColState cs{ pc[t], ocs[t], pc[t + 1], ocs[t + 1], t ? pc[u] - ocs[u] : 0 };
				const int lv0 = lc[t];
				int lv = lv0, of;
				classify_step_reg(ktab, q, t, cs, pc[t + 2], ocs[t + 2], lv, t ? (int)lc[u] : (int)pc[H - 1], [&](int dr) { return (int)l0[t + dr] - (int)oc0f[t + dr + 1]; }, of);
				eff = of != ocs[t] || cs.v0 != pc[t + 1] || cs.o0 != ocs[t + 1] || cs.v1 != pc[t + 2] || lv != lv0;
				eff |= t >= H - 3;                                    /* rows 253, 254 look at (255, 256), which step 0 may change: always walked */
			}
			const uint64_t m = __ballot(eff);
			if ((tid & 63) == 0) em[4 + (tid >> 6)] = m;
		}
		BARRIER();
		if (tid == 0) {
			ColState cs{ 0, 0, 0, 0, 0 };
			int prev = 0, r = 0;
			bool untouched = true;
			while (r < H - 1) {
				if (untouched) {
					r = next_set(em + 4, r);
					if (r >= H - 1) break;
					const int u = r ? r - 1 : 0;
					cs = ColState{ pc[r], ocs[r], pc[r + 1], ocs[r + 1], r ? pc[u] - ocs[u] : 0 };
					prev = r ? (int)lc[u] : (int)pc[H - 1];
				}
				int lv = lc[r], of;
				classify_step_reg(ktab, q, r, cs, pc[r + 2], ocs[r + 2], lv, prev, [&](int dr) { return (int)l0[r + dr] - (int)oc0f[r + dr + 1]; }, of);
				lcf[r] = (int16_t)lv;
				if (r == 0) l0[H - 1] = (int16_t)lv;                  /* (255, 256) is also this column's neighbour of row 255 */
				oc[r] = hundredth(of); rc[r] = held(cs.dm1);
				pcf[r + 1] = (int16_t)cs.v0; pcf[r + 2] = (int16_t)cs.v1; oc[r + 1] = hundredth(cs.o0); rc[r + 1] = held(cs.v0 - cs.o0);   /* what the step leaves below it */
				untouched = cs.v0 == pc[r + 1] && cs.o0 == ocs[r + 1] && cs.v1 == pc[r + 2] && cs.dm1 == pc[r] - ocs[r] && lv == lc[r];
				prev = lv; r++;
#ifdef NHW_PROFILE
				reinterpret_cast<unsigned long long *>(c->prof)[58] += 1000;
#endif
			}
			hl0_255 = pcf[H];
		}
		if (!tid) PROF(c, 56);
		BARRIER();
		vm1 = pcf[j];                                               /* (j, 255) as column 255's walk left it: where Y23 starts */
		p[(H - 1) * W + H + tid] = lcf[tid];                         /* column 255's coefficients, for the piece tile */
		BARRIER();
	}
	if (!tid) PROF(c, 10);
	/* ---- the sweep */
	const int pi = tid >> 6, pc4 = 4 * (tid & 63);                  /* my 8-byte item of a chunk's four new rows */
	auto fetch = [&](int row, uint2 &pv, uint2 &ov) { pv = *reinterpret_cast<const uint2 *>(p + row * W + pc4); ov = *reinterpret_cast<const uint2 *>(o + row * H + pc4); };
	auto put = [&](int slot, uint2 pv, uint2 ov) {
		*reinterpret_cast<uint2 *>(ot + slot * H + pc4) = ov;
		*reinterpret_cast<uint2 *>(dt + slot * H + pc4) = make_uint2(((pv.x - ov.x) & 0xFFFF) | (((pv.x >> 16) - (ov.x >> 16)) << 16), ((pv.y - ov.y) & 0xFFFF) | (((pv.y >> 16) - (ov.y >> 16)) << 16));
	};
	uint2 kp = make_uint2(0, 0), ko = kp, np, no;
	if (pi >= 2) fetch(pi - 2, kp, ko);                             /* rows 0, 1 */
	fetch(2 + pi, np, no);                                          /* rows 2 .. 5 */
	ColState cs{ 0, 0, 0, 0, 0 };
	for (int r0 = 0; r0 < H; r0 += CR) {
		if (pi >= 2) put(pi - 2, kp, ko);                           /* the chunk's first two rows: the last two the chunk before brought */
		put(2 + pi, np, no);
		kp = np; ko = no;
		if (r0 + CR < H) fetch(r0 + CR + 2 + pi, np, no);           /* the next chunk's rows are on their way while this one's columns step (up to row 257: see classify_residuals_par) */
		if (r0 % LW == 0) lh_tile_load(lt, p, r0, tid);
		BARRIER();
		if (r0 == 0) cs = ColState{ (int16_t)(dt[j] + ot[j]), ot[j], (int16_t)(dt[H + j] + ot[H + j]), ot[H + j], 0 };
#pragma unroll
		for (int i = 0; i < CR; i++) {
			const int r = r0 + i;
			int16_t *lh = lt + j * LP + r0 % LW + i;
			int lv = *lh, cell, pv;
			if (j < H - 1) {
				pv = cs.v0;
				if (r < H - 1) {
					const int o2 = ot[(i + 2) * H + j], v2 = (int16_t)(dt[(i + 2) * H + j] + o2);
					classify_step_reg(ktab, q, r, cs, v2, o2, lv, lhm1, [&](int dr) { return (int)dt[(i + dr) * H + j + 1]; }, cell);
					lhm1 = lv;
					if (r == H - 2) p[H * W + j] = (int16_t)cs.v1;        /* the last step's second sample is row 256: the first row of HL1 (the reference's walk leaves its quadrant there) */
				} else cell = cs.o0;
			} else { cell = 100 * oc[r]; pv = rc[r]; }              /* (a cell that is no code: 0, and its residual in its place) */
			const int out = code_step_tab(ytab, pv, cell, lv, vm1);
			vm1 = lv;
			cb[i * H + j] = (uint8_t)out; *lh = (int16_t)lv;
		}
		BARRIER();
		{                                                           /* (nobody writes cb again before the next barrier) */
			const uint32_t w = *reinterpret_cast<const uint32_t *>(cb + pi * H + pc4);
			*reinterpret_cast<uint2 *>(o + (r0 + pi) * H + pc4) = make_uint2((w & 0xFF) | ((w >> 8 & 0xFF) << 16), (w >> 16 & 0xFF) | ((w >> 24) << 16));
		}
		if ((r0 + CR) % LW == 0) lh_tile_store(lt, p, r0 + CR - LW, tid);
	}
	if (tid == 0) p[H * W + H - 1] = (int16_t)hl0_255;             /* (column 255's, kept until here: row 256 is column 254's neighbour as it was) */
	BARRIER();
	if (!tid) PROF(c, 11);
}

/* Y24 (nhw_encoder.c:1426-1496): every code of the code plane adds a constant to one, two or three consecutive cells of the first-order
 * plane, linearly indexed, transposed: code (r, j) to cells j * 256 + r .. */
/* The adds commute, so the pass is a gather: cell x = j * 256 + c of the first-order plane takes A0(code(c, j)) + A1(code(c - 1, j)) +
 * A2(code(c - 2, j)), code(r, j) = the code plane's cell (r, j), j < 254 -- and, since the reference indexes linearly, the first two cells
 * of a row take what codes (254, j - 1) and (255, j - 1) spill over the row end.  Row j of the result needs COLUMN j of the code plane:
 * 64 columns at a time go through LDS as bytes (a code is 121 .. 149), transposed, with the two spilling codes of the row before in front
 * of every row; then a wavefront takes a row, a lane four cells and the six bytes they look at -- all zero for nearly every lane, which
 * then touches nothing.  (Until round 3: a thread per row stepped through 32-column tiles of the plane, 0.36 ms per image at q23.) */
#define FO_TP 260                                                 /* bytes per transposed row: 2 + 256, a multiple of 4 */
__device__ static const int8_t k_fo_add[3][32] = {               /* by code - 120 */
	{ 0, -4, 4, 2, -2, -9, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5, -5, 0, 0, -3, 3, 0, 0, -8, 8, 0, 0 },
	{ 0, -3, 3, 2, -2, -3, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 },
	{ 0, 0, 0, 2, -2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 } };
DEV void adjust_first_order_par(Ctx *c, int tid, int16_t *lds /* 64 * FO_TP + 96 + 4 bytes */)
{
	int16_t *f = c->first_order;
	const int16_t *code = c->ll1;
	uint8_t *T = reinterpret_cast<uint8_t *>(lds);
	int8_t *add = reinterpret_cast<int8_t *>(T + 64 * FO_TP);      /* the table, in LDS: the lanes of a wavefront index it differently */
	uint8_t *carry = reinterpret_cast<uint8_t *>(add + 96);         /* codes (254, j), (255, j) of the last column of the band before */
	const int lane = tid & 63, wv = tid >> 6;
	if (tid < 96) add[tid] = k_fo_add[tid >> 5][tid & 31];
	if (tid < 2) carry[tid] = 0;
	for (int b = 0; b < 4; b++) {
		BARRIER();
		for (int idx = tid; idx < H * 64; idx += NT) {             /* 64 columns of every row of the code plane, transposed */
			const int r = idx >> 6, jl = idx & 63, j = 64 * b + jl;
			const unsigned u = (unsigned)(code[r * H + j] - 120);
			T[jl * FO_TP + 2 + r] = (uint8_t)((u < 30u && j < H - 2) ? u : 0u);
		}
		BARRIER();
		if (tid < 64) {                                            /* what the row before spills into a row's first two cells */
			const uint8_t s0 = tid ? T[(tid - 1) * FO_TP + 2 + H - 2] : carry[0], s1 = tid ? T[(tid - 1) * FO_TP + 2 + H - 1] : carry[1];
			T[tid * FO_TP] = s0; T[tid * FO_TP + 1] = s1;
		}
		BARRIER();
		if (tid < 2) carry[tid] = T[63 * FO_TP + 2 + H - 2 + tid];
		for (int jl = wv; jl < 64; jl += NT / 64) {
			const uint32_t *w = reinterpret_cast<const uint32_t *>(T + jl * FO_TP + 4 * lane);   /* bytes 4 l .. 4 l + 5: codes (c - 2 .. c + 3, j) of my cells c = 4 l .. */
			const uint32_t w0 = w[0], w1 = w[1] & 0xFFFFu;
			if (!(w0 | w1)) continue;
			const unsigned by[6] = { w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 & 255u, w1 >> 8 };
			uint2 *cell = reinterpret_cast<uint2 *>(f + (size_t)(64 * b + jl) * H + 4 * lane);
			uint2 v = *cell;
			int x[4] = { (int16_t)(v.x & 0xFFFF), (int16_t)(v.x >> 16), (int16_t)(v.y & 0xFFFF), (int16_t)(v.y >> 16) };
#pragma unroll
			for (int k = 0; k < 4; k++) x[k] += add[by[k + 2]] + add[32 + by[k + 1]] + add[64 + by[k]];
			v.x = (uint32_t)(uint16_t)x[0] | ((uint32_t)(uint16_t)x[1] << 16); v.y = (uint32_t)(uint16_t)x[2] | ((uint32_t)(uint16_t)x[3] << 16);
			*cell = v;
		}
	}
	BARRIER();
}

/* ---------------------------------------------------------------- Y27 (R) */
/* (:1912-2098).  LH1 and HL1 zero everything below 6 and never move a value across 6, so "loud" (|v| >= 6) of
 * any cell is the same before, during and after these two passes: the vertical neighbour test does not care
 * which row went first -> one thread per row, serial along the row.  HH1 zeroes below 7, so a 6 above (already
 * visited in raster order) reads as quiet while a 6 below (not yet visited) reads as loud: the vertical
 * neighbours are taken from a snapshot of the band, ">= 7" for the row above, ">= 6" for the row below (a cell
 * is >= 7 after its visit exactly when it was >= 7 before).  HH1 also reads column 256, which HL1's ripple may
 * have written, hence the barrier between them. */
/* What travels along a row is only what a cell's ripple did to the cell on its right -- nothing, -1 or +1 (the fourth case of the
 * reference's ripple, "8 next to -7", is dead code behind the first) -- and the ripple moves only values beyond +-7, so every loudness test
 * can be made on the values as they were.  A wavefront takes a row, a lane four cells of it: the lanes evaluate their cells for a guessed
 * incoming step (none), hand the step they produce to the lane on their right and repeat until no lane's input moves (one extra round as
 * a rule).  Rows are independent, the next one is on its way while this one is evaluated. */
struct CleanP { int thresh, lim, lim2, last_look; };
/* one cell, without branches (every lane of a row took most of the chain's branches anyway): e = the cell as the pass leaves it, dout = what
 * its ripple hands to the cell on its right */
template <int MODE>
DEV int clean_cell(const CleanP &f, int x, int n, int v1, int v2, bool look2, int &dout)
{
	const int ax = iabs(x);
	const bool keep = ax >= f.thresh, small = keep && ax < f.lim2, in_lim = x < f.lim && x > -f.lim;
	bool to7;
	if (MODE == 0) to7 = small && n < 3 && in_lim && ax > 6;
	else if (MODE == 1) to7 = small && ((n < 3 && in_lim) || !n);
	else to7 = small && n < 3;
	int e = keep ? x : 0;
	e = to7 ? (x < 0 ? -7 : 7) : e;
	/* the ripple (:1957-1976 etc.): three cases that exclude one another (e >= 8 on 0 / 1 modulo 8; e == -7 beside an 8; e < -7 on 0 / 1 modulo 8) */
	const int nv1 = -v1;
	const bool big_p = e >= 8 && (e & 7) < 2, big_n = e < -7 && ((-e) & 7) < 2;
	const bool up1 = big_n && v1 < -14 && ((nv1 & 7) == 7 || ((nv1 & 7) < 2 && look2 && v2 <= 0));
	dout = (big_p && v1 > 7 && v1 < 10000) ? -1 : up1 ? 1 : 0;
	e = (e == -7 && v1 == 8) ? -8 : e;
	return e;
}
/* One row of Y27 for the lanes' four cells each: o = the row as it is, up / dn = the rows above and below (two cells a dword: only their
 * loudness is asked for, two cells an instruction), far = the cell behind the span (HL1: column 256, lane 63 only).  Returns what the last
 * lane's ripple hands to that cell.  The lanes evaluate their cells for "nothing arrives"; where a ripple does arrive, its lane evaluates
 * its first cell again, and only if that cell now hands on something else than before do the other three follow (they never did in
 * 99 % of the rows: the second round of the whole row was a third of the pass's instructions). */
template <int MODE>
DEV int clean_row(const CleanP &f, const int o[4], uint2 up, uint2 dn, int my_far, int upt, int c0, int jb, int je, int lane, int e[4])
{
	typedef short s16x2_ __attribute__((ext_vector_type(2)));
	typedef unsigned short u16x2_ __attribute__((ext_vector_type(2)));
	const int left = __shfl_up(o[3], 1), sd0 = __shfl_down(o[0], 1), r2 = __shfl_down(o[1], 1);
	const int r1 = lane < 63 ? sd0 : my_far;
	auto loud2 = [](uint32_t w, uint32_t bias) {                  /* per half: |v| >= 0x8000 - bias */
		const s16x2_ v = __builtin_bit_cast(s16x2_, w);
		const u16x2_ a = __builtin_bit_cast(u16x2_, __builtin_elementwise_max(v, -v));
		return __builtin_bit_cast(uint32_t, (u16x2_)((a + __builtin_bit_cast(u16x2_, bias)) >> (u16x2_)(15)));
	};
	const uint32_t bu = 0x80008000u - 0x00010001u * (uint32_t)upt, b6 = 0x7FFA7FFAu;
	const uint32_t s01 = loud2(up.x, bu) + loud2(dn.x, b6), s23 = loud2(up.y, bu) + loud2(dn.y, b6);
	int n[4]; bool proc[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int j = c0 + k;
		proc[k] = j >= jb && j < je;
		const int lv = k ? o[k ? k - 1 : 0] : left, rv = k < 3 ? o[k < 3 ? k + 1 : k] : r1;
		const int lt = (MODE == 2 && j - 1 >= jb) ? 7 : 6;
		n[k] = (iabs(lv) >= lt) + (iabs(rv) >= 6) + (int)(((k < 2 ? s01 : s23) >> (16 * (k & 1))) & 3u);
	}
	int din = 0, dd = 0, d0out = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int x = o[k] + dd;
		if (proc[k]) {
			const int v1 = k < 3 ? o[k < 3 ? k + 1 : k] : r1, v2 = k < 2 ? o[k < 2 ? k + 2 : k] : (k == 2 ? r1 : r2);
			e[k] = clean_cell<MODE>(f, x, n[k], v1, v2, c0 + k < f.last_look, dd);
		} else { e[k] = x; dd = 0; }
		if (k == 0) d0out = dd;
	}
	int dout = dd;
	for (;;) {
		int nd = __shfl_up(dout, 1);
		if (!lane) nd = 0;
		if (!__any(nd != din)) break;
		din = nd;
		int d0 = 0;
		{
			const int x = o[0] + din;
			if (proc[0]) e[0] = clean_cell<MODE>(f, x, n[0], o[1], o[2], c0 < f.last_look, d0);
			else e[0] = x;
		}
		if (__any(d0 != d0out)) {                                  /* the first cell hands on something else: the rest of the lane's cells again */
			d0out = d0; dd = d0;
#pragma unroll
			for (int k = 1; k < 4; k++) {
				const int x = o[k] + dd;
				if (proc[k]) {
					const int v1 = k < 3 ? o[k < 3 ? k + 1 : k] : r1, v2 = k < 2 ? o[k < 2 ? k + 2 : k] : (k == 2 ? r1 : r2);
					e[k] = clean_cell<MODE>(f, x, n[k], v1, v2, c0 + k < f.last_look, dd);
				} else { e[k] = x; dd = 0; }
			}
			dout = dd;
		}
	}
	return dout;
}
/* Y27 with every row read once (round 5).  Until round 4 the rows of a band went to the wavefronts in turn (row r to wavefront
 * r mod 4), so a row's two neighbours came from memory again with it -- three reads a row -- and HH1, whose test of the row below wants that
 * row as it was BEFORE the pass, worked against a snapshot of the whole band (a copy out and a read back): 1.25 MB per image for bands that
 * weigh 0.77 MB read + written.  Here a wavefront takes 64 consecutive rows and keeps the row above, the row itself and the row below AS
 * THEY WERE in registers, the next row on its way: a row is read once and written once, there is no snapshot.  LH1 and HL1 do not care
 * which version of a neighbour row they see (the pass never moves a value across 6); HH1's look up asks ">= 7", which is the same before
 * and after a row's visit, and its look down needs the raw row -- which the rolling registers hold, except for the first row of the next
 * wavefront's range: that one is read before anybody starts (one barrier).  A row of the lower half is its HL1 half, then its HH1 half (the
 * ripple HL1 may hand to column 256 goes to HH1's first lane over a readlane instead of through memory). */
DEV void clean_details_seq(Ctx *c, int tid, bool ll_in_plane /* Y26 has put the level-2 block back into the work plane (else HL1's first row looks up into its copy l2save) */)
{
	int16_t *p = c->proc;
	const int q = c->q, lane = tid & 63, wv = tid >> 6;
	const CleanP fa = { DEADZONE - 2, q > 22 ? 8 : 9, q > 22 ? 4 : 9, W - 2 };
	const CleanP fb = { DEADZONE - 2, q > 17 ? 8 : 9, q > 22 ? 4 : 9, H - 2 };
	const int lim = q > 22 ? 8 : 11;
	const CleanP fc = { DEADZONE - 1, lim, lim, W - 2 };
	auto ld = [&](int r, int c0) { return *reinterpret_cast<const uint2 *>(p + (size_t)r * W + c0); };
	auto st = [&](int r, int c0, const int e[4]) {
		uint2 w;
		w.x = (uint32_t)(uint16_t)e[0] | ((uint32_t)(uint16_t)e[1] << 16); w.y = (uint32_t)(uint16_t)e[2] | ((uint32_t)(uint16_t)e[3] << 16);
		*reinterpret_cast<uint2 *>(p + (size_t)r * W + c0) = w;
	};
	const int cl = 4 * lane, ch = H + 4 * lane;                     /* my four cells of a row's left / right half */
	const int lo0 = H + 64 * wv, lo1 = lo0 + 63 < W - 2 ? lo0 + 63 : W - 2;   /* my rows of the lower half */
	const uint2 hh_below = ld(lo1 + 1, ch);                         /* the HH1 row below my range, before the wavefront that owns it rewrites it */
	BARRIER();
	{                                                               /* LH1: rows 1 .. 254, columns 257 .. 510 */
		const int r0 = 1 + 64 * wv, r1 = r0 + 63 < H - 2 ? r0 + 63 : H - 2;
		uint2 up = ld(r0 - 1, ch), cur = ld(r0, ch), dn = ld(r0 + 1, ch);
		for (int r = r0; r <= r1; r++) {
			const uint2 nx = ld(r + 2 < W ? r + 2 : r + 1, ch);      /* (row r1 + 2 is never used) */
			int o[4], e[4];
			unpack4(cur, o);
			clean_row<0>(fa, o, up, dn, 0, 6, ch, H + 1, W - 1, lane, e);
			st(r, ch, e);
			up = cur; cur = dn; dn = nx;
		}
	}
	{                                                               /* rows 256 .. 510: HL1 (columns 1 .. 255), then HH1 (columns 257 .. 510) */
		uint2 upl = (wv == 0 && !ll_in_plane) ? *reinterpret_cast<const uint2 *>(c->l2save + (size_t)(H - 1) * H + cl) : ld(lo0 - 1, cl), curl = ld(lo0, cl), dnl = ld(lo0 + 1, cl);   /* row 255 under HL1's first row is the last row of the level-2 block */
		uint2 uph = ld(lo0 - 1, ch), curh = ld(lo0, ch), dnh = lo0 + 1 > lo1 ? hh_below : ld(lo0 + 1, ch);
		for (int r = lo0; r <= lo1; r++) {
			const int rn = r + 2 < W ? r + 2 : W - 1;
			const uint2 nxl = ld(rn, cl);
			const uint2 nxh = r + 2 > lo1 ? (r + 2 == lo1 + 1 ? hh_below : make_uint2(0, 0)) : ld(rn, ch);
			int o[4], e[4];
			unpack4(curl, o);
			const int my_far = (int16_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)curh.x) & 0xFFFFu);   /* column 256 of the row: HL1's last cell looks at it, and its ripple may move it */
			const int dout = clean_row<1>(fb, o, upl, dnl, my_far, 6, cl, 1, H, lane, e);
			st(r, cl, e);
			const int moved = __builtin_amdgcn_readlane(dout, 63);    /* HL1's ripple out of column 255 lands in column 256: HH1's first cell (not processed itself, but its neighbour's left) */
			unpack4(curh, o);
			if (lane == 0) o[0] += moved;
			clean_row<2>(fc, o, uph, dnh, 0, r > H ? 7 : 6, ch, H + 1, W - 1, lane, e);
			st(r, ch, e);
			upl = curl; curl = dnl; dnl = nxl;
			uph = curh; curh = dnh; dnh = nxh;
		}
	}
}
/* ---------------------------------------------------------------- a10 quantiser */
/* (the luma quantiser is a wavefront-per-image kernel for every quality: wave_quantise_luma, nhw_tail_wave.h) */

/* offsetUV (image_processing.c:108-183), a wavefront per row (lane l: columns l, l + 64, l + 128, l + 192), 64 consecutive rows per wavefront.
 * The reference's walk along a row has two ways of reaching into the next cell: a pair of -7 / -8 neighbours becomes two 120s and the walk
 * skips the second; a cell above 6 on residue 6 / 7 raises a 7 behind it to 8 (which then raises nothing itself).  Both only test values
 * from before the walk, so each is "fires if visited / if not raised" along runs of candidates: the cells at even distance from the head
 * of a run (alt_runs); the two never meet (one is about negative cells, the other about positive ones).  Everything else looks at the cell
 * and at the cell behind it as it was; the look from column 255 goes to the first cell of the next row, unguarded, before any row is
 * rewritten (nf: those cells, read before the pass).
 * The quantised plane only feeds the symbol stream, in serpentine order: 32 strips of 8 columns, within a strip row after row, odd rows
 * right to left, U in the even and V in the odd bytes (nhw_encoder.c:2553-2570): 16 rows are parked as bytes in a wave-private LDS block and
 * leave as 64-byte runs (8 rows of a strip) per lane -- U into a byte plane of its own, V merged with it into the stream.  The int16
 * plane is only written for the tests' stage check.
 * (Until round 3 this was a thread per row on eight 32-column LDS tiles: 72-byte row pieces in and out, 0.93 + 0.39 MB per plane moved
 * for 128 KB of coefficients and 64 KB of symbols.) */
#define CQROW 264
#define CQ_LDS_BYTES (4 * 16 * CQROW + 2 * (H + 2))
DEV void quantise_chroma_par(Ctx *c, int comp, int tid, int16_t *lds, bool write_plane, bool dense /* the merged byte stream as well (stage checks) */)
{
	unsigned vtotal = 0;                                            /* V: values this wavefront has appended to the list (wave-uniform) */
	int16_t *p = c->cproc;
	uint8_t *ubytes = c->ubytes;                                    /* Q bytes of its own (until round 5: the band plane, which Y29 holds above q21) */
	uint8_t *scan = c->scan + 4 * Q;
	const int lane = tid & 63, wv = tid >> 6;
	uint8_t *park = reinterpret_cast<uint8_t *>(lds) + wv * 16 * CQROW;
	int16_t *nf = lds + 4 * 16 * CQROW / 2;                        /* nf[r]: cell (r, 0) as it was (r = 256: what lies behind the plane) */
	nf[tid + 1] = p[(tid + 1) * H];
	BARRIER();
	int cur[4], nxt[4] = { 0, 0, 0, 0 };
	const int r0 = 64 * wv;
	for (int k = 0; k < 4; k++) cur[k] = p[r0 * H + lane + 64 * k];
	for (int i = 0; i < 64; i++) {
		const int r = r0 + i;
		if (i + 1 < 64) for (int k = 0; k < 4; k++) nxt[k] = p[(r + 1) * H + lane + 64 * k];
		const int first_next = nf[r + 1];
		unsigned P, S, T;
		BS_PRED(P, cur, 4, (unsigned)(x + 8) < 2u); BS_PRED(S, cur, 4, x > 6 && x <= 127 && (x & 7) >= 6); BS_PRED(T, cur, 4, x == 7);
		const unsigned inrow = lane == 63 ? 0x7u : 0xFu;           /* columns 0..254: neither reaches across the row end */
		const unsigned pc = P & bs_dn(P, lane) & inrow, bc = S & bs_dn(T, lane) & inrow;
		const unsigned pf = __any(pc != 0) ? bs_from4(alt_runs(bs_ballot4(pc))) : 0u;
		const unsigned bf = __any(bc != 0) ? bs_from4(alt_runs(bs_ballot4(bc))) : 0u;
		const unsigned pair = pf | bs_up<4>(pf, lane), raised = bs_up<4>(bf, lane);
		for (int k = 0; k < 4; k++) {
			const int x = cur[k], nx = right_of_dpp(cur, k, 4, lane, first_next);
			int a = (raised >> k) & 1 ? 8 : x;
			const bool neg = a < 0;
			int m = neg ? -a : a;
			const bool keep = (nx < 0 && nx > -8) ? (m & 7) >= 6 : (m & 7) == 7;
			m = (neg && !keep) ? (m & 504) : m;
			a = neg ? -m : m;
			int sym = (unsigned)(a + DEADZONE - 1) < (unsigned)(2 * DEADZONE - 1) ? 128 : ((a + 128) & 248);
			sym = (pair >> k) & 1 ? 120 : sym;
			if (__ballot(x > 127 || x < -127)) {                     /* marks of the pass before and values beyond +-127: rare, whole words skip this */
				if (x == 12400) sym = 124; else if (x == 12600) sym = 126; else if (x == 12900) sym = 122; else if (x == 13000) sym = 130;
				else if (x > 127) sym = big_code(x, k_big_pos);
				else if (x < -127) sym = big_code(-x, k_big_neg);
			}
			if (write_plane) p[r * H + lane + 64 * k] = (int16_t)sym;
			park[(i & 15) * CQROW + lane + 64 * k] = (uint8_t)sym;
		}
		if ((i & 15) == 15) {                                      /* 16 rows complete: lane l takes rows 8 (l & 1) .. + 7 of strip l >> 1, 64 stream bytes */
			__threadfence_block();
			const int strip = lane >> 1, rb = r - 15 + 8 * (lane & 1);
			uint32_t w[16];
			for (int j = 0; j < 8; j++) {
				const uint2 x = *reinterpret_cast<const uint2 *>(park + (8 * (lane & 1) + j) * CQROW + 8 * strip);
				if ((rb + j) & 1) { w[2 * j] = __builtin_bswap32(x.y); w[2 * j + 1] = __builtin_bswap32(x.x); } else { w[2 * j] = x.x; w[2 * j + 1] = x.y; }
			}
			const int pos = strip * (8 * H) + 8 * rb;
			if (!comp) { for (int j = 0; j < 4; j++) reinterpret_cast<uint4 *>(ubytes + pos)[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]); }
			else {
				uint32_t o[32];
				for (int j = 0; j < 4; j++) {
					const uint4 u4 = reinterpret_cast<const uint4 *>(ubytes + pos)[j];
					const uint32_t uu[4] = { u4.x, u4.y, u4.z, u4.w };
					for (int e = 0; e < 4; e++) {
						const uint32_t u = uu[e], v = w[4 * j + e];
						o[8 * j + 2 * e] = (u & 0xFF) | ((v & 0xFF) << 8) | ((u & 0xFF00) << 8) | ((v & 0xFF00) << 16);
						o[8 * j + 2 * e + 1] = ((u >> 16) & 0xFF) | (((v >> 16) & 0xFF) << 8) | ((u >> 24) << 16) | ((v >> 24) << 24);
					}
				}
				if (dense) for (int j = 0; j < 8; j++) reinterpret_cast<uint4 *>(scan + 2 * pos)[j] = make_uint4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
				/* the merged stream leaves as a list, like the luma part (wave_quantise_luma): the lane's 128 symbols are two slices of 64 -- a
				 * non-zero map each, [flush][lane][2] -- and their symbols that are not 128 go behind those of the lanes before it (the wavefronts
				 * of the workgroup run side by side: each appends to a region of its own, 32768 x wavefront, the size of all its symbols).  The
				 * packetiser's chroma part puts the maps into stream order (pack_chroma_order). */
				const int F = 4 * wv + (i >> 4);
				const uint64_t M0 = (uint64_t)ne_mask32(o, 0x80808080u) | (uint64_t)ne_mask32(o + 8, 0x80808080u) << 32;
				const uint64_t M1 = (uint64_t)ne_mask32(o + 16, 0x80808080u) | (uint64_t)ne_mask32(o + 24, 0x80808080u) << 32;
				*reinterpret_cast<uint4 *>(c->cnzq + (F * 64 + lane) * 2) = make_uint4((uint32_t)M0, (uint32_t)(M0 >> 32), (uint32_t)M1, (uint32_t)(M1 >> 32));
				const unsigned cnt = (unsigned)(__builtin_popcountll(M0) + __builtin_popcountll(M1));
				unsigned incl = cnt;
				for (int o_ = 1; o_ < 64; o_ <<= 1) { const unsigned t_ = (unsigned)__shfl_up((int)incl, o_); if (lane >= o_) incl += t_; }
				if (lane == 0) { c->cfbase[F] = 32768u * wv + vtotal; c->cfbase[16 + wv] = vtotal + (unsigned)__builtin_amdgcn_readlane((int)incl, 63); }   /* [16 + wv]: the wavefront's values so far (its total behind the last flush) */
				uint8_t *vp = c->cvals + 32768u * wv + vtotal + incl - cnt;
				vtotal += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
				/* the symbols by position: a slice's sixteen dwords wait in the wavefront's parked rows -- every lane has read its own above, the
				 * next sixteen rows have not begun -- word k of lane l at dword 64 k + l (no bank conflicts), and leave byte by byte for the set bits */
				uint32_t *sw = reinterpret_cast<uint32_t *>(park);
				static_assert(16 * CQROW >= 16 * 64 * 4, "a slice of every lane fits the parked rows");
#pragma unroll
				for (int hs = 0; hs < 2; hs++) {
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
					for (int k = 0; k < 16; k++) sw[64 * k + lane] = o[16 * hs + k];
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
					const uint8_t *sb = reinterpret_cast<const uint8_t *>(sw) + 4 * lane;
					for (uint64_t m = hs ? M1 : M0; m; m &= m - 1) {
						const int bit = __builtin_ctzll(m);
						*vp++ = sb[256 * (bit >> 2) + (bit & 3)];
					}
				}
			}
			__threadfence_block();
		}
		for (int k = 0; k < 4; k++) cur[k] = nxt[k];
	}
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
DEV void scan_and_rewrite_par(Ctx *c, int tid, int *sh_counts, uint32_t *sh_z /* shared [n/16/32 + 2] */, int16_t *lds /* the list of run starts */)
{
	const int16_t *p = c->proc;
	uint8_t *s = c->scan;
	const int n = 4 * Q;
	PROF_BEGIN();
	uint32_t *bits = reinterpret_cast<uint32_t *>(c->half);      /* n bits of selection flags */

	if (tid == 0) { sh_counts[0] = 0; sh_counts[1] = 0; }
	BARRIER();

	/* The three rewrites read 16 stream bytes per thread and step (48-byte register window: 16 before, 16 own,
	 * 16 after), so a wavefront touches 1 KiB of consecutive memory per load instruction. */
#define WIN_LOAD(dst_, base) do { const uint4 a_ = *reinterpret_cast<const uint4 *>(s + (base) - 16), b_ = *reinterpret_cast<const uint4 *>(s + (base)), \
		c_ = (base) + 16 < n ? *reinterpret_cast<const uint4 *>(s + (base) + 16) : make_uint4(0, 0, 0, 0);   /* im_nhw is calloc'ed and the chroma part not yet written when the reference is here: bytes behind the luma part read 0 (the chroma sequence may be writing them on its own stream) */ \
		dst_[0] = a_.x; dst_[1] = a_.y; dst_[2] = a_.z; dst_[3] = a_.w; dst_[4] = b_.x; dst_[5] = b_.y; dst_[6] = b_.z; dst_[7] = b_.w; dst_[8] = c_.x; dst_[9] = c_.y; dst_[10] = c_.z; dst_[11] = c_.w; } while (0)
#define WB(w, k) ((int)(((w)[((k) + 16) >> 2] >> (8 * (((k) + 16) & 3))) & 0xFF))      /* byte at base + k, -16 <= k < 32 */
#define PM8(v) ((v) == 136 || (v) == 120)
	/* one bit per window byte 8 .. 39 (bit j = byte base - 16 + j) that equals the byte replicated in `pat`: all a test on an own byte looks at */
#define WIN_MASK(w, pat) ((unsigned long long)(~ne_mask32((w) + 2, pat)) << 8)
	/* 4-bit mask of the bytes of a word that are the zero symbol 128 (exact per byte, then the four flags gathered by a multiply) */
#define Z4(x) ((((~((((x) ^ 0x80808080u) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu | ((x) ^ 0x80808080u) | 0x7F7F7F7Fu)) >> 7) * 0x00204081u) >> 21 & 15u)
	/* any +-8 symbol (136 / 120) among the 16 own bytes of a window?  (zero-byte test on the words xor-ed with the symbol) */
#define HASZ(x) ((((x) - 0x01010101u) & ~(x)) & 0x80808080u)
#define ANY_PM8(w) ((HASZ((w)[4] ^ 0x88888888u) | HASZ((w)[5] ^ 0x88888888u) | HASZ((w)[6] ^ 0x88888888u) | HASZ((w)[7] ^ 0x88888888u) | \
                     HASZ((w)[4] ^ 0x78787878u) | HASZ((w)[5] ^ 0x78787878u) | HASZ((w)[6] ^ 0x78787878u) | HASZ((w)[7] ^ 0x78787878u)) != 0)
	for (int w = tid; w < n / 32; w += NT) bits[w] = 0;
	if (tid < 4) sh_z[n / 16 / 32 + tid] = 0;
	BARRIER();
	if (!tid) PROF(c, 40);
	for (int base = 16 * tid; base < n; base += 16 * NT) {         /* rewrite 1, selection */
		uint32_t w[12];
		WIN_LOAD(w, base);
		{                                                          /* bitmap of all-zero 16-byte groups (never changes: rewrites only touch non-zero symbols) */
			const unsigned long long mask = __ballot(w[4] == 0x80808080u && w[5] == 0x80808080u && w[6] == 0x80808080u && w[7] == 0x80808080u);
			if ((tid & 63) == 0) { sh_z[base >> 9] = (uint32_t)mask; sh_z[(base >> 9) + 1] = (uint32_t)(mask >> 32); }
		}
		if (!ANY_PM8(w)) continue;
		/* The tests on the 16 own symbols are done on all of them at once: one bit per window byte (bit j = byte base - 16 + j) for "is the
		 * zero symbol" and "is +-8", the patterns as shifted ANDs; only the hits are visited one by one. */
		const unsigned long long Zm = WIN_MASK(w, 0x80808080u), P = WIN_MASK(w, 0x88888888u) | WIN_MASK(w, 0x78787878u);
		unsigned long long hit = P & (Zm >> 1) & (Zm >> 2) & (Zm >> 3) & (P >> 4) & 0xFFFF0000ull;       /* (+-8, 0, 0, 0, +-8) starting at an own byte */
		if (base > n - 5 - 15) hit &= (1ull << (n - 4 - base + 16)) - 1;                                /* cpos <= n - 5 */
		const unsigned long long prev = (P << 4) & (Zm << 3) & (Zm << 2) & (Zm << 1);                   /* the same pattern one step back */
		while (hit) {
			const int j = __ffsll((long long)hit) - 1, cpos = base + j - 16;
			hit &= hit - 1;
			int m = 1, back = cpos - 4;
			if (back >= 0 && ((prev >> j) & 1)) {                  /* only a chain of them goes on through memory */
				m++; back -= 4;
				while (pair_cand(s, back, n)) { m++; back -= 4; }
			}
			if (m & 1) atomicOr(&bits[cpos >> 5], 1u << (cpos & 31));
		}
	}
	BARRIER();
	if (!tid) PROF(c, 41);
	for (int w = tid; w < n / 32; w += NT) {                       /* rewrite 1, application */
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

	if (!tid) PROF(c, 42);
	{                                                              /* rewrite 2 (tests on the window; writes are byte stores) */
		int n1 = 0, n2 = 0;
		for (int base = 16 * tid; base < n; base += 16 * NT) {
			uint32_t w[12];
			WIN_LOAD(w, base);
			if (!ANY_PM8(w)) continue;
			const unsigned long long Zm = WIN_MASK(w, 0x80808080u), P6 = WIN_MASK(w, 0x88888888u), P0 = WIN_MASK(w, 0x78787878u), P = P6 | P0;
			unsigned long long act = P & 0xFFFF0000ull;            /* own +-8 symbols at 4 <= i < n - 4 */
			unsigned long long left = P << 1;                      /* the left neighbour is +-8 (asked only for i > 4) */
			if (base == 0) { act &= ~0xFull << 16; left &= ~0x1Full << 16; }
			if (base == n - 16) act &= (1ull << 28) - 1;
			const unsigned long long before4 = (Zm << 1) & (Zm << 2) & (Zm << 3) & (Zm << 4);
			const unsigned long long after3 = (Zm >> 2) & (Zm >> 3) & (Zm >> 4);                        /* bytes +2 .. +4 */
			/* did the left neighbour take me as the second of a pair? */
			const unsigned long long taken = left & (Zm >> 1) & (((Zm << 2) & (Zm << 3) & (Zm << 4) & (Zm << 5)) | ((Zm << 2) & after3));
			act &= ~taken;
			const unsigned long long pairA = act & (P >> 1) & (Zm >> 2) & (before4 | ((Zm << 1) & (Zm >> 3) & (Zm >> 4) & (Zm >> 5)));
			const unsigned long long lone = act & ~pairA & (Zm >> 1) & (before4 | ((Zm << 1) & after3));
			for (unsigned long long h = pairA; h; h &= h - 1) {
				const int j = __ffsll((long long)h) - 1;
				s[base + j - 15] = (uint8_t)(((P0 >> (j + 1)) & 1) ? 157 : 159);
			}
			for (unsigned long long h = lone; h; h &= h - 1) {
				const int j = __ffsll((long long)h) - 1;
				s[base + j - 16] = (uint8_t)(((P6 >> j) & 1) ? 153 : 155);
			}
			n2 += __popcll(pairA); n1 += __popcll(lone);
		}
		if (n1) atomicAdd(&sh_counts[0], n1);
		if (n2) atomicAdd(&sh_counts[1], n2);
	}
	BARRIER();
	if (tid == 0) { c->m->select1 = sh_counts[0]; c->m->select2 = sh_counts[1]; }

	if (!tid) PROF(c, 43);
	/* rewrite 3: runs of zero symbols.  Only a run of 252 or more does anything, and one that long covers the rest of the group it starts
	 * in and the 14 groups behind it, so a group holds at most one such start: the first of its trailing zero symbols.  Pass one finds the
	 * starts (from the bitmap of all-zero groups alone inside a zero region: nothing is loaded there) and lists them; pass two deals them out
	 * to the threads, so that the dependent loads of a run (its end, the symbols behind it) are a chain of two or three per thread and not
	 * one in every step of a 64-step sweep. */
	int *cand = reinterpret_cast<int *>(lds);                      /* [0]: count, then the start positions (at most one per 15 groups) */
	if (!tid) cand[0] = 0;
	BARRIER();
	for (int base = 16 * tid; base < n; base += 16 * NT) {
		const int gi = base >> 4;
		unsigned long long zz;                                     /* bit 0: the group before is all zero symbols, bit 1: this one, bits 2..15: the 14 behind it */
		if (gi) { const int g = gi - 1; zz = ((unsigned long long)sh_z[(g >> 5) + 1] << 32 | sh_z[g >> 5]) >> (g & 31); }
		else zz = ((unsigned long long)sh_z[1] << 32 | sh_z[0]) << 1;
		if (((zz >> 2) & 0x3FFF) != 0x3FFF) continue;
		int i;
		if ((zz >> 1) & 1) {                                       /* all zero: a start only if the run does not come from the group before */
			if (zz & 1) continue;
			if (base > 0 && s[base - 1] == 128) continue;
			i = base;
		} else {
			const uint4 o = *reinterpret_cast<const uint4 *>(s + base);
			const unsigned own = Z4(o.x) | (Z4(o.y) << 4) | (Z4(o.z) << 8) | (Z4(o.w) << 12);   /* bit k: byte base + k is the zero symbol */
			const int tz = __clz((int)~(own << 16));               /* trailing zero symbols of the group */
			if (tz == 0) continue;
			i = base + 16 - tz;
		}
		cand[1 + atomicAdd(&cand[0], 1)] = i;
	}
	BARRIER();
	const int ncand = cand[0];
	for (int ci = tid; ci < ncand; ci += NT) {
		const int i = cand[1 + ci];
		{
			/* run [i, b]: hop over the all-zero groups with the bitmap, finish on one 16-byte load of the group the run ends in */
			int b;
			{
				int g2 = (i >> 4) + 1;                             /* first group not yet examined (all zero, like the thirteen behind it) */
				for (;;) {
					const uint32_t inv = ~(sh_z[g2 >> 5] >> (g2 & 31));
					const int room = 32 - (g2 & 31);
					const int z = inv ? __ffs((int)inv) - 1 : 32;
					if (z >= room) { g2 += room; if (g2 >= n / 16) break; continue; }
					g2 += z; break;
				}
				b = 16 * g2 - 1;                                   /* last byte of the last all-zero group */
				if (b + 1 < n) {
					const uint4 e = *reinterpret_cast<const uint4 *>(s + b + 1);
					const unsigned lead = Z4(e.x) | (Z4(e.y) << 4) | (Z4(e.z) << 8) | (Z4(e.w) << 12);
					b += __ffs((int)~lead) - 1;                    /* not all sixteen: the bitmap said so */
				}
			}
			/* the reference's walk (:2222-2252) fires every 254 cells from i+255 on -- at kk = i + 255 + 254 t <= b - 1 it looks at the four
			 * symbols from kk on -- then once at the end: closed form.  Inside the run those are zero symbols, which a fix leaves alone: only
			 * the last firing can reach behind the run (a run of 200 000 zeros, the rule at quality 1, is 800 firings that do nothing: walked
			 * one by one by the thread that owns the run, they were 1.3 of the 1.5 ms of this pass there) */
			{
				const int fired = b - i >= 256 ? (b - i - 256) / 254 + 1 : 0;
				if (fired) { const int kk = i + 255 + 254 * (fired - 1); for (int at = b + 1; at <= kk + 3; at++) if (at < n) fix_sign_code(s, at); }
				const int tail_run = fired ? b - (i + 255 + 254 * (fired - 1)) + 1 : b - i;
				if (tail_run >= 252 && b + 1 < n) fix_sign_code(s, b + 1);
			}
		}
	}
	BARRIER();
	if (!tid) PROF(c, 44);
#undef WIN_LOAD
#undef WB
#undef PM8
#undef Z4
#undef WIN_MASK
#undef HASZ
#undef ANY_PM8
}


/* ---------------------------------------------------------------- Y31 on the symbol LIST (round 5)
 * The same three rewrites (:2134-2252) on what the luma quantiser leaves since round 5 (wave_quantise_luma): a 64-bit non-zero map per
 * slice of 64 stream symbols and the non-zero symbols themselves.  Every rule of Y31 asks two things of a symbol -- is it the zero symbol,
 * is it a +-8 (136 / 120) -- and only ever rewrites symbols that are not zero into others that are not zero, so the map is its "is zero"
 * plane (whole-word shifts answer the (+-8, 0, 0, 0, +-8) and the "four zeros before / three behind" questions for 64 positions at once), the
 * +-8 masks of a slice are built from its handful of values, and a rewrite is a byte store into the value list.  The map and the slices'
 * value offsets sit in LDS in stream order (the quantiser writes them flush-major); the few questions that cross a slice edge go through
 * sl_sym().  What leaves: the map and the offsets in stream order (c->nzs, c->voff) for the packetiser, the values rewritten in place.
 * The two symbols the reference zeroes at either end of the stream (:2169-2176) are cleared bits: the values stay where they are, dead. */
#define SL_SLICES (4 * Q / 64)
#define SL_LDS_BYTES (SL_SLICES * 12 + 64)   /* with the kernel's own 2 KB: three workgroups to a CU */
#define SL_OFF(x) ((x) & 0x1FFFFFFFu)          /* a slice's offset word: bits 29..31 say how many symbols at its head a 132..135 code of the slice before covers */
struct SymList { uint64_t *nz; uint32_t *vo; uint8_t *vals; };      /* nz, vo: LDS, stream order */
DEV int sl_sym(const SymList &L, int pos)                            /* the symbol at stream position 0 <= pos < 4 Q */
{
	const uint64_t M = L.nz[pos >> 6];
	const int bit = pos & 63;
	if (!((M >> bit) & 1)) return 128;
	return L.vals[SL_OFF(L.vo[pos >> 6]) + (unsigned)__builtin_popcountll(M & ((1ull << bit) - 1))];
}
DEV void sl_set(const SymList &L, int pos, int v)                    /* pos holds a symbol that is not zero */
{
	const uint64_t M = L.nz[pos >> 6];
	L.vals[SL_OFF(L.vo[pos >> 6]) + (unsigned)__builtin_popcountll(M & ((1ull << (pos & 63)) - 1))] = (uint8_t)v;
}
/* slice gi's map; what lies outside the stream is "not zero" (the reference finds a 0 byte there, which is not the zero symbol 128) */
DEV uint64_t sl_word(const SymList &L, int gi) { return (gi < 0 || gi >= SL_SLICES) ? ~0ull : L.nz[gi]; }
DEV bool sl_zero(const SymList &L, int pos) { return pos >= 0 && pos < 4 * Q && !((L.nz[pos >> 6] >> (pos & 63)) & 1); }
DEV bool sl_cand(const SymList &L, int c)                            /* (+-8, 0, 0, 0, +-8) starting at c */
{
	if (c < 0 || c > 4 * Q - 5) return false;
	if (!(sl_zero(L, c + 1) && sl_zero(L, c + 2) && sl_zero(L, c + 3))) return false;
	return is_pm8(sl_sym(L, c)) && is_pm8(sl_sym(L, c + 4));
}
/* the +-8 masks of slice g from its values (bit k: symbol k is 136 / is 120); `first`: the slice's first sixteen values, asked for a few
 * slices ahead (a dependent byte load per value was most of this pass's time: a slice holds one or two) */
DEV void sl_pm8(const SymList &L, int g, uint64_t M, uint4 first, uint64_t *p6, uint64_t *p0)
{
	uint64_t a = 0, b = 0;
	uint64_t lo = (uint64_t)first.x | (uint64_t)first.y << 32, hi = (uint64_t)first.z | (uint64_t)first.w << 32;
	const uint8_t *v = L.vals + SL_OFF(L.vo[g]) + 16;
	int r = 0;
	for (uint64_t m = M; m; m &= m - 1, r++) {
		int x;
		if (r < 16) { x = (int)(lo & 0xFF); lo = (lo >> 8) | (hi << 56); hi >>= 8; } else x = *v++;
		const uint64_t bit = m & (0 - m);
		if (x == 136) a |= bit; else if (x == 120) b |= bit;
	}
	*p6 = a; *p0 = b;
}
#ifndef SL_AHEAD
#define SL_AHEAD 4                                                 /* slices a thread has in flight */
#endif
#define SL_FETCH(Mk, vk, g0) do { for (int k_ = 0; k_ < SL_AHEAD; k_++) { const int g_ = (g0) + TN * k_; Mk[k_] = L.nz[g_]; \
		if (Mk[k_]) __builtin_memcpy(&vk[k_], L.vals + SL_OFF(L.vo[g_]), 16); } } while (0)
/* TN: threads of the workgroup, 256 (inside k_phase<L4D>, with the stage checks' dense form behind it) or 512 (k_y31: the kernel's 49 KB of LDS
 * hold a CU to three workgroups whatever their size, so twice the threads are twice the wavefronts a CU for a pass that is bound by its
 * threads' dependent loads) */
DEV unsigned block_exscan_max(unsigned v, int tid, unsigned *shm);
template <int TN>
DEV void scan_rewrite_list_par(Ctx *c, int tid, uint8_t *lds /* SL_LDS_BYTES */, int *sh_counts)
{
	const int n = 4 * Q;
	SymList L;
	L.nz = reinterpret_cast<uint64_t *>(lds); L.vo = reinterpret_cast<uint32_t *>(lds + SL_SLICES * 8); L.vals = c->vals;

	uint32_t *sel = reinterpret_cast<uint32_t *>(c->half);          /* the selected pairs of rewrite 1 (at most n / 8 of them) */
	PROF_BEGIN();
	if (tid == 0) sh_counts[0] = 0;
	{                                                               /* the map into stream order, every slice's first value: thread = (flush, sixteen strips) */
		constexpr int SPT = SL_SLICES / TN, TPF = 128 / SPT;          /* strips a thread takes of its flush, threads a flush */
		const int f = tid / TPF, s0 = (tid % TPF) * SPT;
		uint64_t m[SPT];
		unsigned tot = 0;
		for (int k = 0; k < SPT; k++) { m[k] = c->nzq[f * 128 + s0 + k]; tot += (unsigned)__builtin_popcountll(m[k]); }
		unsigned incl = tot;
		for (int o = 1; o < TPF; o <<= 1) { const unsigned t_ = (unsigned)__shfl_up((int)incl, o, TPF); if ((tid % TPF) >= o) incl += t_; }
		unsigned at = c->fbase[f] + incl - tot;
		for (int k = 0; k < SPT; k++) { L.nz[(s0 + k) * 32 + f] = m[k]; L.vo[(s0 + k) * 32 + f] = at; at += (unsigned)__builtin_popcountll(m[k]); }
	}
	BARRIER();
	if (!tid) PROF(c, 40);
	for (int g0 = tid; g0 < SL_SLICES; g0 += TN * SL_AHEAD) {        /* rewrite 1, selection (:2134-2167): of a chain of candidates four apart every other one, from the chain's head */
		uint64_t Mk[SL_AHEAD]; uint4 vk[SL_AHEAD];
		SL_FETCH(Mk, vk, g0);
#pragma unroll
		for (int k = 0; k < SL_AHEAD; k++) {
		const int g = g0 + TN * k;
		const uint64_t M = Mk[k];
		if (!M) continue;
		uint64_t p6, p0;
		sl_pm8(L, g, M, vk[k], &p6, &p0);
		const uint64_t P = p6 | p0;
		if (!P) continue;
		const uint64_t Z = ~M, Zn = ~sl_word(L, g + 1);
		uint64_t own = P & ((Z >> 1) | (Zn << 63)) & ((Z >> 2) | (Zn << 62)) & ((Z >> 3) | (Zn << 61));
		uint64_t cand = own & (P >> 4);
		for (uint64_t h = own >> 60 << 60; h; h &= h - 1) {           /* the second +-8 lies in the next slice */
			const int j = __builtin_ctzll(h), c4 = 64 * g + j + 4;
			if (c4 < n && is_pm8(sl_sym(L, c4))) cand |= 1ull << j;
		}
		for (uint64_t h = cand; h; h &= h - 1) {
			const int j = __builtin_ctzll(h), cpos = 64 * g + j;
			int m = 1, back = cpos - 4;
			while (back >= 64 * g ? (int)((cand >> (back & 63)) & 1) : (int)sl_cand(L, back)) { m++; back -= 4; }
			if (m & 1) sel[atomicAdd(&sh_counts[0], 1)] = (uint32_t)cpos;
		}
		}
	}
	BARRIER();
	if (!tid) PROF(c, 41);
	for (int e = tid; e < sh_counts[0]; e += TN) {                  /* rewrite 1, application */
		const int cpos = (int)sel[e];
		const int x = sl_sym(L, cpos), y = sl_sym(L, cpos + 4);
		sl_set(L, cpos, x == 136 ? (y == 136 ? 132 : 133) : (y == 136 ? 134 : 135));
		sl_set(L, cpos + 4, 201);
		if ((cpos & 63) >= 60) L.vo[(cpos >> 6) + 1] |= (uint32_t)((cpos & 63) - 59) << 29;   /* the packetiser's walk of the next slice starts behind the code's five symbols (selected pairs lie eight apart: nobody else writes this word now) */
	}
	BARRIER();
	if (tid == 0) {                                                 /* the first and the last four symbols become zero symbols (:2169-2176): bits off; the first slice's values start behind the dead ones */
		const uint64_t m0 = L.nz[0];
		L.vo[0] += (unsigned)__builtin_popcountll(m0 & 0xFull);
		L.nz[0] = m0 & ~0xFull;
		L.nz[SL_SLICES - 1] &= ~(0xFull << 60);
	}
	BARRIER();
	if (!tid) PROF(c, 42);
	for (int g0 = tid; g0 < SL_SLICES; g0 += TN * SL_AHEAD) {        /* rewrite 2 (:2178-2220): every decision reads what no decision writes (scan_and_rewrite_par has the argument) */
		uint64_t Mk[SL_AHEAD]; uint4 vk[SL_AHEAD];
		SL_FETCH(Mk, vk, g0);
#pragma unroll
		for (int k = 0; k < SL_AHEAD; k++) {
		const int g = g0 + TN * k;
		const uint64_t M = Mk[k];
		if (!M) continue;
		uint64_t p6, p0;
		sl_pm8(L, g, M, vk[k], &p6, &p0);
		const uint64_t P = p6 | p0;
		uint64_t act = P;
		if (g == 0) act &= ~0xFull;                                /* 4 <= i < n - 4 */
		if (g == SL_SLICES - 1) act &= ~(0xFull << 60);
		if (!act) continue;
		const int base = 64 * g;
		const uint64_t Z = ~M, Zp = ~sl_word(L, g - 1), Zn = ~sl_word(L, g + 1);
#define ZL(k) ((Z << (k)) | (Zp >> (64 - (k))))                     /* bit j: position base + j - k is the zero symbol */
#define ZR(k) ((Z >> (k)) | (Zn << (64 - (k))))
		int v_next = 0;                                             /* the symbol behind the slice, where a pair rule asks for it */
		uint64_t Pl = P << 1, Pr = P >> 1;                          /* the left / right neighbour is a +-8 */
		if ((act & 1) && g > 0 && !((Zp >> 63) & 1)) Pl |= is_pm8(sl_sym(L, base - 1)) ? 1ull : 0ull;
		if ((act >> 63) && g < SL_SLICES - 1 && !(Zn & 1)) { v_next = sl_sym(L, base + 64); if (is_pm8(v_next)) Pr |= 1ull << 63; }
		if (g == 0) Pl &= ~0x1Full;                                /* the left neighbour is asked only for i > 4 */
		const uint64_t before4 = ZL(1) & ZL(2) & ZL(3) & ZL(4), after3 = ZR(2) & ZR(3) & ZR(4);
		const uint64_t taken = Pl & ZR(1) & ((ZL(2) & ZL(3) & ZL(4) & ZL(5)) | (ZL(2) & after3));   /* the left neighbour took me as the second of a pair */
		act &= ~taken;
		const uint64_t pairA = act & Pr & ZR(2) & (before4 | (ZL(1) & ZR(3) & ZR(4) & ZR(5)));
		const uint64_t lone = act & ~pairA & ZR(1) & (before4 | (ZL(1) & after3));
#undef ZL
#undef ZR
		for (uint64_t h = pairA; h; h &= h - 1) {
			const int j = __builtin_ctzll(h);
			const bool neg = j < 63 ? (bool)((p0 >> (j + 1)) & 1) : v_next == 120;
			sl_set(L, base + j + 1, neg ? 157 : 159);
		}
		for (uint64_t h = lone; h; h &= h - 1) {
			const int j = __builtin_ctzll(h);
			sl_set(L, base + j, ((p6 >> j) & 1) ? 153 : 155);
		}
		}
	}
	BARRIER();
	if (!tid) PROF(c, 43);
	/* rewrite 3 (:2222-2252): the sign codes behind a zero run of 252 or more; a run belongs to the slice its successor is in.  Where the run
	 * starts = the last non-zero symbol before: a thread takes consecutive slices and knows it for them from a prefix maximum over the
	 * threads (walking back over the empty map words took a thread thousands of steps at the low qualities, whose streams are mostly
	 * empty: 0.2 of Y31's 0.21 ms an image at q10) */
	constexpr int SPT3 = SL_SLICES / TN;
	int last_nz = -1;
	for (int k = 0; k < SPT3; k++) { const uint64_t M = L.nz[tid * SPT3 + k]; if (M) last_nz = 64 * (tid * SPT3 + k) + 63 - __builtin_clzll(M); }
	int prev_nz = (int)block_exscan_max((unsigned)(last_nz + 1), tid, reinterpret_cast<unsigned *>(lds + SL_SLICES * 12)) - 1;   /* before my first slice; -1: none */
	for (int k = 0; k < SPT3; k++) {
		const int g = tid * SPT3 + k;
		const uint64_t M = L.nz[g];
		if (!M) continue;
		const int p = 64 * g + __builtin_ctzll(M);
		const int zeros = p - prev_nz - 1;
		prev_nz = 64 * g + 63 - __builtin_clzll(M);
		if (zeros < 252) continue;
		const int i = p - zeros, b = p - 1;                         /* the run [i, b] */
		auto fix = [&](int at) { const int v = sl_sym(L, at); if (v == 153) sl_set(L, at, 124); else if (v == 155) sl_set(L, at, 123); };
		const int fired = b - i >= 256 ? (b - i - 256) / 254 + 1 : 0;
		const int kk = i + 255 + 254 * (fired - 1);
		if (fired) for (int at = b + 1; at <= kk + 3; at++) if (at < n) fix(at);
		const int tail_run = fired ? b - kk + 1 : b - i;
		if (tail_run >= 252 && b + 1 < n) fix(b + 1);
	}
	BARRIER();
	for (int g = tid; g < SL_SLICES; g += TN) { c->nzs[g] = L.nz[g]; c->voff[g] = L.vo[g]; }
	if (!tid) PROF(c, 44);
}


/* offsetUV_recons256 (image_processing.c:3192-3353): p is only read, every row writes its own jp cells */
DEV void dequant_sim_chroma_par(Ctx *c, int comp, int tid)
{
	int16_t *p = c->cproc, *jp = c->cjpeg;
	for (int idx = tid; idx < (H / 4) * (H / 4); idx += NT) {
		const int r = idx >> 6, j = idx & 63, i = r * H + j;
		if (comp && c->q <= 15) jp[i] = (int16_t)((p[i] & 0xFFFC) + 1);   /* :3221-3230: two low bits dropped, midpoint */
		else if (comp) {
			if (j & 1) continue;
			if (r == 0) { jp[i] = p[i]; jp[i + 1] = clear_bit0(p[i + 1]); }
			else { jp[i] = clear_bit0(p[i]); jp[i + 1] = p[i + 1]; }
		} else jp[i] = (p[i] > 0 && p[i] < 256) ? clear_bit0(p[i]) : p[i];
	}
	/* detail rows: one wavefront per row, lane l owns columns l and l + 64; the -7/-8 pairs of the second loop are a
	 * walk that skips the partner (alt_runs on "pair starts here"), everything else is a stencil on the untouched plane */
	const int lane = tid & 63, wv = tid >> 6;
	int nv[3];                                                      /* the next row's cells, requested while this row is worked on (rows are independent) */
	nv[0] = wv < H / 4 ? 0 : p[wv * H + lane]; nv[1] = p[wv * H + 64 + lane]; nv[2] = p[wv * H + 128 + lane];
	for (int r = wv; r < H / 2; r += 4) {
		const int col0 = r < H / 4 ? H / 4 : 0;
		int v[3] = { nv[0], nv[1], nv[2] };                          /* the last cell looks at column 128 */
		if (r + 4 < H / 2) { const int rn = r + 4; nv[0] = rn < H / 4 ? 0 : p[rn * H + lane]; nv[1] = p[rn * H + 64 + lane]; nv[2] = p[rn * H + 128 + lane]; }
		uint64_t pair[2] = { 0, 0 };
		if (!comp) {
			const uint64_t m0 = __ballot(v[0] == -7 || v[0] == -8), m1 = __ballot(v[1] == -7 || v[1] == -8);
			const M4 m = M4{ { col0 ? 0 : m0, m1, 0, 0 } };
			const M4 fired = alt_runs(m & dn1(m) & col_range(col0, H / 2 - 2));
			const M4 both = fired | up1(fired);
			pair[0] = both.w[0]; pair[1] = both.w[1];
		}
		for (int k = col0 ? 1 : 0; k < 2; k++) {
			int a = v[k];
			const int nx = right_of(v, k, 3, 1, lane);
			if (a < 0) {
				a = -a;
				if (nx < 0 && nx > -8) { if ((a & 7) < 6) a &= 0xFFF8; }
				else { if ((a & 7) < 7) a &= 0xFFF8; }
				a = -a;
			}
			jp[r * H + lane + 64 * k] = (int16_t)(((pair[k] >> lane) & 1) ? -11 : dequant_value(a));
		}
	}
}


/* Y25 (nhw_encoder.c:1498-1887): the three compaction sweeps over the code plane run one row per thread (count,
 * prefix, write); packing the (short) lists stays on thread 0 */
DEV unsigned block_exscan(unsigned v, int tid, unsigned *shm, unsigned *total);
DEV unsigned block_exscan_max(unsigned v, int tid, unsigned *shm);

/* The position lists' packing (nhw_encoder.c:1546-1631, 1751-1763), workgroup-parallel.  Every step of the
 * reference's walk over the list is a filter or a skip-the-partner walk on values the step does not change:
 *   prune  -- drop a row marker whose neighbours say the column index fell across it: a pure stencil on the raw list;
 *   fuse   -- two small steps in a row share a byte and the walk jumps over the second: "fires if visited" is a
 *             stencil on the halved list, the visited cells follow from the parity inside each run of fire bits;
 *   bits   -- the low bits of the non-marker entries, 8 per byte: a compaction;
 *   words  -- the payload symbols, 8 (or 4) per byte.
 * Each thread owns a contiguous chunk of the list; counts go through workgroup prefix sums. */
DEV void poslist_finish_par(Ctx *c, PosList *pl, const uint8_t *__restrict__ raw, int n, const uint8_t *__restrict__ payload, int payload_len, int word_mode, int tid, unsigned *shm)
{
	/* (distinct buffers, and told so: a compaction loop "P[at++] = raw[i]" is otherwise a memory round trip a turn, the store may alias the next load) */
	uint8_t *__restrict__ P = c->cc, *__restrict__ F = c->half;   /* pruned list; per-entry flags / compacted low bits */
	uint8_t *__restrict__ out_list = pl->list, *__restrict__ out_bits = pl->bits, *__restrict__ out_word = pl->word;
	unsigned total;
	{                                                            /* prune (:1546-1561) */
		const int L = (n + NT - 1) / NT, i0 = tid * L, i1 = i0 + L < n ? i0 + L : n;
		unsigned cnt = 0;
		for (int i = i0; i < i1; i++) {
			const bool drop = i >= 1 && i < n - 1 && raw[i] == H - 2 && raw[i - 1] != H - 2 && raw[i + 1] != H - 2 && raw[i - 1] > raw[i + 1];
			cnt += !drop;
		}
		unsigned at = block_exscan(cnt, tid, shm, &total);
		for (int i = i0; i < i1; i++) {
			const bool drop = i >= 1 && i < n - 1 && raw[i] == H - 2 && raw[i - 1] != H - 2 && raw[i + 1] != H - 2 && raw[i - 1] > raw[i + 1];
			if (!drop) P[at++] = raw[i];
		}
	}
	const int m = (int)total;
	BARRIER();
	const int L = (m + NT - 1) / NT, i0 = tid * L, i1 = i0 + L < m ? i0 + L : m;
#define PL_FIRE(i) ((i) >= 1 && (i) <= m - 2 && (unsigned)((P[i] >> 1) - (P[(i) - 1] >> 1)) < 8u && (unsigned)((P[(i) + 1] >> 1) - (P[i] >> 1)) < 16u)
	{                                                            /* fuse (:1569-1592): which entries the walk fuses with their successor */
		int lastnf = 0;                                          /* index + 1 of my last entry that cannot fire */
		for (int i = i0; i < i1; i++) if (!PL_FIRE(i)) lastnf = i + 1;
		int rs = (int)block_exscan_max((unsigned)lastnf, tid, shm);   /* first entry of the run of fire bits reaching into my chunk */
		for (int i = i0; i < i1; i++) {
			const bool f = PL_FIRE(i);
			F[i] = f && !((i - rs) & 1);
			if (!f) rs = i + 1;
		}
	}
	BARRIER();
	{
		unsigned cnt = 0;
		for (int i = i0; i < i1; i++) cnt += i >= 1 && i <= m - 2 && !F[i - 1];
		unsigned at = 1 + block_exscan(cnt, tid, shm, &total);
		for (int i = i0; i < i1; i++) {
			if (!(i >= 1 && i <= m - 2 && !F[i - 1])) continue;
			const int h = P[i] >> 1;
			out_list[at++] = (uint8_t)(F[i] ? 128 + ((h - (P[i - 1] >> 1)) << 4) + ((P[i + 1] >> 1) - h) : h);
		}
		if (tid == 0) { out_list[0] = P[0] >> 1; pl->len->list_len = 1 + (int)total; }
	}
#undef PL_FIRE
	BARRIER();
	{                                                            /* plane of the dropped low bits, markers excluded (:1594-1615) */
		unsigned cnt = 0;
		for (int i = i0; i < i1; i++) cnt += P[i] != H - 2;
		unsigned at = block_exscan(cnt, tid, shm, &total);
		for (int i = i0; i < i1; i++) if (P[i] != H - 2) F[at++] = P[i] & 1;
		BARRIER();
		const int nb = (int)total, groups = (nb >> 3) + 1;
		for (int g = tid; g < groups; g += NT) {
			int v = 0;
			for (int b = 0; b < 8; b++) v = (v << 1) | (8 * g + b < nb ? F[8 * g + b] : 0);
			out_bits[g] = (uint8_t)v;
		}
		if (tid == 0) pl->len->bits_len = groups;
	}
	{                                                            /* payload symbols (:1620-1631, 1751-1763); symbols behind payload_len read as 0 */
		const int groups = (payload_len >> 3) + 1;
		for (int g = tid; g < groups; g += NT) {
			int sym[8];
			for (int b = 0; b < 8; b++) sym[b] = (8 * g + b < payload_len) ? payload[8 * g + b] : 0;
			if (word_mode == 2) {
				out_word[2 * g] = (uint8_t)(((sym[0] & 3) << 6) | ((sym[1] & 3) << 4) | ((sym[2] & 3) << 2) | (sym[3] & 3));
				out_word[2 * g + 1] = (uint8_t)(((sym[4] & 3) << 6) | ((sym[5] & 3) << 4) | ((sym[6] & 3) << 2) | (sym[7] & 3));
			} else {
				int v = 0;
				for (int b = 0; b < 8; b++) v = (v << 1) | (sym[b] & 1);
				out_word[g] = (uint8_t)v;
			}
		}
		if (tid == 0) pl->len->word_len = word_mode == 2 ? 2 * groups : groups;
	}
	BARRIER();
}

/* Y25 (nhw_encoder.c:1498-1763): three filters over the LL1 tag plane, each collecting (column, payload) of its
 * codes row by row behind a row marker, and leaving a follow-up code for the next filter in some cells.  A cell's
 * fate through the three filters depends on the cell alone, so one sweep evaluates all three: a wavefront per
 * row (lane l owns columns l + 64k), match masks by ballot, the entries written in the same sweep (see below). */
DEV int poslist_match(int pass, int v, int *payload, int *keep)
{
	*keep = 0;
	if (pass == 0) {
		if (v == 141) { *payload = 1; return 1; } if (v == 140) { *payload = 0; return 1; }
		if (v == 126) { *payload = 0; *keep = 122; return 1; } if (v == 125) { *payload = 1; *keep = 121; return 1; }
		if (v == 148) { *payload = 1; *keep = 144; return 1; } if (v == 149) { *payload = 0; *keep = 145; return 1; }
	} else if (pass == 1) {
		if (v >= 121 && v <= 124) { *payload = v == 121 ? 1 : v == 122 ? 0 : v == 123 ? 2 : 3; return 1; }
	} else { if (v == 144) { *payload = 1; return 1; } if (v == 145) { *payload = 0; return 1; } }
	return 0;
}
DEV void build_poslists_par(Ctx *c, int tid, int *pos, int16_t *lds)
{
	PROF_BEGIN();
	int16_t *o = c->ll1;
	const int q = c->q, lane = tid & 63, wv = tid >> 6;
	const int npass = q >= 21 ? 3 : (q >= 19 ? 2 : 1);
	uint8_t *raw[3] = { c->raw, c->raw + Q + 512, reinterpret_cast<uint8_t *>(c->hs) };
	uint8_t *pay[3] = { c->pay, c->pay + Q, reinterpret_cast<uint8_t *>(c->hs) + Q + 512 };
	/* One sweep.  A wavefront takes 64 consecutive rows and writes their entries, row after row, where its rows' entries would start if every
	 * row above were full (64 x 255 raw bytes, 64 x 254 payload bytes a wavefront: the lists' buffers hold four such segments); then the
	 * three later segments move left to where they belong.  (Until round 4: a sweep to count the rows' matches, prefix sums over the rows,
	 * a second sweep -- every row loaded and matched twice -- to write.) */
	const int RSEG = 64 * (H - 1), PSEG = 64 * (H - 2);
	int *segn = reinterpret_cast<int *>(lds);                     /* [3][4][2]: a wavefront's raw bytes / payload bytes of a pass */
	unsigned *shm = reinterpret_cast<unsigned *>(lds) + 3 * H;
	unsigned total[3] = { 0, 0, 0 };
	{
		int rl[3] = { 0, 0, 0 }, pln[3] = { 0, 0, 0 };
		uint8_t *rseg[3], *pseg[3];
		for (int pass = 0; pass < 3; pass++) { rseg[pass] = raw[pass] + wv * RSEG; pseg[pass] = pay[pass] + wv * PSEG; }
		const int rbeg = 64 * wv;
		int nv[4];                                                  /* the next row, requested while the row is worked on (a row only rewrites itself) */
#pragma unroll
		for (int k = 0; k < 4; k++) nv[k] = o[rbeg * H + lane + 64 * k];
		for (int r = rbeg; r < rbeg + 64; r++) {
			int v[4];
#pragma unroll
			for (int k = 0; k < 4; k++) v[k] = nv[k];
			if (r + 1 < rbeg + 64) {
#pragma unroll
				for (int k = 0; k < 4; k++) nv[k] = o[(r + 1) * H + lane + 64 * k];
			}
			if (lane >= 62) v[3] = 0;                               /* columns 254, 255 take no part and are cleared */
			for (int pass = 0; pass < npass; pass++) {
				/* which codes a pass takes, as a bit a code from 120 on (poslist_match): the match is two instructions, what a match carries is
				 * only worked out in the lanes that have one (the pass is bound by its instruction count: 300 a row before this) */
				const uint32_t takes = pass == 0 ? (1u << 21 | 1u << 20 | 1u << 6 | 1u << 5 | 1u << 28 | 1u << 29) : pass == 1 ? (15u << 1) : (3u << 24);
				int pl[4] = { 0, 0, 0, 0 }, kp[4] = { 0, 0, 0, 0 };
				uint64_t m[4];
				for (int k = 0; k < 4; k++) {
					const unsigned u = (unsigned)(v[k] - 120);
					const bool hit = u < 30u && ((takes >> u) & 1u);
					m[k] = __ballot(hit);
					if (hit) poslist_match(pass, v[k], &pl[k], &kp[k]);
				}
				const int n = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
				int before = 0;
				for (int k = 0; k < 4; k++) {
					if ((m[k] >> lane) & 1) {
						const int at = before + __popcll(m[k] & low_bits(lane));
						rseg[pass][rl[pass] + at] = (uint8_t)(lane + 64 * k); pseg[pass][pln[pass] + at] = (uint8_t)pl[k];
					}
					before += __popcll(m[k]);
				}
				if (lane == 0) rseg[pass][rl[pass] + n] = H - 2;      /* the row's marker behind its entries */
				rl[pass] += n + 1; pln[pass] += n;
				for (int k = 0; k < 4; k++) if ((m[k] >> lane) & 1) v[k] = kp[k];
			}
			/* (the reference leaves the follow-up codes and the cleared columns 254, 255 in the plane: nothing reads it behind this pass, so it is not written back) */
		}
		if (lane == 0) for (int pass = 0; pass < 3; pass++) { segn[(pass * 4 + wv) * 2] = rl[pass]; segn[(pass * 4 + wv) * 2 + 1] = pln[pass]; }
	}
	BARRIER();
	for (int pass = 0; pass < npass; pass++) {
		int dr = segn[(pass * 4) * 2], dp = segn[(pass * 4) * 2 + 1];   /* where the next segment belongs */
		for (int w = 1; w < 4; w++) {
			const int lr = segn[(pass * 4 + w) * 2], lp = segn[(pass * 4 + w) * 2 + 1];
			for (int which = 0; which < 2; which++) {
				uint8_t *buf = which ? pay[pass] : raw[pass];
				const int src = w * (which ? PSEG : RSEG), dst = which ? dp : dr, len = which ? lp : lr;
				if (src == dst) continue;
				/* left moves in pieces of 1 KB, a piece read by everybody before anybody writes it: a piece's place ends in front of the next piece's source */
				for (int c0 = 0; c0 < len; c0 += 4 * NT) {
					uint8_t b4[4];
#pragma unroll
					for (int u = 0; u < 4; u++) { const int i = c0 + tid + u * NT; b4[u] = i < len ? buf[src + i] : (uint8_t)0; }
					BARRIER();
#pragma unroll
					for (int u = 0; u < 4; u++) { const int i = c0 + tid + u * NT; if (i < len) buf[dst + i] = b4[u]; }
				}
			}
			dr += lr; dp += lp;
		}
		total[pass] = (unsigned)dp;
		BARRIER();
	}
	if (!tid) PROF(c, 45);
	for (int pass = 0; pass < npass; pass++) {
		poslist_finish_par(c, pass == 0 ? &c->res1 : pass == 1 ? &c->res3 : &c->res5, raw[pass], (int)total[pass] + H, pay[pass], (int)total[pass], pass == 1 ? 2 : 1, tid, shm);
		if (!tid) PROF(c, 46 + pass);
	}
}

/* ---------------------------------------------------------------- phases (256 threads per image) */
/* ---------------------------------------------------------------- Y16: the LL2 byte coder, workgroup-parallel */
/* exclusive prefix sum of one 32-bit value per thread over the workgroup (wave shuffles + one LDS hop) */
DEV unsigned block_exscan(unsigned v, int tid, unsigned *shm /* [NT / 64 + 1] */, unsigned *total)
{
	const int lane = tid & 63, wv = tid >> 6;
	unsigned x = v;
	for (int d = 1; d < 64; d <<= 1) { const unsigned y = __shfl_up(x, d); if (lane >= d) x += y; }
	BARRIER();
	if (lane == 63) shm[wv] = x;
	BARRIER();
	unsigned base = 0, sum = 0;
	for (int k = 0; k < NT / 64; k++) { if (k < wv) base += shm[k]; sum += shm[k]; }
	*total = sum;
	return base + x - v;
}

/* exclusive prefix maximum (0 for the first thread) */
DEV unsigned block_exscan_max(unsigned v, int tid, unsigned *shm /* [NT / 64 + 1] */)
{
	const int lane = tid & 63, wv = tid >> 6;
	unsigned x = v;
	for (int d = 1; d < 64; d <<= 1) { const unsigned y = __shfl_up(x, d); if (lane >= d && y > x) x = y; }
	BARRIER();
	if (lane == 63) shm[wv] = x;
	BARRIER();
	unsigned base = 0;
	for (int k = 0; k < wv; k++) base = shm[k] > base ? shm[k] : base;
	const unsigned prev = __shfl_up(x, 1);
	const unsigned mine = lane ? prev : 0;
	return mine > base ? mine : base;
}

/* one token of Y_highres_compression (compress_pixel.c:510-790) as if the walk stood at sample i: returns the
 * sample the walk visits next; OUT: the token's bytes after the marker strip of :828-866 (a single stays, a
 * (64, x, y) triple keeps x y, a (128, a, b) verbatim record keeps b), and whether it is a verbatim record */
template <bool OUT>
DEV int ll_luma_token(const uint8_t *s, int i, int n, int mode, int *nb, int *b0, int *b1, int *verb, bool low /* q <= 15: an escape is (128, halved sample i), nothing verbatim, and the walk moves one sample (:573-577, :857-861) */)
{
	const int d0 = s[i] - s[i - 1], d1 = s[i + 1] - s[i];
	int kind = 2, byte = 0, t0 = 0, t1 = 0, t2 = 0, next;           /* kind 0: one byte, 1: triple, 2: verbatim */
	if (d0 == 0 && d1 == 0) {
		int a = 0, ii, d;
		if (mode == 0) {                                            /* :515-553 */
			if (s[i + 2] == s[i + 1]) a = 1;
			ii = i + a + 2; byte = a << 3;
			d = s[ii] - s[ii - 1];
			if (d == 2) { const int f = s[ii + 1] - s[ii]; if (f == -2) { byte += 2; ii++; } else if (f == 0) { byte += 3; ii++; } else byte += 1; }
			else if (d == -2) { const int f = s[ii + 1] - s[ii]; if (f == 2) { byte += 4; ii++; } else if (f == 0) { byte += 5; ii++; } else byte += 6; }
			else if (d == 4) byte += 7;
			else ii--;
		} else if (mode == 1) {                                     /* :652-673 */
			while (a < 7 && s[i + a + 2] == s[i + a + 1]) a++;
			ii = i + a + 2; byte = a << 2;
			d = s[ii] - s[ii - 1];
			if (d == 2) byte += 1; else if (d == -2) byte += 2; else if (d == 0) byte += 3; else ii--;
		} else {                                                    /* :762-775 */
			while (a < 63 && s[i + a + 2] == s[i + a + 1]) a++;
			ii = i + a + 1; byte = a;
		}
		kind = 0; next = ii + 1;
	} else {
		const int d2 = s[i + 2] - s[i + 1];
		const bool d2ok = iabs(d2) <= 32 && i < n - 2;
		if (mode == 0 && iabs(d0) <= 6 && iabs(d1) <= 8) {          /* :554-599 */
			const int e0 = d0 + 6, e1 = d1 + 8;
			if (e0 == 12 || e1 == 16) { if (d2ok) { kind = 1; t0 = e0 + 26; t1 = e1 + 8; t2 = d2 + 32; } }
			else { kind = 0; byte = e0 < 8 ? 32 + (e0 << 2) + (e1 >> 1) : (e0 == 8 ? 16 + (e1 >> 1) : 24 + (e1 >> 1)); }
		}
		else if (mode == 1 && iabs(d0) <= 4 && iabs(d1) <= 8) {     /* :674-706 */
			const int e0 = d0 + 4, e1 = d1 + 8;
			if (e0 == 8 || e1 == 16) { if (d2ok) { kind = 1; t0 = e0 + 28; t1 = e1 + 8; t2 = d2 + 32; } }
			else { kind = 0; byte = 32 + (e0 << 2) + (e1 >> 1); }
		}
		else if (iabs(d0) <= 32 && iabs(d1) <= 16 && d2ok) { kind = 1; t0 = d0 + 32; t1 = d1 + 16; t2 = d2 + 32; }   /* :600-630 */
		if (kind == 1 && (t0 == 64 || t1 == 32 || t2 == 64)) kind = 2;
		next = kind == 1 ? i + 3 : (kind == 2 && low ? i + 1 : i + 2);
	}
	if (OUT) {
		*verb = kind == 2 && !low;
		if (kind == 0) { *nb = 1; *b0 = byte; }
		else if (kind == 1) { t1 >>= 1; *nb = 2; *b0 = 64 + t0 + (t1 >> 3); *b1 = ((t1 & 7) << 5) + (t2 >> 1); }
		else { *nb = 1; *b0 = 128 + (s[low ? i : i + 1] >> 1); }
	}
	return next;
}

/* The coder is a parse: the token at sample i decides which sample is looked at next (1 to 66 further).  What a
 * token would be at i depends only on the samples, so every thread first works out the stride at its samples;
 * which samples the walk really visits is then a pointer chase, done in three short hops instead of one long
 * one: (1) per 64-sample block, backwards: from each sample, where does the walk leave the block; (2) one thread
 * hops block to block (<= 256 hops) and notes where each block is entered; (3) every thread re-walks its block
 * from its entry, twice: to count its output bytes / verbatim records, and, after a prefix sum, to write them.
 * LDS: the samples (+ zero padding the reference also reads) and one byte per sample for strides / exits, the
 * latter padded 4 bytes per block so that one block per lane is bank-conflict free. */
#define LLX(i) ((i) + (((i) >> 6) << 2))
#define LL_LDS_BYTES (16640 + 16384 + 1024 + 512 + 64)
DEV void ll_code_luma_par(Ctx *c, int tid, uint8_t *lds /* LL_LDS_BYTES, the samples already in the first 16640 */)
{
	const int n = Q >> 2, lane = tid & 63, wv = tid >> 6;
	const uint8_t *s = lds;
	uint8_t *X = lds + 16640;
	int16_t *entry = reinterpret_cast<int16_t *>(lds + 16640 + 16384 + 1024);
	unsigned *shm = reinterpret_cast<unsigned *>(lds + 16640 + 16384 + 1024 + 512);   /* [16] */

	/* statistics (compress_pixel.c:482-497): in every run of equal neighbours, cut into pieces of 16 matches, count the
	 * pieces that reach 8 and 16 matches.  A piece must start before n; its matches may lie in the padding. */
	const int words = (n + 16 + 63) / 64, wpw = (words + 3) / 4;
	int last_mis = -1;                                           /* last sample of this wave's range that differs from its left neighbour */
	for (int k = 0; k < wpw; k++) {
		const int i = (wv * wpw + k) * 64 + lane;
		const bool mis = i == 0 || (i < n + 16 && s[i] != s[i - 1]);
		const uint64_t mm = __ballot(mis);
		if (mm) last_mis = (wv * wpw + k) * 64 + 63 - __builtin_clzll(mm);
	}
	if (lane == 0) reinterpret_cast<int *>(shm)[wv] = last_mis;
	BARRIER();
	int carry = 0;
	for (int k = 0; k < wv; k++) { const int v = reinterpret_cast<int *>(shm)[k]; carry = v > carry ? v : carry; }
	int r8 = 0, r16 = 0;
	for (int k = 0; k < wpw; k++) {
		const int base = (wv * wpw + k) * 64, i = base + lane;
		const bool in = i >= 1 && i < n + 16;
		const bool mis = i == 0 || (i < n + 16 && s[i] != s[i - 1]);
		const uint64_t mm = __ballot(mis);
		const uint64_t below = mm & ((2ull << lane) - 1);
		const int lm = below ? base + 63 - __builtin_clzll(below) : carry;
		const int off = i - lm - 1;                              /* this sample is match number off + 1 of its run */
		r8 += __popcll(__ballot(in && !mis && (off & 15) == 7 && i - 7 < n));
		r16 += __popcll(__ballot(in && !mis && (off & 15) == 15 && i - 15 < n));
		if (mm) carry = base + 63 - __builtin_clzll(mm);
	}
	BARRIER();
	if (lane == 0) { shm[wv] = (unsigned)r8; shm[4 + wv] = (unsigned)r16; }
	BARRIER();
	const int runs16 = (int)(shm[4] + shm[5] + shm[6] + shm[7]), runs8 = (int)(shm[0] + shm[1] + shm[2] + shm[3]) + runs16;
	const int mode = runs16 > 299 ? 2 : (runs8 > 179 ? 1 : 0);       /* :506-508 */
	BARRIER();

	const bool low = c->q <= 15;
	for (int i = tid; i < n; i += NT) X[LLX(i)] = i ? (uint8_t)(ll_luma_token<false>(s, i, n, mode, nullptr, nullptr, nullptr, nullptr, low) - i) : 1;
	entry[tid] = -1;
	BARRIER();
	{                                                            /* (1) where the walk leaves my block, from each of its samples */
		const int b0 = tid * 64, end = b0 + 64;
		for (int e = end - 1; e >= b0; e--) {
			const int nx = e + X[LLX(e)];
			X[LLX(e)] = (uint8_t)(nx >= end ? nx - end : X[LLX(nx)]);
		}
	}
	BARRIER();
	if (tid == 0) { int pos = 1; while (pos < n) { const int b = pos >> 6; entry[b] = (int16_t)pos; pos = (b + 1) * 64 + X[LLX(pos)]; } }   /* (2) */
	BARRIER();
	const int e0 = entry[tid], end = tid * 64 + 64;
	unsigned cnt = 0;
	if (e0 >= 0)
		for (int i = e0; i < end;) {
			int nb, b0, b1, verb;
			i = ll_luma_token<true>(s, i, n, mode, &nb, &b0, &b1, &verb, low);
			cnt += (unsigned)nb + ((unsigned)verb << 16);
		}
	unsigned total;
	const unsigned off = block_exscan(cnt, tid, shm, &total);
	if (e0 >= 0) {
		uint8_t *o = c->ll_comp + 1 + (off & 0xFFFF);
		int m = (int)(off >> 16);
		for (int i = e0; i < end;) {
			int nb, b0, b1, verb;
			const int at = i;
			i = ll_luma_token<true>(s, i, n, mode, &nb, &b0, &b1, &verb, low);
			*o++ = (uint8_t)b0;
			if (nb == 2) *o++ = (uint8_t)b1;
			if (verb) { c->ll_word[m] = c->ll_full[at]; c->ll_mem[m] = (uint16_t)at; m++; }
		}
	}
	if (tid == 0) {
		c->ll_comp[0] = s[0];
		c->m->res_low = mode;
		c->m->ll_comp_y_len = 1 + (int)(total & 0xFFFF);
		c->m->ll_word_len = (int)(total >> 16);
		c->m->ll_mem_len = (int)(total >> 16);
	}
}

/* ---------------------------------------------------------------- Z1: the chroma LL2 byte coder, same scheme */
/* one token of highres_compression (compress_pixel.c:890-1010) at sample i: always one byte; returns the next sample */
DEV int ll_chroma_token(const uint8_t *s, int i, int *byte)
{
	const int d0 = s[i] - s[i - 1], d1 = s[i + 1] - s[i];
	if (d0 == 0 && d1 == 0) {                                       /* :898-945 run of equal samples, up to 14 */
		int a = 0;
		while (a < 14 && s[i + a + 2] == s[i + a + 1]) a++;
		int ii = i + a + 1;
		if (a >= 7) { *byte = 64 + (7 << 3) + a - 7; return ii + 1; }
		ii++;
		int b = 64 + (a << 3);
		const int d = s[ii] - s[ii - 1];
		if (d == 4) {
			if (s[ii + 1] - s[ii] == -4) { if (s[ii + 2] - s[ii + 1] == 0) { b += 3; ii += 2; } else { b += 2; ii++; } }
			else b += 1;
		} else if (d == -4) {
			if (s[ii + 1] - s[ii] == 4) { if (s[ii + 2] - s[ii + 1] == 0) { b += 4; ii += 2; } else { b += 5; ii++; } }
			else b += 6;
		} else if (d == 8) b += 7;
		else ii--;
		*byte = b;
		return ii + 1;
	}
	if (iabs(d0) <= 4 && iabs(d1) <= 4) {                           /* :946-984 steps of 0/+-4 */
		int code = 0;
		if (!d0 && d1 == 4) code = 0; else if (!d0 && d1 == -4) code = 1;
		else if (d0 == 4 && !d1) code = 2; else if (d0 == -4 && !d1) code = 3;
		else if (d0 == 4 && d1 == 4) code = 4; else if (d0 == 4 && d1 == -4) code = 5;
		else if (d0 == -4 && d1 == 4) code = 6; else if (d0 == -4 && d1 == -4) code = 7;
		const int d2 = s[i + 2] - s[i + 1];
		if (d2 == 0) { *byte = 128 + 64 + (code << 2); return i + 3; }
		if (d2 == 4) { *byte = 128 + 64 + (code << 2) + 1; return i + 3; }
		if (d2 == -4) { *byte = 128 + 64 + (code << 2) + 2; return i + 3; }
		if (d2 == 8) { *byte = 128 + 64 + (code << 2) + 3; return i + 3; }
		*byte = ((d0 + 16) << 1) + ((d1 + 16) >> 2);
		return i + 2;
	}
	if (iabs(d0) <= 16 && iabs(d1) <= 16) {                         /* :985-1003 */
		const int e0 = d0 + 16, e1 = d1 + 16;
		if (e0 == 32 || e1 == 32) { *byte = 128 + (s[i] >> 2); return i + 1; }
		*byte = (e0 << 1) + (e1 >> 2);
		return i + 2;
	}
	*byte = 128 + (s[i] >> 2);                                      /* :1004-1010 */
	return i + 1;
}

#define LCX(i) ((i) + (((i) >> 5) << 2))
#define LLC_LDS_BYTES (8192 + 128 + 8192 + 1024 + 512 + 64)
DEV void ll_code_chroma_par(Ctx *c, int tid, uint8_t *lds /* LLC_LDS_BYTES */)
{
	const int n = Q >> 3, lo = Q >> 2;
	uint8_t *s = lds, *X = lds + 8192 + 128;
	int16_t *entry = reinterpret_cast<int16_t *>(lds + 8192 + 128 + 8192 + 1024);
	unsigned *shm = reinterpret_cast<unsigned *>(lds + 8192 + 128 + 8192 + 1024 + 512);
	for (int i = tid; i < (n + 128) / 4; i += NT) {                 /* :886 the samples lose their two low bits (the padding behind them is zero) */
		uint32_t v = reinterpret_cast<const uint32_t *>(c->ll_bytes + lo)[i];
		if (4 * i < n) { v &= 0xFCFCFCFCu; reinterpret_cast<uint32_t *>(c->ll_bytes + lo)[i] = v; }
		reinterpret_cast<uint32_t *>(s)[i] = v;
	}
	BARRIER();
	for (int i = tid; i < n; i += NT) { int byte; X[LCX(i)] = i ? (uint8_t)(ll_chroma_token(s, i, &byte) - i) : 1; }
	BARRIER();
	{
		const int b0 = tid * 32, end = b0 + 32;
		for (int e = end - 1; e >= b0; e--) {
			const int nx = e + X[LCX(e)];
			X[LCX(e)] = (uint8_t)(nx >= end ? nx - end : X[LCX(nx)]);
		}
	}
	BARRIER();
	if (tid == 0) { int pos = 1; while (pos < n) { const int b = pos >> 5; entry[b] = (int16_t)pos; pos = (b + 1) * 32 + X[LCX(pos)]; } }   /* strides are <= 16: every block is entered */
	BARRIER();
	const int e0 = entry[tid], end = tid * 32 + 32;
	unsigned cnt = 0;
	for (int i = e0; i < end;) { int byte; i = ll_chroma_token(s, i, &byte); cnt++; }
	unsigned total;
	const unsigned off = block_exscan(cnt, tid, shm, &total);
	const int j = c->m->ll_comp_y_len;
	uint8_t *o = c->ll_comp + j + 1 + off;
	for (int i = e0; i < end;) { int byte; i = ll_chroma_token(s, i, &byte); *o++ = (uint8_t)byte; }
	if (tid == 0) {
		c->ll_comp[j] = s[0];
		c->m->res_high = c->m->res_low;                              /* :887 */
		c->m->ch_res_len = j + 1 + (int)total;
	}
}

DEV void luma_p1_par(Ctx *c, int tid, int *pos)
{
	PROF_BEGIN();
	if (c->compat) for (int k = tid; k < 512; k += NT) c->ll1[Q + k] = 0;   /* the passes before the LL2 emission see nothing behind res256 (nhw_tail_par.h, luma_p3_par) */
	tag_l2_details_par(c, tid);
	BARRIER();
	if (!tid) PROF(c, 0);
}
DEV void luma_p2_par(Ctx *c, int tid, int16_t *lds)
{
	PROF_BEGIN();
	apply_tags_par(c, tid, lds);
	BARRIER();
	if (!tid) PROF(c, 2);
	precompensate_ll1_par(c, tid, lds);
	if (!tid) PROF(c, 3);
}
DEV void luma_p3_par(Ctx *c, int tid, int *pos, int *sh_misc, int16_t *lds, bool restore /* Y17: the level-2 block back into the work plane -- only where somebody reads it there: quality <= 12 (no second closed loop, whose dequantiser simulation is the one reader and takes the block from l2save itself) and the stage checks */)
{
	PROF_BEGIN();
	if (c->compat) {
		/* the stock binary's heap: tree1 is carved out of the freed kernel map (q<=21), so what the LL2 coder reads behind the luma
		 * samples is map bytes; and res256 is followed by 8 bytes of map, the next chunk's size word and resIII */
		for (int i = (Q >> 2) + tid; i < (Q >> 2) + (Q >> 3) + 64; i += NT) {
			int b = 0;
			if (c->q < 22) {
				const int s = 131088 + (i >> 1), v = (uint16_t)c->stale[4 + ((s >> 9) - 272) * W + (s & 511)];
				b = (i & 1) ? v >> 8 : v & 255;
			}
			c->ll_bytes[i] = (uint8_t)b;
		}
		int16_t *tail = c->ll1 + Q;
		for (int k = tid; k < 512; k += NT)
			tail[k] = (int16_t)(k < 4 ? (c->q < 22 ? c->stale[k] : 0) : k == 4 ? 0x0011 : k == 5 ? 0x0002 : k < 8 ? 0 : c->l2save[k - 8]);
	}
	else for (int i = (Q >> 2) + tid; i < (Q >> 2) + (Q >> 3) + 64; i += NT) c->ll_bytes[i] = 0;
	BARRIER();
	{                                                             /* Y16 */
		uint8_t *ls = reinterpret_cast<uint8_t *>(lds);
		for (int i = tid; i < 16640 / 16; i += NT) reinterpret_cast<uint4 *>(ls)[i] = reinterpret_cast<const uint4 *>(c->ll_bytes)[i];
		BARRIER();
		ll_code_luma_par(c, tid, ls);
		if (!tid) PROF(c, 5);
	}
	if (c->compat && c->q <= 13 && tid < 128) {
		/* the same heap once more: Y20 finds its level-2 parents up to 128 entries behind resIII -- 8 bytes of map (row 256, columns 8..11),
		 * the next chunk's size word (0x6011) and that chunk, tree1 */
		const int k = tid;
		c->l2save[Q + k] = (int16_t)(k < 4 ? c->stale[4 + 9 * W + k] : k == 4 ? 0x6011 : k < 8 ? 0 : (c->ll_bytes[2 * (k - 8)] | c->ll_bytes[2 * (k - 8) + 1] << 8));
	}
	BARRIER();
	if (restore) copy_block_par(c->l2save, H, c->proc, W, H, H, tid);          /* Y17 :749-755 */
	BARRIER();
	if (!tid) PROF(c, 6);
}
/* ---------------------------------------------------------------- Y20 for q <= 15 (nhw_encoder.c:804-968) */
/* zeroing rule shared by the three bands at q <= 13: a small coefficient goes when its level-2 parent is small, or together with a
 * neighbour when the pair nearly cancels (:875-886 etc.).  lv: the value the walk finds on the left (for the first cell of a row of the
 * lower bands that is the last cell of the row above, which belongs to another thread: handed in, and a write to it handed back) */
DEV void thin_by_parent(int16_t *v, int lv, int parent, int lim_parent, int lim_pair, bool *zero_left)
{
	if (iabs(parent) < lim_parent) v[0] = 0;
	else if (iabs(v[0] + lv) < lim_pair && iabs(v[1]) < lim_pair) { v[0] = 0; *zero_left = true; }
	else if (iabs(v[0] + v[1]) < lim_pair && iabs(lv) < lim_pair) { v[0] = 0; v[1] = 0; }
}
DEV int16_t keep_loud(int v, int q) { return q > 10 ? (int16_t)(v >= 16 ? 7 : v <= -16 ? -7 : 0) : (int16_t)0; }   /* :947-953 */
struct ThinT { int t1, t2, t3, t4, t5; };
/* One cell of the walk: x the cell as the walk finds it, lv the cell on its left as its own visit left it, x1 the cell on its right (not yet
 * visited); the step may zero either of them.  Three sets of limits: rows < 256 (limC 0: no "else"), rows >= 256 left / right half. */
struct ThinRule { int limA, limB, limC, lim_parent, lim_pair, loud; };
DEV int thin_cell(int x, int &lv, int &x1, int parent, const ThinRule &R, int q, bool &zl)
{
	int v = x;
	zl = false;
	if (iabs(v) >= DEADZONE && iabs(v) < R.limA) {
		if (iabs(parent) < R.lim_parent) v = 0;
		else if (iabs(v + lv) < R.lim_pair && iabs(x1) < R.lim_pair) { v = 0; zl = true; lv = 0; }
		else if (iabs(v + x1) < R.lim_pair && iabs(lv) < R.lim_pair) { v = 0; x1 = 0; }
	}
	if (iabs(v) >= DEADZONE && iabs(v) < R.limB) {
		if ((iabs(lv) < DEADZONE && iabs(x1) < DEADZONE) || iabs(v) < R.limC) v = R.loud ? keep_loud(v, q) : 0;
	}
	return v;
}
/* A row of the walk on a wavefront, a lane NC consecutive cells (the first `nact` of them visited, the rest only looked at): what travels
 * from cell to cell is the value a visit leaves behind and "your cell has been zeroed"; the lanes start from the cells as they are and hand
 * theirs on until nothing moves.  o: the lane's cells, o_next: the cell behind them, left_in: the cell in front of lane 0's.
 * Returns: e = the cells afterwards; zero_left: lane 0's first cell wants the cell in front of the row zeroed; zero_right: lane 63's last
 * visit zeroed the cell behind the lanes. */
template <int NC>
DEV void thin_row_wave(const int *o, int o_next, int left_in, const int *par, int nact, const ThinRule &R, int q, int lane, int *e, bool &zero_left, bool &zero_right)
{
	const int left0 = __shfl_up(o[NC - 1], 1);
	int lv_in = lane ? left0 : left_in, f_in = 0, lv_out, f_out;
	bool zlf;
	for (;;) {
		int cur[NC];
#pragma unroll
		for (int k = 0; k < NC; k++) cur[k] = o[k];
		if (f_in) cur[0] = 0;
		int lv = lv_in;
		zlf = false; f_out = 0;
#pragma unroll
		for (int k = 0; k < NC; k++) {
			int x1 = k < NC - 1 ? cur[k < NC - 1 ? k + 1 : k] : o_next;
			if (k < nact) {
				bool zl;
				e[k] = thin_cell(cur[k], lv, x1, par[k], R, q, zl);
				if (zl) { if (k) e[k - 1] = 0; else zlf = true; }
			} else e[k] = cur[k];
			if (k < NC - 1) cur[k < NC - 1 ? k + 1 : k] = x1; else f_out = x1 != o_next;
			lv = e[k];
		}
		lv_out = lv;
		int nl = __shfl_up(lv_out, 1), nf = __shfl_up(f_out, 1);
		if (!lane) { nl = left_in; nf = 0; }
		if (!__any(nl != lv_in || nf != f_in)) break;
		lv_in = nl; f_in = nf;
	}
	const int zn = __shfl_down((int)zlf, 1);                        /* the lane on my right wants my last cell zeroed (its own visits are over) */
	if (lane < 63 && zn) e[NC - 1] = 0;
	zero_left = __shfl((int)zlf, 0) != 0;
	zero_right = __shfl(f_out, 63) != 0;
}
DEV void thin_l1_low_par(Ctx *c, int tid, int *sh /* shared, >= 2 * NT + 2 ints */)
{
	int16_t *p = c->proc;
	const int16_t *par = c->l2save;
	const int q = c->q;
	if (q >= 14) {                                               /* :804-832, pointwise */
		const int t2 = q == 15 ? 19 : 20;
		int16_t *lo = p + 2 * Q;
		for (int i0 = tid; i0 < 2 * Q / 8; i0 += 4 * NT) {         /* eight cells an item, four items in flight */
			uint4 w[4];
#pragma unroll
			for (int u = 0; u < 4; u++) w[u] = reinterpret_cast<const uint4 *>(lo)[i0 + u * NT];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int idx = 8 * (i0 + u * NT);
				const bool left = (idx & (W - 1)) < H;
				uint32_t x[4] = { w[u].x, w[u].y, w[u].z, w[u].w };
				bool ch = false;
#pragma unroll
				for (int e = 0; e < 8; e++) {
					const int v = (int16_t)(x[e >> 1] >> (16 * (e & 1))), m = iabs(v);
					if (m < DEADZONE) continue;
					int nv = v;
					if (left) { if (m < 11) nv = 0; }
					else if (m < t2) nv = v >= 14 ? 7 : v <= -14 ? -7 : 0;
					if (nv != v) { x[e >> 1] = (x[e >> 1] & ~(0xFFFFu << (16 * (e & 1)))) | ((uint32_t)(uint16_t)nv << (16 * (e & 1))); ch = true; }
				}
				if (ch) reinterpret_cast<uint4 *>(lo)[i0 + u * NT] = make_uint4(x[0], x[1], x[2], x[3]);
			}
		}
		BARRIER();
		return;
	}
	ThinT t;
	if (q == 13) t = ThinT{ 15, 27, 10, 6, 3 };
	else {                                                       /* thresholds follow the number of loud coefficients of the lower bands (:836-868) */
		unsigned loud = 0;
		for (int idx = tid; idx < 2 * Q / 8; idx += NT) {
			const uint4 w = reinterpret_cast<const uint4 *>(p + 2 * Q)[idx];
			const uint32_t ww[4] = { w.x, w.y, w.z, w.w };
			for (int e = 0; e < 4; e++) { loud += iabs((int16_t)(ww[e] & 0xFFFF)) >= 12; loud += iabs((int16_t)(ww[e] >> 16)) >= 12; }
		}
		{ unsigned total; block_exscan(loud, tid, reinterpret_cast<unsigned *>(sh), &total); loud = total; BARRIER(); }
		t = ThinT{ 16, 28, 11, 8, 5 };
		if (loud > 12500) t = ThinT{ 19, 31, 13, 9, 6 };
		else if (loud > 10000) t = ThinT{ 18, 30, 12, 8, 6 };
		else if (loud >= 7000) t = ThinT{ 17, 29, 11, 8, 5 };
		if (q == 11) { if (loud > 12500) { t.t1++; t.t2++; t.t3++; t.t4++; t.t5++; } else t.t1++; }
		else if (q <= 10) {
			if (loud > 12500) { t.t1 += 3; t.t2 += 3; t.t3 += 2; t.t4 += 3; t.t5 += 3; }
			else { t.t1 += 3; t.t2 += 2; t.t3 += 2; t.t4 += 2; t.t5 += 2; }
		}
	}
	/* Both walks run a wavefront per row (thin_row_wave).  Until round 3 they were a thread per row on 32-column LDS tiles (72-byte row pieces
	 * in and out of eight tiles, a copy of the lower half for the rows that had to be walked again: 13 GB per batch, 2.3 ms at quality 1). */
	const int lane = tid & 63, wv = tid >> 6;
	/* rows 0..255, columns 256..511 (:871-896): the walk reaches one cell over either end of its band -- column 255 of its row and the first
	 * cell of the next row -- which no other row of this walk reads or writes */
	{
		const ThinRule R = { t.t3 + 2, t.t3, 0, t.t4, t.t5, 0 };
		uint2 w = make_uint2(0, 0); uint32_t pw = 0; int edge = 0;
#define UP_LOAD(r) do { w = *reinterpret_cast<const uint2 *>(p + (size_t)(r) * W + H + 4 * lane); pw = *reinterpret_cast<const uint32_t *>(par + (r) * (H / 2) + H / 2 + 2 * lane); \
		if (lane == 0) edge = p[(size_t)(r) * W + H - 1]; if (lane == 63) edge = p[(size_t)((r) + 1) * W]; } while (0)
		int r = wv;
		UP_LOAD(r);
		for (; r < H; r += NT / 64) {
			int o[4], e[4];
			unpack4(w, o);
			const int pr[4] = { (int16_t)(pw & 0xFFFF), (int16_t)(pw & 0xFFFF), (int16_t)(pw >> 16), (int16_t)(pw >> 16) };
			const int my_edge = edge, row = r;
			if (r + NT / 64 < H) UP_LOAD(r + NT / 64);
			bool busy = false;                                      /* a row without a cell inside the walk's range stays as it is (most rows of quality 1) */
#pragma unroll
			for (int k = 0; k < 4; k++) busy |= iabs(o[k]) >= DEADZONE && iabs(o[k]) < R.limA;
			if (!__any(busy)) continue;
			const int sd = __shfl_down(o[0], 1);
			bool zl, zr;
			thin_row_wave<4>(o, lane < 63 ? sd : my_edge, __shfl(my_edge, 0), pr, 4, R, q, lane, e, zl, zr);
			uint2 out;
			out.x = (uint32_t)(uint16_t)e[0] | ((uint32_t)(uint16_t)e[1] << 16); out.y = (uint32_t)(uint16_t)e[2] | ((uint32_t)(uint16_t)e[3] << 16);
			*reinterpret_cast<uint2 *>(p + (size_t)row * W + H + 4 * lane) = out;
			if (lane == 0 && zl) p[(size_t)row * W + H - 1] = 0;
			if (lane == 63 && zr) p[(size_t)(row + 1) * W] = 0;
		}
#undef UP_LOAD
	}
	BARRIER();
	/* rows 256..511, columns 0..510 (:898-967).  The first cell of a row looks at (and may zero) the last cell of the row above, which that
	 * row may have zeroed in its own last step: the only link between rows.  A row whose cell above is not zero to begin with is therefore
	 * walked for both cases -- the cell above as it was (result to the plane) and zeroed (result to a scratch row) -- and says for either
	 * whether it zeroes its own last cell; one thread then follows the chain of choices down the rows, the rows that turn out to have been
	 * entered on a zeroed cell are copied from the scratch rows, and the zeroings of the cell above are applied last (the row above read
	 * that cell before this row would have written it). */
	{
		int16_t *scratch = c->jpeg;                               /* free here: the reference has released im_jpeg (:781) */
		int16_t *above = reinterpret_cast<int16_t *>(sh);           /* [H] the cell before every row, as it is now */
		uint8_t *fl = reinterpret_cast<uint8_t *>(sh) + 2 * H;      /* [H] per row: bit 0 / 1: zeroes its own last cell (entered as is / zeroed), 2 / 3: wants the cell above zeroed, 4: walked twice, 5: the choice */
		above[tid] = p[(size_t)(H + tid) * W - 1];
		BARRIER();
		const ThinRule RA = { t.t1 + 2, t.t1, t.t1 - 4, t.t4, t.t5, 0 }, RB = { t.t2 + 1, t.t2, t.t2 - 5, t.t4 + 1, t.t5, 1 };
		const ThinRule R = lane < 32 ? RA : RB;
		uint4 w = make_uint4(0, 0, 0, 0); uint2 pw = make_uint2(0, 0);
#define LO_LOAD(rr) do { w = *reinterpret_cast<const uint4 *>(p + (size_t)(H + (rr)) * W + 8 * lane); pw = *reinterpret_cast<const uint2 *>(par + (rr) * (H / 2) + Q / 2 + 4 * lane); } while (0)
		int rr = wv;
		LO_LOAD(rr);
		for (; rr < H; rr += NT / 64) {
			const uint32_t ww[4] = { w.x, w.y, w.z, w.w }, pp[2] = { pw.x, pw.y };
			int o[8], pr[8], e[8];
#pragma unroll
			for (int k = 0; k < 4; k++) { o[2 * k] = (int16_t)(ww[k] & 0xFFFF); o[2 * k + 1] = (int16_t)(ww[k] >> 16); }
#pragma unroll
			for (int k = 0; k < 8; k++) pr[k] = (int16_t)(pp[k >> 2] >> (16 * ((k >> 1) & 1)));
			const int row = rr, left = above[rr];
			if (rr + NT / 64 < H) LO_LOAD(rr + NT / 64);
			bool busy = false;
#pragma unroll
			for (int k = 0; k < 8; k++) busy |= iabs(o[k]) >= DEADZONE && iabs(o[k]) < R.limA;
			if (!__any(busy)) { if (lane == 0) fl[row] = 0; continue; }
			const int sd = __shfl_down(o[0], 1);
			const int nact = lane < 63 ? 8 : 7;                     /* column 511 is only looked at */
			unsigned flags = 0;
			for (int pass = 0; pass < (left != 0 && row > 0 ? 2 : 1); pass++) {   /* (the first row's cell above is what the upper walk left: no second case) */
				bool zl, zr;
				thin_row_wave<8>(o, lane < 63 ? sd : 0, pass ? 0 : left, pr, nact, R, q, lane, e, zl, zr);
				const int last = __shfl(e[7], 63), last0 = __shfl(o[7], 63);
				flags |= ((last == 0 && last0 != 0) ? 1u : 0u) << pass | (zl ? 4u : 0u) << pass | (pass ? 16u : 0u);
				int16_t *dst = (pass ? scratch : p) + (size_t)(H + row) * W + 8 * lane;
				*reinterpret_cast<uint4 *>(dst) = make_uint4((uint32_t)(uint16_t)e[0] | ((uint32_t)(uint16_t)e[1] << 16), (uint32_t)(uint16_t)e[2] | ((uint32_t)(uint16_t)e[3] << 16),
				                                             (uint32_t)(uint16_t)e[4] | ((uint32_t)(uint16_t)e[5] << 16), (uint32_t)(uint16_t)e[6] | ((uint32_t)(uint16_t)e[7] << 16));
			}
			if (!(flags & 16u)) flags |= (flags & 1u) << 1 | (flags & 4u) << 1;   /* one case: both read the same */
			if (lane == 0) fl[row] = (uint8_t)flags;
		}
#undef LO_LOAD
		BARRIER();
		if (tid == 0) {
			bool zeroed = false;                                    /* the row above zeroes its last cell (in the case that applies to it) */
			for (int r2 = 0; r2 < H; r2++) {
				const unsigned f = fl[r2];
				const bool second = zeroed && (f & 16u);
				if (second) fl[r2] = (uint8_t)(f | 32u);
				zeroed = (f >> (second ? 1 : 0)) & 1u;
			}
		}
		BARRIER();
		for (int r2 = wv; r2 < H; r2 += NT / 64)
			if (fl[r2] & 32u) *reinterpret_cast<uint4 *>(p + (size_t)(H + r2) * W + 8 * lane) = *reinterpret_cast<const uint4 *>(scratch + (size_t)(H + r2) * W + 8 * lane);
		BARRIER();
		{ const unsigned f = fl[tid]; if ((f >> ((f & 32u) ? 3 : 2)) & 1u) p[(size_t)(H + tid) * W - 1] = 0; }
	}
	BARRIER();
}

/* Y19..Y31 run as four kernels (a: Y19-Y23, b: Y24-Y25, c: Y26-Y29, d: Y30-Y31) so that each gets the register
 * and LDS budget of its own passes: one kernel for all of them ran at 4 waves/SIMD. */
DEV void luma_p4a_par(Ctx *c, int tid, int16_t *lds)
{
	const int q = c->q;
	PROF_BEGIN();
	if (q > 21) copy_block_par(c->jpeg, W, c->first_order, H, H, H, tid);   /* Y19 :766-777 */
	if (q <= 15) thin_l1_low_par(c, tid, reinterpret_cast<int *>(lds));     /* Y20 (:804-968) */
	else if (q < 20) {                                                      /* Y20 (:783-801) */
		/* (eight cells an item, four items in flight, an item stored only if it changed: a cell a thread and turn was 512 dependent 2-byte
		 * round trips a thread) */
		int16_t *p = c->proc + 2 * Q;
		for (int i0 = tid; i0 < 2 * Q / 8; i0 += 4 * NT) {
			uint4 w[4];
#pragma unroll
			for (int u = 0; u < 4; u++) w[u] = reinterpret_cast<const uint4 *>(p)[i0 + u * NT];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int idx = 8 * (i0 + u * NT);
				const bool left = (idx & (W - 1)) < H;                       /* (an item lies in one half of its row) */
				uint32_t x[4] = { w[u].x, w[u].y, w[u].z, w[u].w };
				bool ch = false;
#pragma unroll
				for (int e = 0; e < 8; e++) {
					const int v = (int16_t)(x[e >> 1] >> (16 * (e & 1))), m = iabs(v);
					if (m >= DEADZONE && (left ? m < 9 : m <= 14)) { x[e >> 1] = (x[e >> 1] & ~(0xFFFFu << (16 * (e & 1)))) | ((uint32_t)(uint16_t)(v > 0 ? 7 : -7) << (16 * (e & 1))); ch = true; }
				}
				if (ch) reinterpret_cast<uint4 *>(p)[i0 + u * NT] = make_uint4(x[0], x[1], x[2], x[3]);
			}
		}
	}
	BARRIER();
	if (!tid) PROF(c, 8);
	if (q > 16) tag_small_runs_par(c, tid, lds);                            /* Y21 (:970) */
	BARRIER();
	if (!tid) PROF(c, 9);
	if (q <= 12) return;                                                    /* no second closed loop, no residual lists (:1081, :1498) */
	const int res_setting = q >= 20 ? 3 : q >= 18 ? 4 : q >= 15 ? 6 : 8;    /* :1075-1079 */
	residuals_fused_par(c, res_setting, tid, lds);                          /* Y22 + Y23 */
}
DEV void luma_p4b_par(Ctx *c, int tid, int *pos, int16_t *lds)
{
	PROF_BEGIN();
	if (c->q > 21) adjust_first_order_par(c, tid, lds);                     /* Y24 */
	build_poslists_par(c, tid, pos, lds);                                   /* Y25 */
	if (!tid) PROF(c, 12);
}
DEV void luma_p4c_par(Ctx *c, int tid, int *pos, int16_t *lds, bool copy_ll /* quality > 21 and the stage checks: the level-2 block goes back into the work plane (Y26); otherwise the quantiser, its only reader, takes it from l2save itself (quant_load_row) */)
{
	PROF_BEGIN();
	if (copy_ll)
	for (int g = tid; g < Q / 8; g += NT) {                                  /* Y26 :1893-1910, 8 cells per item */
		const int r = g >> 5, j0 = (g & 31) * 8;
		uint4 v = *reinterpret_cast<const uint4 *>(c->l2save + r * H + j0);
		if (r < H / 2 && j0 < H / 2) {
			uint32_t w[4] = { v.x, v.y, v.z, v.w };
			for (int e = 0; e < 4; e++) {
				if ((int16_t)(w[e] & 0xFFFF) <= 8000) w[e] &= 0xFFFF0000u;
				if ((int16_t)(w[e] >> 16) <= 8000) w[e] &= 0x0000FFFFu;
			}
			v = make_uint4(w[0], w[1], w[2], w[3]);
		}
		*reinterpret_cast<uint4 *>(c->proc + r * W + j0) = v;
	}
	BARRIER();
	if (!tid) PROF(c, 13);
	(void)lds;
	clean_details_seq(c, tid, copy_ll);                                     /* Y27 */
	if (!tid) PROF(c, 14);
}
/* Y29 (q > 21): im_recons_wavelet_band (image_processing.c:523-556) + wavelet_synthesis_high_quality_settings
 * (wavelet_filterbank.c:498-707), workgroup-parallel.
 *
 * band_recons walks the quantised LH1 band (rows < 256, columns 256..511) in raster order and writes the decoder's value of every cell
 * into a compact 256 x 256 plane through a running index t: a zero code advances t, a plain code writes b[t++], a triple mark (127 /
 * 129) writes b[t-1], b[t], b[t+1], advances t by TWO and skips the next cell.  A skipped cell is exactly the second step of the
 * mark, so t stays the cell's linear index -- except behind a mark in the LAST column of a row, whose skip has no cell to eat: from
 * there on everything sits one slot further.  A thread takes a row: the cells a walk visits are every second one of each run of marks
 * (serial along the row), the row's base index is 256 r + the number of such row-end marks above it (prefix sum), and the three
 * writes that cross a row boundary are ordered by hand: (r, 255)'s b[t+1] before row r+1 writes, (r, 0)'s b[t-1] after row r-1 has. */
DEV int band_value(int a)
{
	if ((a & 7) != 0) { const int k = (a >= 0 && a < 109) ? big_index(a) : 0; return k > 0 ? 123 + (k << 3) : (k << 3) - 123; }
	return a > 128 ? a - 125 : a - 131;
}
DEV void luma_p4c2_par(Ctx *c, int tid, unsigned *shm /* [NT / 64 + 1] */, int *sh /* [2 * NT + 2] */)
{
	PROF_BEGIN();
	const int q = c->q;
	if (q <= 21) return;
	const int16_t *p = c->proc;
	int16_t *b = c->band;
	for (int i = tid; i < Q / 8; i += NT) reinterpret_cast<uint4 *>(b)[i] = make_uint4(0, 0, 0, 0);
	const int r = tid, lane = tid & 63, wv = tid >> 6;
	/* A wavefront takes a row, lane l its cells l, l + 64, l + 128, l + 192 (coalesced loads; a thread walking "its" row reads one cell of a
	 * different line at every step -- 148 GB per batch that way).  The cells a walk visits are every second one of each run of marks:
	 * alt_runs on the ballots.  First the rows' end marks (does the walk end in a mark on the last cell?) ... */
	for (int row = wv; row < H; row += NT / 64) {
		int a[4];
		for (int k = 0; k < 4; k++) a[k] = p[(size_t)row * W + H + lane + 64 * k];
		M4 cand;
		BALLOT4(cand, a, x == 127 || x == 129);
		if (!(cand.w[3] >> 63)) { if (!lane) sh[row] = 0; continue; }        /* the last cell is no mark (the rule): the walk cannot end in one */
		const M4 fired = alt_runs(cand);
		if (!lane) sh[row] = (int)(fired.w[3] >> 63);
	}
	BARRIER();
	unsigned tot;
	const int end_mark = sh[r];
	const int base_r = r * H + (int)block_exscan((unsigned)end_mark, tid, shm, &tot);
	sh[H + r] = base_r;
	BARRIER();
	/* ... then every slot of the compact plane from the one cell whose write is the last to land there: the mark on its right (b[t-1]),
	 * the cell itself (a mark's centre or a plain code), the mark on its left (b[t+1]) -- in that order of precedence, which is the
	 * raster order of the writes.  Across a row boundary: the last slot of a row that does not end in a mark takes the b[t-1] of a mark
	 * in column 0 of the next row; behind a row that does end in one, the extra slot (where everything moves one further) holds that
	 * mark's b[t+1] unless the next row's column 0 is a mark as well. */
	for (int row = wv; row < H; row += NT / 64) {
		int a[4];
		for (int k = 0; k < 4; k++) a[k] = p[(size_t)row * W + H + lane + 64 * k];
		const int nxt = row + 1 < H ? (int)p[(size_t)(row + 1) * W + H] : 0;   /* column 0 of the next row is always visited */
		const bool next_mark0 = nxt == 127 || nxt == 129;
		M4 cand;
		BALLOT4(cand, a, x == 127 || x == 129);
		const int base = sh[H + row], ends = sh[row];
		if (!(cand.w[0] | cand.w[1] | cand.w[2] | cand.w[3])) {                 /* a row without marks (most): every code lands in its own slot */
			for (int k = 0; k < 4; k++) {
				const int j = lane + 64 * k, x = a[k];
				if (j == H - 1 && next_mark0) b[base + j] = (int16_t)(nxt == 127 ? 5 : -5);
				else if (x != 128) b[base + j] = (int16_t)band_value(x);
			}
			continue;
		}
		const M4 fired = alt_runs(cand), fnext = dn1(fired), fprev = up1(fired);
		for (int k = 0; k < 4; k++) {
			const int j = lane + 64 * k, x = a[k];
			const int xr = right_of(a, k, 4, 1, lane);                            /* cell j + 1 */
			const int seam = k ? __shfl(a[k - 1], 63) : 0;
			const int from_left = __shfl(a[k], (lane + 63) & 63);                 /* (every lane takes part: a lane that sits out answers 0) */
			const int xl = lane ? from_left : seam;                               /* cell j - 1 */
			int val = 0; bool wr = true;
			if (TB(fnext, k)) val = xr == 127 ? 5 : -5;
			else if (TB(fired, k)) val = x == 127 ? 6 : -7;
			else if (TB(fprev, k)) val = xl == 127 ? 5 : -5;
			else if (x != 128) val = band_value(x);
			else wr = false;
			if (j == H - 1 && !ends && next_mark0) { val = nxt == 127 ? 5 : -5; wr = true; }
			if (wr) b[base + j] = (int16_t)val;
			if (j == H - 1 && ends) b[base + H] = (int16_t)(next_mark0 ? (nxt == 127 ? 5 : -5) : (x == 127 ? 5 : -5));
		}
	}
	BARRIER();

	if (!tid) PROF(c, 50);
	/* half synthesis of the kept first-order LL + that band against the original pass-1 plane (:509-541): pointwise */
	int16_t *hs = c->hs;
	const int thr = q > 22 ? 30 : 34;
	auto wave_exscan = [&](int v, int &total) {
		int x = v;
		for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
		total = __shfl(x, 63);
		return x - v;
	};
	/* (a wavefront per row, a lane the row's eight cells from 8 l on -- the layout of the list passes below, whose counts are taken here, where
	 * the values are made: sh[row] = the row's 32000 / 32500 cells, sh[H + row] = its 30000 / 31000 cells | those of columns 254, 255 << 16) */
	/* (an item's seven loads are asked for an item ahead: behind the store of the item before they may alias it for all the compiler knows,
	 * and every turn waited for its own) */
	struct HqIn { uint2 lw, hw; int l4, h0, h5; uint4 kw; };
	auto hq_fetch = [&](int idx) {
		const int rr = idx >> 6, k0 = (idx & 63) * 4;
		const int16_t *lo = c->first_order + rr * H, *hi = b + rr * H;
		HqIn in;
		in.lw = *reinterpret_cast<const uint2 *>(lo + k0); in.hw = *reinterpret_cast<const uint2 *>(hi + k0);
		in.l4 = k0 + 4 < H ? (int)lo[k0 + 4] : 0x7fffffff; in.h0 = k0 > 0 ? (int)hi[k0 - 1] : 0x7fffffff; in.h5 = k0 + 4 < H ? (int)hi[k0 + 4] : 0x7fffffff;
		in.kw = *reinterpret_cast<const uint4 *>(c->keep + rr * W + 2 * k0);
		return in;
	};
	HqIn nin = hq_fetch(tid);
	for (int idx = tid; idx < Q / 4; idx += NT) {                               /* four cells (eight outputs) per item: 8- and 16-byte accesses */
		const int rr = idx >> 6, k0 = (idx & 63) * 4;
		int n_q3 = 0, n_e = 0, n_c = 0;
		const HqIn in = nin;
		if (idx + NT < Q / 4) nin = hq_fetch(idx + NT);
		int l[5], h[6];                                                          /* lo[k0 .. k0 + 4], hi[k0 - 1 .. k0 + 4] */
		{
			const uint2 lw = in.lw, hw = in.hw;
			l[0] = (int16_t)(lw.x & 0xFFFF); l[1] = (int16_t)(lw.x >> 16); l[2] = (int16_t)(lw.y & 0xFFFF); l[3] = (int16_t)(lw.y >> 16);
			h[1] = (int16_t)(hw.x & 0xFFFF); h[2] = (int16_t)(hw.x >> 16); h[3] = (int16_t)(hw.y & 0xFFFF); h[4] = (int16_t)(hw.y >> 16);
			l[4] = in.l4 != 0x7fffffff ? in.l4 : l[3];
			h[0] = in.h0 != 0x7fffffff ? in.h0 : h[1];
			h[5] = in.h5 != 0x7fffffff ? in.h5 : h[4];
		}
		const uint4 kw = in.kw;
		const uint32_t kk[4] = { kw.x, kw.y, kw.z, kw.w };
		uint32_t out[4];
#pragma unroll
		for (int t = 0; t < 4; t++) {
			int16_t o[2];
			o[0] = (int16_t)((int16_t)(l[t] << 3) - ((h[t + 1] + h[t]) << 1));
			o[1] = (int16_t)((int16_t)((l[t] + l[t + 1]) << 2) + (6 * h[t + 1] - h[t] - h[t + 2]));
			for (int e = 0; e < 2; e++) {
				const int d = (int16_t)(kk[t] >> (16 * e)) - o[e];
				if (iabs(d) > thr) o[e] = (int16_t)((q > 22 && iabs(d) > 56) ? (d > 0 ? 32000 : 32500) : (d > 0 ? 30000 : 31000));
			}
			out[t] = (uint32_t)(uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16);
			for (int e = 0; e < 2; e++) {
				n_q3 += o[e] == 32000 || o[e] == 32500;
				const bool hit = o[e] == 30000 || o[e] == 31000;
				if ((lane == 31 || lane == 63) && t == 3) { if (lane == 31) n_c += hit; } else n_e += hit;   /* columns 254, 255 (char_res1) and 510, 511 (nowhere) are no list entries */
			}
		}
		*reinterpret_cast<uint4 *>(hs + rr * W + 2 * k0) = make_uint4(out[0], out[1], out[2], out[3]);
		/* the row's totals (a lane counts 0 .. 8): a ballot and a population count a bit, no trip through the LDS crossbar */
		int t_q3 = 0, t_e = 0;
		for (int bit = 0; bit < 4; bit++) { t_q3 += __popcll(__ballot((n_q3 >> bit) & 1)) << bit; t_e += __popcll(__ballot((n_e >> bit) & 1)) << bit; }
		const int t_c = __builtin_amdgcn_readlane(n_c, 31);
		if (!lane) { sh[rr] = t_q3; sh[H + rr] = t_e | (t_c << 16); }
	}
	BARRIER();
	if (!tid) PROF(c, 51);
	/* The two list passes below visit the cells of hs in raster order; a wavefront takes a row (512 cells, lane l the eight from 8 l on:
	 * 16-byte loads), counts first, the rows' offsets from a prefix sum over the rows, then the entries at offset + (entries of the lanes
	 * before mine). */
	auto load8 = [&](const int16_t *rowp, int v[8]) {
		const uint4 w = reinterpret_cast<const uint4 *>(rowp)[lane];
		const uint32_t ww[4] = { w.x, w.y, w.z, w.w };
		for (int e = 0; e < 4; e++) { v[2 * e] = (int16_t)(ww[e] & 0xFFFF); v[2 * e + 1] = (int16_t)(ww[e] >> 16); }
	};
	/* the rows' offsets from prefix sums over the rows' counts; then one pass writes both lists: the entries at offset + (entries of the
	 * lanes before mine) */
	const unsigned q3_r = (unsigned)sh[r], e_r = (unsigned)sh[H + r] & 0xFFFFu, nc_r = (unsigned)sh[H + r] >> 16;
	BARRIER();
	unsigned te, tc;
	const unsigned aq_r = block_exscan(q3_r, tid, shm, &tot);
	const unsigned ae_r = block_exscan(e_r, tid, shm, &te);
	const unsigned ac_r = block_exscan(nc_r, tid, shm, &tc);
	sh[r] = (int)aq_r; sh[H + r] = (int)(ae_r | (ac_r << 20));
	BARRIER();
	if (!tid) PROF(c, 52);
	/* qsetting3: the 32000 / 32500 cells (:547-564, q23).  The position list of the 30000 / 31000 cells (:571-610): columns 254, 255 and 510,
	 * 511 are a row mark each (raw value 254, behind the entries of the columns before them), the first pair reported through char_res1 instead */
	uint8_t *raw = c->raw, *pay = c->pay;
	for (int row = wv; row < H; row += NT / 64) {
		int v[8], cnt = 0, nq = 0, total, tq;
		load8(hs + (size_t)row * W, v);
		const bool special = lane == 31 || lane == 63;
		for (int e = 0; e < 8; e++) { nq += v[e] == 32000 || v[e] == 32500; if (e < (special ? 6 : 8)) cnt += v[e] == 30000 || v[e] == 31000; }
		if (tot) {
			int at = sh[row] + wave_exscan(nq, tq);
			for (int e = 0; e < 8; e++) {
				const int i = row * W + 8 * lane + e;
				if (v[e] == 32000) c->qsetting3[at++] = (uint32_t)(i << 1);
				else if (v[e] == 32500) c->qsetting3[at++] = (uint32_t)(i << 1) + 1;
			}
		}
		const int off = wave_exscan(cnt, total);
		const int first_half = __shfl(off, 32);                                  /* entries of columns 0..253 */
		const int base_e = sh[H + row] & 0xFFFFF;
		int ae = base_e + off;                                                   /* payload index; the raw list has two marks per row before this row's, one more from column 256 on */
		int an = ae + 2 * row + (lane >= 32);
		for (int e = 0; e < (special ? 6 : 8); e++) {
			if (v[e] == 30000) { raw[an++] = (uint8_t)((8 * lane + e) & 255); pay[ae++] = 0; }
			else if (v[e] == 31000) { raw[an++] = (uint8_t)((8 * lane + e) & 255); pay[ae++] = 1; }
		}
		if (lane == 31) {
			raw[base_e + 2 * row + first_half] = H - 2;
			int ac = sh[H + row] >> 20;
			if (v[6] == 30000) c->char_res1[ac++] = (uint16_t)(row * H); else if (v[6] == 31000) c->char_res1[ac++] = (uint16_t)(row * H + 1);
			if (v[7] == 30000) c->char_res1[ac++] = (uint16_t)(row * H + 2); else if (v[7] == 31000) c->char_res1[ac++] = (uint16_t)(row * H + 3);
		}
		if (lane == 63) raw[base_e + 2 * row + total + 1] = H - 2;
	}
	BARRIER();
	if (tid == 0) { c->m->qsetting3_len = (int)tot; c->m->char_res1_len = (int)tc; sh[0] = (int)(te + 2 * H); sh[1] = (int)te; }
	BARRIER();
	poslist_finish_par(c, &c->res6, raw, sh[0], pay, sh[1], 1, tid, shm);
	BARRIER();
	if (!tid) PROF(c, 16);
}
DEV void luma_p4d_par(Ctx *c, int tid, int *sh_counts, uint32_t *sh_z, int16_t *lds, bool dense)
{
	PROF_BEGIN();
	scan_rewrite_list_par<NT>(c, tid, reinterpret_cast<uint8_t *>(lds), sh_counts);   /* Y31 on the list (Y30: the quantiser wrote the symbols in stream order) */
	if (dense) {                                                            /* ... and on the byte stream, where a stage check reads it */
		BARRIER();
		scan_and_rewrite_par(c, tid, sh_counts, sh_z, lds);
	}
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
/* the one cell the chroma passes read behind cll1 (res256): zero, or in the compatibility mode what the stock binary finds there -- both
 * chroma res256 blocks are carved out of the freed 4:2:0 U plane, whose bytes 32768, 32769 follow them.  Written
 * by the phase before the first reader (after the analysis kernel that fills cll1). */
DEV void chroma_ll1_neighbour(Ctx *c, int tid)
{
	if (!tid) c->cll1[Q >> 2] = (int16_t)(c->compat ? (c->pu[32768] | (c->pu[32769] << 8)) : 0);
}
DEV void chroma_p3_par(Ctx *c, int comp, int tid)                     /* :2316-2336 (U), :2629-2648 (V): pointwise */
{
	int16_t *jp = c->cjpeg, *p = c->cproc, *o = c->cll1;
	for (int g = tid; g < Q / 32; g += NT) {                         /* 8 cells of a row per item: 16-byte loads, and the cell behind them */
		const int r = g >> 4, j0 = (g & 15) * 8, e0 = r * H + j0, k0 = r * (H / 2) + j0;
		const uint4 pv = *reinterpret_cast<const uint4 *>(p + e0), ov = *reinterpret_cast<const uint4 *>(o + k0);
		const int p8 = p[e0 + 8], o8 = o[k0 + 8];
		const uint32_t pw[4] = { pv.x, pv.y, pv.z, pv.w }, ow[4] = { ov.x, ov.y, ov.z, ov.w };
		int pc[9], oc[9];
#pragma unroll
		for (int t = 0; t < 8; t++) { pc[t] = (int16_t)(pw[t >> 1] >> (16 * (t & 1))); oc[t] = (int16_t)(ow[t >> 1] >> (16 * (t & 1))); }
		pc[8] = p8; oc[8] = o8;
		uint32_t out[4] = { 0, 0, 0, 0 };
#pragma unroll
		for (int t = 0; t < 8; t++) {
			const int d = pc[t] - oc[t], nx = pc[t + 1] - oc[t + 1];
			int step = 0;
			if (d > 10) step = -6; else if (d > 7) step = -3; else if (d > 4) step = -2; else if (d > 3) step = -1;
			else if (d > 2 && (comp ? nx > 0 : nx >= 0)) step = -1;
			else if (d < -10) step = 6; else if (d < -7) step = 3; else if (d < -4) step = 2; else if (d < -3) step = 1;
			else if (d < -2 && (comp ? nx < 0 : nx <= 0)) step = 1;
			out[t >> 1] |= (uint32_t)(uint16_t)(oc[t] + step) << (16 * (t & 1));
		}
		*reinterpret_cast<uint4 *>(jp + e0) = make_uint4(out[0], out[1], out[2], out[3]);
	}
}
DEV void chroma_p5_par(Ctx *c, int comp, int tid, int16_t *lds, int *sh_misc, bool write_plane)
{
	int16_t *p = c->cproc, *o = c->cll1;
	const int q = c->q;
	PROF_BEGIN();
	if (q >= 18) {
		/* :2372-2427.  The reference walks the LL1 band with an index into cll1 that is NOT reset per row: it runs one
		 * further ahead of the position each time a pair mark is taken at the last column of a row (the skip of the
		 * partner then eats the first increment of the next row).  So row r compares against cll1 shifted by the
		 * number of such events in the rows above it.  Whether a row ends in one depends only on its own shift, so
		 * the shifts are found by iterating "evaluate every row with its current shift, re-count" to a fixed point
		 * (one round when there is no event, which is the rule), then the rows are marked, all in parallel.
		 *
		 * One wavefront per row, lane l owns columns l and l + 64 of the 128-wide band; the pair marks are a walk that
		 * skips the partner, resolved on the row's "pair here and a free detail cell" mask (alt_runs).  Cell (r, 127)
		 * reads column 128 of its row as its right neighbour: that is the HL detail cell of (r, 0), which (r, 0)
		 * may just have marked -- then neither a pair nor the d == -5 rule can hold at (r, 127). */
		const int res_uv = q > 17 ? 4 : 5;
		const int lane = tid & 63, wv = tid >> 6;
		int *shift = reinterpret_cast<int *>(lds), *haz = shift + H / 2;
		/* a row can only end in a pair mark if its last cell is a pair candidate with a free detail cell: where no row's is (the rule), the
		 * evaluation pass is not needed -- every shift is 0 -- and the rows are marked at once (they were loaded twice: 1.3 GB per batch) */
		bool last_cand = false;
		if (tid < H / 2) {
			shift[tid] = 0; haz[tid] = 0;
			const int at = tid * H + H / 2 - 1, ko = tid * (H / 2) + H / 2 - 1;
			const int d = p[at] - o[ko], d1 = p[tid * H + H / 2] - o[ko + 1];
			const bool pair = (d > 3 && d < 7 && d1 > 2 && d1 < 7) || (d < -3 && d > -7 && d1 < -2 && d1 > -8);
			last_cand = pair && (iabs(p[at + H / 2]) < 8 || iabs(p[at + Q / 2]) < 8 || iabs(p[at + Q / 2 + H / 2]) < 8);
		}
		const int first_pass = __syncthreads_or(last_cand) ? 0 : 1;
		for (int pass = first_pass;; pass++) {                     /* pass >= 1 with stable shifts: mark */
			bool mark = false;
			if (pass > 0) {
				int ns = 0;
				if (tid < H / 2) for (int r = 0; r < tid; r++) ns += haz[r];
				const int changed = __syncthreads_or(tid < H / 2 && ns != shift[tid]);
				if (tid < H / 2) shift[tid] = ns;
				BARRIER();
				mark = !changed;
			}
			/* a row's twelve cells per lane are requested while the row before is evaluated (rows are independent; a row's marks land in its own row) */
			int16_t nx[12];
			auto fetch = [&](int r) {
				const int sh = shift[r];
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const int at = r * H + lane + 64 * k, ko = r * (H / 2) + sh + lane + 64 * k;
					nx[6 * k] = p[at]; nx[6 * k + 1] = p[at + H / 2]; nx[6 * k + 2] = p[at + Q / 2]; nx[6 * k + 3] = p[at + Q / 2 + H / 2];
					nx[6 * k + 4] = o[ko]; nx[6 * k + 5] = o[ko + 1];
				}
			};
			fetch(wv);
			for (int r = wv; r < H / 2; r += 4) {
				int pl[2], hl[2], lh[2], hh[2], d[2], d1[2];
#pragma unroll
				for (int k = 0; k < 2; k++) {
					pl[k] = nx[6 * k]; hl[k] = nx[6 * k + 1]; lh[k] = nx[6 * k + 2]; hh[k] = nx[6 * k + 3];
					d[k] = pl[k] - nx[6 * k + 4];
					d1[k] = nx[6 * k + 5];                              /* the reference cell of the right neighbour */
				}
				if (r + 4 < H / 2) fetch(r + 4);
				const int hl_first = __shfl(hl[0], 0);
				for (int k = 0; k < 2; k++) {
					int pn = right_of(pl, k, 2, 1, lane);
					if (k == 1 && lane == 63) pn = hl_first;
					d1[k] = pn - d1[k];
				}
				uint64_t pp[2], pn_[2], fr[2], sg[2], fhl[2], flh[2], m5[2];
				for (int k = 0; k < 2; k++) {
					pp[k] = __ballot(d[k] > 3 && d[k] < 7 && d1[k] > 2 && d1[k] < 7);
					pn_[k] = __ballot(d[k] < -3 && d[k] > -7 && d1[k] < -2 && d1[k] > -8);
					fhl[k] = __ballot(iabs(hl[k]) < 8); flh[k] = __ballot(iabs(lh[k]) < 8);
					fr[k] = fhl[k] | flh[k] | __ballot(iabs(hh[k]) < 8);
					sg[k] = __ballot(iabs(d[k]) > res_uv && (d[k] > 0 || d[k] != -5 || d1[k] < 0));
					m5[k] = __ballot(d[k] == -5);
				}
				uint64_t f[2] = { (pp[0] | pn_[0]) & fr[0], (pp[1] | pn_[1]) & fr[1] };
				if ((fhl[0] & 1) && ((f[0] | sg[0]) & 1)) {              /* (r, 0) marks its HL cell */
					f[1] &= ~(1ull << 63);
					if (m5[1] >> 63) sg[1] &= ~(1ull << 63);
				}
				const M4 fired = alt_runs(M4{ { f[0], f[1], 0, 0 } });
				if (!mark) { if (lane == 0) haz[r] = (int)(fired.w[1] >> 63); continue; }
				const M4 vis = ~up1(fired);
				for (int k = 0; k < 2; k++) {
					const bool pair = (fired.w[k] >> lane) & 1;
					const bool single = !pair && ((vis.w[k] >> lane) & 1) && ((sg[k] >> lane) & 1);
					if (!pair && !single) continue;
					const int16_t code = pair ? (((pp[k] >> lane) & 1) ? 12400 : 12600) : (d[k] > 0 ? 12900 : 13000);
					const int at = r * H + lane + 64 * k;
					if (iabs(hl[k]) < 8) p[at + H / 2] = code;
					else if (iabs(lh[k]) < 8) p[at + Q / 2] = code;
					else if (iabs(hh[k]) < 8) p[at + Q / 2 + H / 2] = code;
				}
			}
			BARRIER();
			if (mark) break;
		}
		if (tid == 0) { int ev = 0; for (int r = 0; r < H / 2; r++) ev += haz[r]; c->m->pad = comp ? c->m->pad + (ev << 8) : ev; }   /* diagnostics: rows that ended in a pair mark */
	}
	BARRIER();
	if (!tid) PROF(c, 27);
	copy_block_par(c->cl2save, H / 2, p, H, H / 2, H / 2, tid);    /* :2431-2439 */
	for (int idx = tid; idx < (H / 4) * (H / 4); idx += NT) lds[idx] = c->cl2save[(idx >> 6) * (H / 2) + (idx & 63)];   /* the LL2 band for the emission below */
	BARRIER();
	if (q <= 11 && tid < 64) {
		/* chroma LL2 smoothing (:2438-2478, :2739-2779): two raster walks over 62 x 62 cells; a cell reads rows r .. r+2 at columns j .. j+2 and
		 * may rewrite (r+1, j+1), which only the next cell of its row and the rows below read.  One wavefront runs the rows skewed by two
		 * columns (lane l on row l, at column t - 2 l in step t: its upper neighbour is two columns ahead, and it reads (r+2, j+1) two steps
		 * before the row below rewrites it) -- 184 steps a walk instead of 3844 cells one after the other on one thread.  Only the emission
		 * reads the band afterwards (it clears it), so the walks run on the LDS copy. */
		const int L = H / 4, lane = tid;
		for (int pass = 0; pass < 2; pass++)
			for (int t = 0; t < (L - 2) + 2 * (L - 3); t++) {
				const int r = lane, j = t - 2 * lane;
				if (r < L - 2 && j >= 0 && j < L - 2) {
					int16_t *v = lds + r * L + j;
					if (!pass) {
						if (iabs(v[1] - v[2 * L + 1]) < 5 && iabs(v[L] - v[L + 2]) < 5 && iabs(v[L + 1] - v[L]) < 7 && iabs(v[1] - v[L + 1]) < 8)
							v[L + 1] = (int16_t)((v[1] + v[2 * L + 1] + v[L] + v[L + 2] + 2) >> 2);
					} else {
						if (iabs(v[2] - v[1]) < 5 && iabs(v[1] - v[0]) < 5 && iabs(v[0] - v[L]) < 5 && iabs(v[2] - v[L + 2]) < 5 &&
						    iabs(v[2 * L + 1] - v[L]) < 5 && iabs(v[L] - v[L + 1]) < 8)
							v[L + 1] = (int16_t)((v[1] + v[2 * L + 1] + v[L] + v[L + 2] + 1) >> 2);
					}
				}
			}
	}
	BARRIER();
	if (!tid) PROF(c, 28);
	{                                                              /* :2489-2525 LL2 emission, 16 consecutive samples per thread */
		/* a sample outside 0..255 goes to the exception list (row, column | sign, magnitude) and repeats the byte
		 * before it in the stream, i.e. the byte of the nearest earlier sample that was in range (sample 0 always is) */
		unsigned *shm = reinterpret_cast<unsigned *>(lds + 4096);
		const int e0 = c->m->exw_len, base = comp ? (Q >> 2) + (Q >> 4) : (Q >> 2);
		int nexc = 0, lastgood = -1;
		for (int u = 0; u < 16; u++) {
			const int idx = tid * 16 + u, s = lds[idx];
			if ((s > 255 || s < 0) && idx > 0) nexc++; else lastgood = idx;
		}
		unsigned total;
		const unsigned eoff = block_exscan((unsigned)nexc, tid, shm, &total);
		int lg = (int)block_exscan_max((unsigned)(lastgood + 1), tid, shm) - 1;
		uint32_t w[4] = { 0, 0, 0, 0 };
		uint8_t *ex = c->exw + e0 + 2 + 3 * eoff;
		for (int u = 0; u < 16; u++) {
			const int idx = tid * 16 + u, s = lds[idx];
			if ((s > 255 || s < 0) && idx > 0) {
				const int mag = s > 255 ? s - 255 : -s;
				*ex++ = (uint8_t)(idx >> 6);
				*ex++ = (uint8_t)((idx & 63) + (s > 255 ? 128 : 0));
				*ex++ = (uint8_t)(mag > 255 ? 255 : mag);
			} else lg = idx;
			int v = lds[lg];
			v = v > 255 ? 255 : (v < 0 ? 0 : v);
			w[u >> 2] |= (uint32_t)(v & 254) << (8 * (u & 3));
		}
		*reinterpret_cast<uint4 *>(c->ll_bytes + base + tid * 16) = make_uint4(w[0], w[1], w[2], w[3]);
		BARRIER();
		if (tid == 0) { c->exw[e0] = 0; c->exw[e0 + 1] = 0; c->m->exw_len = e0 + 2 + 3 * (int)total; }   /* :2489 (U), :2770 (V) */
	}
	for (int idx = tid; idx < (H / 4) * (H / 4); idx += NT) p[(idx >> 6) * H + (idx & 63)] = 0;   /* the emission clears the band */
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
	BARRIER();
	quantise_chroma_par(c, comp, tid, lds, write_plane, write_plane);
	if (!tid) PROF(c, 22);
}


/* ---------------------------------------------------------------- Z2: RLE + VLC packetiser, workgroup-parallel */
/* (compress_pixel.c:53-469).  The reference's walk turns the symbol stream into tokens: a non-zero symbol, a lone
 * zero, or a zero run that is cut into pieces of 254 while more than 255 remain.  Every token boundary is a
 * pure function of the maximal zero run a position sits in, so the stream is cut into 256 slices, each thread
 * tokenises the slice it owns (tokens belong to the slice their first symbol is in), the code lengths are
 * prefix-summed and every thread then drops its bits at its own offset (atomicOr into zeroed words; the
 * reference ORs MSB-first into 32-bit words the same way, :334-345). */
#define PK_SLICE 64
#define PK_SCAN(sh) ((sh)->n2 + 150)                            /* scan scratch that the book entries do not reach (they end at n2[148]) */
#define PK_CHUNK (PK_SLICE * NT)
struct PackShared {
	/* Two pairs of tables are never alive together and share their space (22.5 KB of LDS per workgroup instead of 30: seven packetisers to a
	 * CU instead of five; with the ranks in the staging area and the stale code book in global memory): the histograms are done with when the code words are made from the ranks; the book entries and their weights
	 * live between the histogram sweep and the ranking, the per-thread scan words before and after that (the scans inside that span keep
	 * their scratch behind the part the entries cover: PK_SCAN). */
	union { struct { int hist[256], runs[256]; }; struct { uint32_t code_sym[256], code_run[256]; }; };
	union { struct { unsigned bits[NT], n1[NT], n2[160]; }; struct { unsigned weight[360]; uint16_t entry[600]; }; };   /* n2: scan scratch only */
	uint16_t sorted[360];                                         /* code book entries by rank (the ranks themselves sit in the slice staging area, which is idle between the sweeps) */
	int k, select, zone, top_is_zero, rc;
	unsigned total_bits, total_n1, total_n2;
	/* the reference's `codebook[580]` scratch is ONE stack array for both parts and is never cleared (compress_pixel.c:58): when the
	 * chroma table ends in a run of 128s the collapse at :442-456 reads on into what the luma part left there (its de-interleaved
	 * table; behind that, zeros in the canonical model) */
	int stale_len;                                                /* the bytes themselves: c->cc (written and read by thread 0) */
};

DEV bool book_ok(int v) { return v < 109 ? !(v & 1) : (v == 112 || (v >= 120 && v < 141) || (v >= 144 && !(v & 3))); }

/* tokens whose first symbol lies in [lo, hi); N = stream length (the last symbol can only be swallowed by a run).
 * MODE 0: histogram (every symbol counts, nothing is skipped); MODE 1: count code bits; MODE 2: write code bits */
/* what a slice's walk leaves for the placement behind the prefix sums: its code bits, MSB first, if they fit 128 (they nearly always do: a
 * slice holds a handful of tokens), and its sign symbols as bit masks -- so that a slice is walked once per sweep, not twice */
struct SliceBits { uint32_t b[4]; uint64_t s1m, s2m; };

#define PK_LDS_BYTES ((17 * NT + 1) * 4)                         /* a slot of 17 words per thread (its slice's values: up to 64 bytes; the 17-word pitch keeps the slots of a wavefront in different banks); the ranks and the code books between the sweeps */
/* The luma part comes as a list (round 5): a slice's non-zero map and where its values start (c->nzs, c->voff: stream order, left by Y31;
 * the top three bits of the offset word say how many symbols at the head of the slice belong to a 132..135 code of the slice before).
 * The walk of a slice needs nothing else: zero runs are the gaps between set bits (prev_nz / next_nz where a run crosses the slice's edge),
 * the symbols are the slice's values in order.  They travel with the prefetch, sixteen bytes at a time for as many as the slice holds, and
 * wait in the thread's own LDS slot (nobody else reads it: no barrier), from where the walk takes them byte by byte.  No chunk of the
 * stream is staged, rebuilt or looked at: a slice without a value costs its zero-run arithmetic and nothing else. */
struct PackPreL { uint64_t M; uint32_t off; uint4 v[4]; int prev_nz, next_nz; };
DEV void pack_fetch_list(const Ctx *c, const uint8_t *vals, int ch, int tid, PackPreL *pre, const int *prevnz, const int *nextnz)
{
	const int g = ch * NT + tid;
	pre->prev_nz = prevnz[g]; pre->next_nz = nextnz[g + 1];
	pre->M = c->nzs[g]; pre->off = c->voff[g];
	const int cnt = __builtin_popcountll(pre->M);
	const uint8_t *v = vals + (pre->off & 0x1FFFFFFFu);         /* (a piece may run past the slice's values: behind `vals` lie the guard's zeros) */
#pragma unroll
	for (int k = 0; k < 4; k++) if (cnt > 16 * k) __builtin_memcpy(&pre->v[k], v + 16 * k, 16);
}
/* ... into the thread's LDS slot (17-word pitch: the slots of a wavefront start in different banks); returns the slot */
DEV const uint8_t *pack_stage_list(const PackPreL *pre, int tid, uint32_t *lw)
{
	uint32_t *w = lw + 17 * tid;
	const int cnt = __builtin_popcountll(pre->M);
#pragma unroll
	for (int k = 0; k < 4; k++) if (cnt > 16 * k) { w[4 * k] = pre->v[k].x; w[4 * k + 1] = pre->v[k].y; w[4 * k + 2] = pre->v[k].z; w[4 * k + 3] = pre->v[k].w; }
	return reinterpret_cast<const uint8_t *>(w);
}
/* pack_walk for a slice of the list: nz = its map, skip = symbols at its head that a 132..135 code of the slice before covers, vb = its values */
template <int MODE>
DEV void pack_walk_list(uint64_t nz, int skip, const uint8_t *vb, int N, int lo, int hi, PackShared *sh, uint32_t *words, unsigned bit0, uint8_t *s1, unsigned i1, uint8_t *s2, unsigned i2,
                        unsigned *out_bits, unsigned *out_n1, unsigned *out_n2, int prev_nz, int next_nz, SliceBits *rec = nullptr)
{
	unsigned bits = 0, n1 = 0, n2 = 0;
	uint32_t cur = 0; int w = (int)(bit0 >> 5), fill = (int)(bit0 & 31);
	uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0; uint64_t s1m = 0, s2m = 0;   /* MODE 1 */
	const int select = sh->select;
#define EMIT(entry) do { const uint32_t e_ = (entry), code_ = e_ & 0xFFFFFF; const int len_ = (int)(e_ >> 24); \
		if (MODE == 1) { \
			r0 = (r0 << len_) | (r1 >> (32 - len_)); r1 = (r1 << len_) | (r2 >> (32 - len_)); r2 = (r2 << len_) | (r3 >> (32 - len_)); r3 = (r3 << len_) | code_; \
			bits += (unsigned)len_; \
		} \
		else { fill += len_; if (fill <= 32) cur |= code_ << (32 - fill); \
			else { const int sp_ = fill - 32; atomicOr(&words[w], cur | (code_ >> sp_)); w++; cur = (code_ & ((1u << sp_) - 1)) << (32 - sp_); fill = sp_; } } } while (0)
	const int send = lo + PK_SLICE < N ? lo + PK_SLICE : N;
	int i = lo, vi = 0;
	if (MODE != 0 && skip) { i = lo + skip; vi = __builtin_popcountll(nz & ((1ull << skip) - 1)); }   /* inside the 4 symbols that follow a 132..135 code */
	uint64_t rest = nz >> (i - lo);
	while (i < hi) {
		if (rest & 1) {
			const int px = vb[vi++];
			if (MODE == 0) { atomicAdd(&sh->hist[px], 1); i++; rest >>= 1; continue; }
			if (px == 153 || px == 155) { if (MODE == 2 && i1 + n1 < S_CAP) s1[i1 + n1] = (uint8_t)(px == 155); if (MODE == 1 && px == 155) s1m |= 1ull << n1; n1++; i++; rest >>= 1; continue; }
			if (px == 157 || px == 159) { if (MODE == 2 && i2 + n2 < S_CAP) s2[i2 + n2] = (uint8_t)(px == 159); if (MODE == 1 && px == 159) s2m |= 1ull << n2; n2++; i++; rest >>= 1; continue; }
			EMIT(sh->code_sym[px]);
			if (px > 131 && px < 136) { vi += __builtin_popcount((unsigned)rest & 0x1Eu); i += 5; rest >>= 5; } else { i++; rest >>= 1; }
			continue;
		}
		int a = i, b;                                /* maximal zero run [a, b] around i: inside the slice from the map, outside from the tables */
		if (i == lo && i > 0 && prev_nz != lo - 1) a = prev_nz + 1;
		if (rest) b = i + __builtin_ctzll(rest) - 1;
		else b = send < N ? next_nz - 1 : send - 1;
		const int L = b - a + 1;
		if (L == 1) {
			if (MODE == 0) atomicAdd(&sh->hist[128], 1); else EMIT(sh->code_sym[128]);
		} else if (a == i && L < 255) {              /* the usual run: one piece, begun here */
			if (MODE == 0) atomicAdd(&sh->runs[L], 1);
			else if (L < select) { for (int z = 0; z < L; z++) EMIT(sh->code_sym[128]); }
			else EMIT(sh->code_run[L]);
		} else {
			const int m = L > 255 ? (L - 255 + 253) / 254 : 0;       /* pieces of exactly 254, then the rest; mine are those that start in [i, hi) */
			int k1 = (hi - 1 - a) / 254;
			if (k1 > m) k1 = m;
			for (int k = (i - a + 253) / 254; k <= k1; k++) {
				const int len = k < m ? 254 : L - 254 * m;
				if (MODE == 0) atomicAdd(&sh->runs[len], 1);
				else if (len < select) { for (int z = 0; z < len; z++) EMIT(sh->code_sym[128]); }
				else EMIT(sh->code_run[len]);
			}
		}
		rest = (b + 1 - i) < 64 ? rest >> (b + 1 - i) : 0;
		i = b + 1;
	}
	if (MODE == 2 && fill > 0) atomicOr(&words[w], cur);
	if (MODE == 1) {
		*out_bits = bits; *out_n1 = n1; *out_n2 = n2;
		if (bits <= 128u) {
			const unsigned sh_ = 128u - bits, ws_ = sh_ >> 5, bs_ = sh_ & 31;
			const uint32_t w_[7] = { r0, r1, r2, r3, 0u, 0u, 0u };
			uint32_t x_[5];
#pragma unroll
			for (int k_ = 0; k_ < 5; k_++) x_[k_] = ws_ == 0 ? w_[k_] : ws_ == 1 ? w_[k_ + 1] : ws_ == 2 ? w_[k_ + 2] : ws_ == 3 ? w_[(k_ + 3) < 7 ? k_ + 3 : 6] : 0u;
#pragma unroll
			for (int k_ = 0; k_ < 4; k_++) rec->b[k_] = bs_ ? (x_[k_] << bs_) | (x_[k_ + 1] >> (32 - bs_)) : x_[k_];
		}
		rec->s1m = s1m; rec->s2m = s2m;
	}
#undef EMIT
}
/* Slices are 64 consecutive symbols and a workgroup sweeps the stream in chunks of 256 slices (16 KiB), so the
 * lanes of a wavefront read adjacent cache lines (a thread-per-kilobyte split makes every lane stream its own
 * line and thrashes L1: measured 5 us per symbol). */
/* The chroma part's list as the chroma quantiser leaves it (quantise_chroma_par: [flush][lane][2 slices], a wavefront's values in a region of
 * its own) -> the form the walks read: map and value offset per slice in STREAM order, in the luma part's arrays (c->nzs, c->voff: the luma
 * part is through with them).  A thread takes eight consecutive slices of a flush (sixteen threads a flush): popcounts, a 16-lane prefix
 * sum on top of the flush's base.  Slice (flush F = 4 wavefront + turn, lane, half) holds rows 64 wv + 16 turn + 8 (lane & 1) + 4 half .. + 3 of
 * strip lane >> 1, U and V interleaved: stream slice 64 strip + rows / 4.  The last symbol of the stream is a copy of the one before it
 * (compress_pixel.c:464-465): only whether it is the zero symbol can matter (it is never walked, a run can end in it). */
DEV void pack_chroma_order(Ctx *c, int tid)
{
	const int F = tid >> 4, e0 = 8 * (tid & 15);                    /* entries e0 .. e0 + 7 of the flush's 128 */
	uint64_t m[8];
	unsigned tot = 0;
	for (int k = 0; k < 8; k++) { m[k] = c->cnzq[F * 128 + e0 + k]; tot += (unsigned)__builtin_popcountll(m[k]); }
	unsigned incl = tot;
	for (int o = 1; o < 16; o <<= 1) { const unsigned t_ = (unsigned)__shfl_up((int)incl, o, 16); if ((tid & 15) >= o) incl += t_; }
	unsigned at = c->cfbase[F] + incl - tot;
	const int wv = F >> 2, turn = F & 3;
	for (int k = 0; k < 8; k++) {
		const int e = e0 + k, lane = e >> 1, half = e & 1;
		const int S = 64 * (lane >> 1) + 16 * wv + 4 * turn + 2 * (lane & 1) + half;
		uint64_t M = m[k];
		if (S == 2 * Q / 64 - 1) M = (M & ~(1ull << 63)) | ((M >> 62) & 1ull) << 63;
		c->nzs[S] = M; c->voff[S] = at;
		at += (unsigned)__builtin_popcountll(m[k]);
	}
}

/* A part of the stream comes as a symbol list: the luma part from Y31 (c->nzs, c->voff, c->vals), the chroma part from the chroma quantiser
 * (c->cnzq / c->cvals, put into stream order here) -- the dense byte stream is only ever written for the stage checks. */
DEV void pack_part_par(Ctx *c, int part, PackShared *sh, int tid, int word0, uint32_t *lw /* PK_LDS_BYTES */)
{
	const uint8_t *vals = part ? c->cvals : c->vals;
	const int N = part ? 2 * Q : 4 * Q;
	if (part) { pack_chroma_order(c, tid); __threadfence_block(); BARRIER(); }
	const int S = N - 1, nchunks = (S + PK_CHUNK - 1) / PK_CHUNK;
	const int nsl = nchunks * NT;

	int *prevnz = reinterpret_cast<int *>(c->raw), *nextnz = prevnz + nsl + 8;   /* last non-zero symbol before / first at-or-after a slice */

	PROF_BEGIN();
	sh->hist[tid] = 0; sh->runs[tid] = 0;
	if (tid == 0) { sh->select = part ? 3 : 4; sh->zone = 0; sh->rc = NHW_OK; }
	for (int ch = 0; ch < nchunks; ch++) {                       /* per slice: last / first symbol that is not 128 */
		const int g = ch * NT + tid, lo = g * PK_SLICE;
		const uint64_t nz = c->nzs[g];                           /* bit k: symbol lo + k is not 128 */
		prevnz[g] = nz ? lo + 63 - __builtin_clzll(nz) : -1;
		nextnz[g] = nz ? lo + __builtin_ctzll(nz) : N;
	}
	BARRIER();
	{                                                            /* exclusive prefix max / inclusive suffix min over the slices */
		int *shm = reinterpret_cast<int *>(sh->bits), *shn = reinterpret_cast<int *>(sh->n1);
		int mx = -1, mn = N;
		for (int k = 0; k < nchunks; k++) { const int g = tid * nchunks + k; mx = prevnz[g] > mx ? prevnz[g] : mx; mn = nextnz[g] < mn ? nextnz[g] : mn; }
		shm[tid] = mx; shn[tid] = mn;
		BARRIER();
		{                                                        /* exclusive prefix max of mx, exclusive suffix min of mn over the threads (wave scans) */
			const int a = (int)block_exscan_max((unsigned)(mx + 1), tid, sh->n2) - 1;
			const int b = N - (int)block_exscan_max((unsigned)(N - shn[NT - 1 - tid]), tid, sh->n2);
			BARRIER();
			shm[tid] = a; shn[NT - 1 - tid] = b;
		}
		BARRIER();
		int run = shm[tid];
		for (int k = 0; k < nchunks; k++) { const int g = tid * nchunks + k; const int v = prevnz[g]; prevnz[g] = run; run = v > run ? v : run; }
		run = shn[tid];
		for (int k = nchunks - 1; k >= 0; k--) { const int g = tid * nchunks + k; const int v = nextnz[g]; run = v < run ? v : run; nextnz[g] = run; }
		if (tid == 0) nextnz[nsl] = N;
	}
	BARRIER();
	/* The histogram (:81-127) needs no walk of the symbols: every symbol that is not the zero symbol counts once, whatever surrounds it -- a flat
	 * histogram over the value list, sixteen bytes a thread and turn, every thread equally loaded (as a walk, a slice of the level-2 quadrant
	 * kept its lane busy for 64 tokens while the other lanes of the wavefront waited) -- and the zero runs are the gaps of the map: a slice's
	 * gaps, with what prev_nz / next_nz say about the ones that cross its edges.  The values in the list that no symbol owns any more are
	 * left out: the first and the last four luma symbols that Y31 cleared (the head of the first flush, the tail of the last one), the
	 * chroma part's last symbol (never walked: compress_pixel.c:81's loop ends one short). */
	{
		auto count16 = [&](const uint8_t *base, unsigned at, unsigned lo_, unsigned hi_) {   /* the bytes of [at, at + 16) that lie in [lo_, hi_) */
			const uint4 v = *reinterpret_cast<const uint4 *>(base + at);
			const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int k = 0; k < 16; k++) { const unsigned pos = at + k; if (pos >= lo_ && pos < hi_) atomicAdd(&sh->hist[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu], 1); }
		};
		if (!part) {
			const unsigned lo_ = c->voff[0] & 0x1FFFFFFFu, hi_ = (c->voff[SL_SLICES - 1] & 0x1FFFFFFFu) + (unsigned)__builtin_popcountll(c->nzs[SL_SLICES - 1]);
			for (unsigned at = (lo_ & ~15u) + 16u * tid; at < hi_; at += 16u * NT) count16(vals, at, lo_, hi_);
		} else {
			for (int w4 = 0; w4 < 4; w4++) {
				const unsigned hi_ = 32768u * w4 + c->cfbase[16 + w4];
				for (unsigned at = 32768u * w4 + 16u * tid; at < hi_; at += 16u * NT) count16(vals, at, 32768u * w4, hi_);
			}
		}
	}
	for (int ch = 0; ch < nchunks; ch++) {
		const int g = ch * NT + tid, lo = g * PK_SLICE, hi = lo + PK_SLICE < S ? lo + PK_SLICE : S;
		if (lo >= S) continue;
		const uint64_t nz = c->nzs[g];
		const int prev_nz = prevnz[g], next_nz = nextnz[g + 1];
		const int send = lo + PK_SLICE < N ? lo + PK_SLICE : N;
		if (part && g == nsl - 1 && (c->cnzq[(15 * 64 + 63) * 2 + 1] >> 63))       /* the chroma part's last symbol as the quantiser left it (slice 2047 = the last entry of the last flush): not walked, not counted */
			atomicSub(&sh->hist[vals[(c->voff[g] & 0x1FFFFFFFu) + (unsigned)__builtin_popcountll(c->cnzq[(15 * 64 + 63) * 2 + 1]) - 1u]], 1);
		int i = lo;
		uint64_t rest = nz;
		while (i < hi) {
			if (rest & 1) {                                          /* a run of symbols: counted above */
				const int n1 = ~rest ? __builtin_ctzll(~rest) : 64;
				i += n1; rest = n1 < 64 ? rest >> n1 : 0;
				continue;
			}
			int a = i, b;                                            /* maximal zero run [a, b] around i (pack_walk_list) */
			if (i == lo && i > 0 && prev_nz != lo - 1) a = prev_nz + 1;
			if (rest) b = i + __builtin_ctzll(rest) - 1;
			else b = send < N ? next_nz - 1 : send - 1;
			const int L = b - a + 1;
			if (L == 1) atomicAdd(&sh->hist[128], 1);
			else if (a == i && L < 255) atomicAdd(&sh->runs[L], 1);
			else {
				const int m = L > 255 ? (L - 255 + 253) / 254 : 0;
				int k1 = (hi - 1 - a) / 254;
				if (k1 > m) k1 = m;
				for (int k = (i - a + 253) / 254; k <= k1; k++) atomicAdd(&sh->runs[k < m ? 254 : L - 254 * m], 1);
			}
			rest = (b + 1 - i) < 64 ? rest >> (b + 1 - i) : 0;
			i = b + 1;
		}
	}
	BARRIER();
	if (!tid) PROF(c, 23);
	for (;;) {                                                   /* L_RATIO (:128-236): short runs become plain zeros until the book fits */
		const int select = sh->select, j = tid;
		unsigned tot, totr, tots;
		(void)block_exscan((j >= 2 && j < select && sh->runs[j] > 0) ? (unsigned)(j * sh->runs[j]) : 0u, tid, PK_SCAN(sh), &tot);
		BARRIER();
		if (j >= 2 && j < select) sh->runs[j] = 0;
		if (tid == 0) sh->hist[128] = (int)((sh->hist[128] > 0 ? (unsigned)sh->hist[128] : 0u) + tot);
		BARRIER();
		const bool fr = j >= select && sh->runs[j] > 0, fs = book_ok(j) && sh->hist[j] > 0;
		const unsigned orr = block_exscan(fr, tid, PK_SCAN(sh), &totr);
		const unsigned os = block_exscan(fs, tid, PK_SCAN(sh), &tots);
		if (fr) { sh->entry[orr] = (uint16_t)((j << 8) | 128); sh->weight[orr] = (unsigned)sh->runs[j]; }
		if (fs && totr + os < 600) { sh->entry[totr + os] = (uint16_t)((1 << 8) | j); sh->weight[totr + os < 360 ? totr + os : 359] = (unsigned)sh->hist[j]; }
		const int k = (int)(totr + tots);
		BARRIER();
		if (k <= 354) { if (tid == 0) sh->k = k; break; }
		if (tid == 0) { sh->select = select + 1; if (select + 1 >= 100) sh->rc = NHW_E_CODEBOOK; }
		BARRIER();
		if (sh->rc) break;
	}
	BARRIER();
	if (sh->rc) return;
	uint16_t *rank_sym = reinterpret_cast<uint16_t *>(lw), *rank_run = rank_sym + 256;
	{                                                            /* stable descending rank == the reference's bubble sort (:238-252) */
		const int k = sh->k;
		for (int e = tid; e < k; e += NT) {
			const unsigned w = sh->weight[e];
			int rank = 0;
			for (int j = 0; j < k; j++) rank += (sh->weight[j] > w) || (sh->weight[j] == w && j < e);
			const uint16_t en = sh->entry[e];
			if ((en >> 8) == 1) rank_sym[en & 0xFF] = (uint16_t)rank; else rank_run[en >> 8] = (uint16_t)rank;
			sh->sorted[rank] = en;
		}
	}
	BARRIER();
	if (tid == 0) {
		const int k = sh->k, select = sh->select;
		sh->top_is_zero = (sh->sorted[0] == ((1 << 8) | 128));
		if (part == 0 && !sh->top_is_zero && k > 290) sh->rc = NHW_E_CODEBOOK;      /* :269-271 */
		if (part == 1 && select != 4 && k > 290) sh->rc = NHW_E_CODEBOOK;
		sh->zone = (part == 0 && select == 4 && sh->top_is_zero);
	}
	BARRIER();
	if (sh->rc) return;
	for (int v = 0; v < 2; v++) {                                /* rank -> code word (the 64 ranks from 110 use the short escape when the zone is on, :300-330) */
		int pos = v ? rank_run[tid] : rank_sym[tid];
		uint32_t e;
		if (pos >= 110 && pos < 174 && sh->zone) e = (15u << 24) | (uint32_t)((1 << 6) | (pos - 110));
		else { if (pos >= 174 && sh->zone) pos -= 64; e = k_vlc[pos < 290 ? pos : 0]; }
		if (v) sh->code_run[tid] = e; else sh->code_sym[tid] = e;
	}
	BARRIER();
	if (!tid) PROF(c, 24);
	/* bit counts and bits in one sweep: while a chunk is in LDS every slice is walked twice -- once to count its code bits
	 * and sign symbols, and, after a workgroup prefix sum on top of the running totals, once to drop its bits at their
	 * offset.  The words a chunk will touch are zeroed just ahead of it (the chunk before may share its first word). */
	uint32_t *words = c->packet + word0;
	unsigned base_bits = 0, base_n1 = 0, base_n2 = 0;
	int zeroed = 0;                                              /* words [0, zeroed) are cleared or already carry bits */
	PackPreL prel;
	pack_fetch_list(c, vals, 0, tid, &prel, prevnz, nextnz);
	for (int ch = 0; ch < nchunks; ch++) {
		const int lo = ch * PK_CHUNK + tid * PK_SLICE, hi = lo + PK_SLICE < S ? lo + PK_SLICE : S;
		unsigned bb = 0, x1 = 0, x2 = 0, tb, tn;
		const int my_prev = prel.prev_nz, my_next = prel.next_nz;
		const uint64_t my_nz = prel.M;
		const int my_skip = (int)(prel.off >> 29);
		const uint8_t *dl = pack_stage_list(&prel, tid, lw);
		if (ch + 1 < nchunks) pack_fetch_list(c, vals, ch + 1, tid, &prel, prevnz, nextnz);
		SliceBits rec;
		if (lo < S) pack_walk_list<1>(my_nz, my_skip, dl, N, lo, hi, sh, nullptr, 0, nullptr, 0, nullptr, 0, &bb, &x1, &x2, my_prev, my_next, &rec);
		const unsigned ob = block_exscan(bb, tid, sh->bits, &tb);
		const unsigned on = block_exscan(x1 | (x2 << 16), tid, sh->bits, &tn);
		const int last = tb ? (int)((base_bits + tb - 1) >> 5) : zeroed - 1;
		/* the packet holds 80000 words per image, like the reference's calloc(80000) (compress_pixel.c:64), which never checks: a
		 * stream that would not fit ends THIS image with a status instead of running into its neighbours' words */
		if (word0 + last >= 80000) { if (tid == 0) sh->rc = NHW_E_SPACE; BARRIER(); return; }
		for (int w = zeroed + tid; w <= last; w += NT) words[w] = 0;
		BARRIER();
		if (lo < S) {
			if (bb <= 128u) {                                        /* the recorded bits, moved to their place (the words were zeroed above; neighbours share words) */
				const unsigned o = base_bits + ob, r = o & 31;
				const int w0 = (int)(o >> 5), nw = bb ? (int)((r + bb + 31) >> 5) : 0;
				const uint32_t v[5] = { rec.b[0] >> r, r ? (rec.b[0] << (32 - r)) | (rec.b[1] >> r) : rec.b[1], r ? (rec.b[1] << (32 - r)) | (rec.b[2] >> r) : rec.b[2],
				                        r ? (rec.b[2] << (32 - r)) | (rec.b[3] >> r) : rec.b[3], r ? rec.b[3] << (32 - r) : 0u };
#pragma unroll
				for (int k = 0; k < 5; k++) if (k < nw && v[k]) atomicOr(&words[w0 + k], v[k]);
				const unsigned a1 = base_n1 + (on & 0xFFFF), a2 = base_n2 + (on >> 16);
				for (unsigned z = 0; z < x1; z++) if (a1 + z < S_CAP) c->s1[a1 + z] = (uint8_t)((rec.s1m >> z) & 1);
				for (unsigned z = 0; z < x2; z++) if (a2 + z < S_CAP) c->s2[a2 + z] = (uint8_t)((rec.s2m >> z) & 1);
			}
			else pack_walk_list<2>(my_nz, my_skip, dl, N, lo, hi, sh, words, base_bits + ob, c->s1, base_n1 + (on & 0xFFFF), c->s2, base_n2 + (on >> 16), nullptr, nullptr, nullptr, my_prev, my_next);
		}
		base_bits += tb; base_n1 += tn & 0xFFFF; base_n2 += tn >> 16;
		if (last + 1 > zeroed) zeroed = last + 1;
	}
	if (tid == 0) { if (!zeroed) words[0] = 0; sh->total_bits = base_bits; sh->total_n1 = base_n1; sh->total_n2 = base_n2; }
	BARRIER();
	const int nwords = sh->total_bits ? (int)((sh->total_bits - 1) >> 5) + 1 : 1;
	if (!tid) PROF(c, 26);

	/* the code book (:400-459): the ranked entries as bytes (a run entry takes two), de-interleaved (even positions, then odd ones), runs of
	 * the marker byte (3 in book 1, 128 in book 2) collapsed into (marker, count).  Until round 3 one thread walked this -- 0.1 of the
	 * packetiser's 0.58 ms per image with 255 threads waiting; now three prefix sums.  Book 2's collapse reads on behind its table into what
	 * book 1 left in the reference's one scratch array (c->cc) when the table ends in a run of 128s. */
	const uint16_t *sorted = sh->sorted;
	uint8_t *book = reinterpret_cast<uint8_t *>(lw), *tmp_book = book + 1024;
	int *book_len = reinterpret_cast<int *>(book + 2048);
	if (part == 0) {
		const int n1 = (int)sh->total_n1, n2 = (int)sh->total_n2;
		const int b1 = (n1 >> 3) + 1, b2 = (n2 >> 3) + 1;        /* sign bits, 8 per byte (:370-398) */
		for (int t = tid; t < b1; t += NT) { int v = 0; for (int u = 0; u < 8; u++) v = (v << 1) | ((8 * t + u < n1 && 8 * t + u < S_CAP ? c->s1[8 * t + u] : 0) & 1); c->sel_word1[t] = (uint8_t)v; }
		for (int t = tid; t < b2; t += NT) { int v = 0; for (int u = 0; u < 8; u++) v = (v << 1) | ((8 * t + u < n2 && 8 * t + u < S_CAP ? c->s2[8 * t + u] : 0) & 1); c->sel_word2[t] = (uint8_t)v; }
		if (tid == 0) {
			c->m->size_data1 = nwords;
			c->m->wavelet_type = (sh->select > 4 || !sh->top_is_zero) ? 4 : 0;       /* :367-368 */
			c->m->select1 = b1; c->m->select2 = b2;
		}
	} else if (tid == 0) c->m->size_data2 = word0 + nwords;
	{
		const int k = sh->k, marker = part ? 128 : 3;
		unsigned *scr = sh->bits;                                  /* scan scratch (the write sweep is done with it) */
		unsigned tot;
		int e;
		{                                                          /* entries 2 tid, 2 tid + 1 -> bytes (:400-411, :431-438) */
			int sz[2];
			for (int h = 0; h < 2; h++) { const int i = 2 * tid + h; sz[h] = i < k ? ((sorted[i] >> 8) == 1 ? 1 : 2) : 0; }
			unsigned at = block_exscan((unsigned)(sz[0] + sz[1]), tid, scr, &tot);
			for (int h = 0; h < 2; h++) {
				const int i = 2 * tid + h;
				if (i >= k) break;
				const int en = sorted[i];
				if ((en >> 8) == 1) book[at++] = (uint8_t)(part ? ((en & 0xFF) | 1) : (en & 0xFF));
				else { book[at++] = (uint8_t)(part ? (en & 0xFF) : 3); book[at++] = (uint8_t)(en >> 8); }
			}
			e = (int)tot;
		}
		BARRIER();
		const int half = (e + 1) >> 1;
		for (int j = tid; j < e; j += NT) tmp_book[j] = j < half ? book[2 * j] : book[2 * (j - half) + 1];
		const int stale = part ? sh->stale_len : 0;
		if (part) for (int i = e + tid; i < stale; i += NT) tmp_book[i] = c->cc[i];
		if (tid == 0) tmp_book[e > stale ? e : stale] = 0;
		BARRIER();
		if (!part) { for (int j = tid; j < e; j += NT) c->cc[j] = tmp_book[j]; if (tid == 0) sh->stale_len = e; }
		{                                                          /* the collapse (:413-424, :440-459): bytes 3 tid .. 3 tid + 2 */
			int osz[3];
			for (int h = 0; h < 3; h++) {
				const int j = 3 * tid + h;
				osz[h] = 0;
				if (j >= e) continue;
				const bool is_m = tmp_book[j] == marker;
				osz[h] = !is_m ? 1 : ((j == 0 || tmp_book[j - 1] != marker) ? 2 : 0);
			}
			unsigned at = block_exscan((unsigned)(osz[0] + osz[1] + osz[2]), tid, scr, &tot);
			BARRIER();                                             /* (the de-interleave's source is overwritten now) */
			for (int h = 0; h < 3; h++) {
				const int j = 3 * tid + h;
				if (osz[h] == 1) book[at++] = tmp_book[j];
				else if (osz[h] == 2) { int len = 0; while (tmp_book[j + len] == marker) len++; book[at++] = (uint8_t)marker; book[at++] = (uint8_t)len; }
			}
			if (tid == 0) {
				if (part) { c->m->tree_end = e; c->m->size_book2 = (int)tot; } else c->m->size_book1 = (int)tot;
				*book_len = (int)tot;
			}
		}
	}
	BARRIER();
	{
		uint8_t *dst = part ? c->book2 : c->book1;
		for (int i = tid; i < *book_len; i += NT) dst[i] = book[i];
	}
	BARRIER();
}

/* container (nhw_encoder.c:3112-3218): every thread walks the same field list, the byte copies are shared */
struct ParSink { uint8_t *p; size_t cap, n; int tid; };
DEV void pput(ParSink *s, const void *d, size_t n) { if (s->n + n <= s->cap) for (size_t i = s->tid; i < n; i += NT) s->p[s->n + i] = ((const uint8_t *)d)[i]; s->n += n; }
DEV void pput16(ParSink *s, unsigned v) { if (s->tid == 0 && s->n + 2 <= s->cap) { s->p[s->n] = (uint8_t)v; s->p[s->n + 1] = (uint8_t)(v >> 8); } s->n += 2; }
DEV void pput32(ParSink *s, uint32_t v) { if (s->tid == 0 && s->n + 4 <= s->cap) { for (int k = 0; k < 4; k++) s->p[s->n + k] = (uint8_t)(v >> (8 * k)); } s->n += 4; }

DEV size_t container_par(Ctx *c, uint8_t *out, size_t cap, int tid)
{
	ParSink s = { out, cap, 0, tid };
	const int q = c->q;
	const NhwMeta *m = c->m;
	if (tid == 0) { out[0] = (uint8_t)(m->res_high + m->wavelet_type); out[1] = (uint8_t)q; }
	s.n = 2;
	pput16(&s, (unsigned)m->size_book1); pput16(&s, (unsigned)m->size_book2);
	pput32(&s, (uint32_t)m->size_data1); pput32(&s, (uint32_t)m->size_data2);
	pput16(&s, (unsigned)m->tree_end); pput16(&s, (unsigned)m->exw_len);
	if (q > 12) pput16(&s, (unsigned)m->r1.list_len);
	if (q >= 19) { pput16(&s, (unsigned)m->r3.list_len); pput16(&s, (unsigned)m->r3.bits_len); }
	if (q > 17) pput16(&s, (unsigned)m->res4_len);
	if (q > 12) pput16(&s, (unsigned)m->r1.bits_len);
	if (q >= 21) { pput16(&s, (unsigned)m->r5.list_len); pput16(&s, (unsigned)m->r5.bits_len); }
	if (q > 21) { pput32(&s, (uint32_t)m->r6.list_len); pput16(&s, (unsigned)m->r6.bits_len); pput16(&s, (unsigned)m->char_res1_len); }
	if (q > 22) pput16(&s, (unsigned)m->qsetting3_len);
	pput16(&s, (unsigned)m->select1); pput16(&s, (unsigned)m->select2);
	if (q > 15) pput16(&s, (unsigned)m->ll_word_len);
	pput16(&s, (unsigned)m->ch_res_len);

	pput(&s, c->book1, (size_t)m->size_book1); pput(&s, c->book2, (size_t)m->size_book2);
	pput(&s, c->exw, (size_t)m->exw_len);
	if (q > 12) { pput(&s, c->res1.list, (size_t)m->r1.list_len); pput(&s, c->res1.bits, (size_t)m->r1.bits_len); pput(&s, c->res1.word, (size_t)m->r1.word_len); }
	if (q > 17) pput(&s, c->res4, (size_t)m->res4_len);
	if (q >= 19) { pput(&s, c->res3.list, (size_t)m->r3.list_len); pput(&s, c->res3.bits, (size_t)m->r3.bits_len); pput(&s, c->res3.word, (size_t)m->r3.word_len); }
	if (q >= 21) { pput(&s, c->res5.list, (size_t)m->r5.list_len); pput(&s, c->res5.bits, (size_t)m->r5.bits_len); pput(&s, c->res5.word, (size_t)m->r5.word_len); }
	if (q > 21) {
		pput(&s, c->res6.list, (size_t)m->r6.list_len); pput(&s, c->res6.bits, (size_t)m->r6.bits_len); pput(&s, c->res6.word, (size_t)m->r6.word_len);
		pput(&s, c->char_res1, (size_t)m->char_res1_len * 2);            /* little-endian u16, as the reference's fwrite */
	}
	if (q > 22) pput(&s, c->qsetting3, (size_t)m->qsetting3_len * 4);
	pput(&s, c->sel_word1, (size_t)m->select1); pput(&s, c->sel_word2, (size_t)m->select2);
	if (q > 15) { pput(&s, c->res_u64, 2 * H); pput(&s, c->res_v64, 2 * H); pput(&s, c->ll_word, (size_t)m->ll_word_len); }
	pput(&s, c->ch_res, (size_t)m->ch_res_len);
	pput(&s, c->packet, (size_t)m->size_data2 * 4);
	return s.n <= cap ? s.n : 0;
}

/* Z1, Z2 and the container */
DEV void final_phase_par(Ctx *c, uint8_t *out, size_t cap, uint32_t *size, int32_t *status, PackShared *sh, int tid, uint32_t *lw)
{
	PROF_BEGIN();
	/* (the reference's sentinel behind the luma part, compress_pixel.c:66, and its copy of the last chroma symbol, :464-465, were writes into the
	 * byte stream; both parts come as lists now: the walks never reach the sentinel, the copy is made on the chroma part's map) */
	pack_part_par(c, 0, sh, tid, 0, lw);
	if (sh->rc) { if (tid == 0) { *size = 0; *status = sh->rc; } return; }
	BARRIER();
	pack_part_par(c, 1, sh, tid, c->m->size_data1, lw);
	if (sh->rc) { if (tid == 0) { *size = 0; *status = sh->rc; } return; }
	if (!tid) PROF(c, 19);
	const size_t n = container_par(c, out, cap, tid);
	if (tid == 0) { *size = (uint32_t)n; *status = n ? NHW_OK : -3; }
	if (!tid) PROF(c, 20);
}

} // namespace nhw
#endif
