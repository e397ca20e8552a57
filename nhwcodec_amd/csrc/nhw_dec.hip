/*
 * nhw_dec.hip -- the NHW decoder (BASELINE config 5) for gfx950: kernels, workspace
 * and the decode half of the C ABI (include/nhw_hip.h).
 *
 * What the reference does per file (decoder/nhw_decoder.c:54 decode_image, :1478 parse_file,
 * decoder/compress_pixel.c:49,446, decoder/wavelet_filterbank.c:52,237, decoder/nhw_decoder_cli.c:108) is re-cut
 * into batch-wide launches, every launch running one stage of all n files:
 *
 *   k_dec_parse    header + section table, packets copied to aligned words; the four byte-serial side streams
 *                  (LL2 DPCM bytes, the three/four position lists) are walked by one lane each, side by side
 *   k_dec_vlc      the prefix-code walk, one wavefront per stream (luma, chroma): a speculative parallel parse and a
 *                  short-state chain for the placement rules; it leaves lists of (position, value), stream order
 *   k_dec_expand   pattern symbols -> coefficients, the +-1 nudge of the HH band, LL2 samples, odd-LL tags,
 *                  exception samples: one wavefront per image, the rows streaming through LDS in order
 *   k_dec_luma_l2  level 2 of the luma on one LDS residency of the block: shrink, synthesis both ways, residual lists
 *   k_dec_chroma   a chroma plane on one LDS residency: built from the value list, LL2 + exception samples, level 2, pair
 *                  corrections, level 1
 *   k_dec_marks, k_dec_sharpen
 *   k_dec_final    level 1 of the luma both ways, corrections, smoothing, chroma up-sampling, colour matrix -> BGR24
 *
 * Everything is int16/uint8 arithmetic; the only floating point is the colour matrix (compiled with
 * -ffp-contract=off like the rest of the library).  No stage falls back to the host.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "../../include/nhw_hip.h"

#define DW 512
#define DH 256
#define DQ 65536
#define DEV __device__ __forceinline__

namespace {

/* ---------------------------------------------------------------------------------------------- workspace
 * structure of arrays over the batch: buffer b of image i at base + off[b] + i * size[b] */
enum {
	D_META, D_LL, D_SPARE, D_P1, D_P3, D_P5, D_P6, D_MARKS, D_A, D_B, D_CA, D_CB, D_CU, D_SEG, D_NZG, D_COUNT
};
enum { P16_CAP = 65536 + 64, P6_CAP = 131072 + 64, PK_WORDS = 98304 /* sanity bound on the packet words of a file (the encoder's buffer holds 80000) */ };
const size_t k_dec_bytes[D_COUNT] = {
	/* META */ 512, /* LL */ 24832, /* SPARE */ 1024, /* P1 */ P16_CAP * 2, /* P3 */ P16_CAP * 2, /* P5 */ P16_CAP * 2, /* P6 */ (size_t)P6_CAP * 4,
	/* MARKS */ 2 * DQ, /* A */ 8 * DQ + 8192, /* B: luma value list */ 16 * DQ + 8192, /* CA */ 2 * (2 * DQ + 4096), /* CB: chroma value list */ 8 * DQ + 8192, /* CU */ 2 * DQ,
	/* SEG */ 5120, /* NZG: which 16-byte groups of plane A's rows are in memory (k_dec_expand), a 64-bit word a row */ 4096
};

struct DecMeta {
	int status, q, res_high;
	int book1_len, book2_len, data1, data2, tree_end, exw_len;
	int res1_len, res1_bits, res3_len, res3_bits, res4_len, res5_len, res5_bits, res6_len, res6_bits, char_res1_len, qs3_len;
	int select1, select2, ll_word_len, ch_res_len;
	uint32_t o_book1, o_book2, o_exw, o_res1, o_res1_bit, o_res1_word, o_res4, o_res3, o_res3_bit, o_res3_word;
	uint32_t o_res5, o_res5_bit, o_res5_word, o_res6, o_res6_bit, o_res6_word, o_char, o_qs3;
	uint32_t o_sel1, o_sel2, o_u64, o_v64, o_llword, o_chres, o_packet1, o_packet2;
	int carry;          /* the left-over `count` that reaches nhw_decoder.c:571 */
	int nmarks;
	int size;
};

struct DecWs {
	uint8_t *base;
	size_t off[D_COUNT];
	int n;
	const uint8_t *blob;       /* device arena holding the .nhw files */
	const uint64_t *blob_off;  /* n offsets into it */
	const uint32_t *blob_len;  /* n lengths */
	int dense;                 /* a stage check is going to read plane A: every group of it is written (production leaves out the all-zero groups of the level-1 detail bands) */
	template <typename T> __host__ __device__ T *buf(int b, int img) const { return (T *)(base + off[b] + (size_t)img * k_dec_bytes_dev(b)); }
	__host__ __device__ static size_t k_dec_bytes_dev(int b)
	{
		switch (b) {
		case D_META: return 512; case D_LL: return 24832; case D_SPARE: return 1024;
		case D_P1: case D_P3: case D_P5: return P16_CAP * 2; case D_P6: return (size_t)P6_CAP * 4;
		case D_MARKS: return 2 * DQ; case D_A: return 8 * DQ + 8192; case D_B: return 16 * DQ + 8192;
		case D_CA: return 2 * (2 * DQ + 4096); case D_CB: return 8 * DQ + 8192; case D_SEG: return 5120; case D_NZG: return 4096; default: return 2 * DQ;
		}
	}
};

/* plane A starts 4096 bytes into its buffer (the reference writes one cell in front of a plane in a corner case) */
DEV int16_t *plane_a(const DecWs &ws, int img) { return ws.buf<int16_t>(D_A, img) + 2048; }
/* D_B / D_CB: what the prefix-code walk found, as a list in stream order -- (value << 18) | position in the stream, one word per value that
 * is not part of a zero run (at most one per cell, plus a few words of slack) -- and D_SEG: the index of the first entry at or behind the
 * start of each of the 128 strips (2048 symbols) of the luma stream, the total behind them (k_dec_expand follows every strip with a
 * cursor); of the chroma list only the total, at word SEG_CHROMA + 128 (k_dec_chroma goes through the whole list) */
#define SEG_CHROMA 1040
#define ENT_POS(e) ((int)((e) & 0x3FFFFu))
#define ENT_VAL(e) ((int)(e) >> 18)
#define ENT_MAKE(pos, v) (((uint32_t)(v) << 18) | (uint32_t)(pos))
DEV uint16_t *mark_rows(const DecWs &ws, int img) { return ws.buf<uint16_t>(D_SPARE, img) + 8; }   /* behind the verdict words */
/* the prefix-code walk's verdict on a file: a word per stream (each written by that stream's workgroup only), 0 = fine */
DEV int walk_verdict(const DecWs &ws, int img) { const int *v = ws.buf<int>(D_SPARE, img); return v[0] ? v[0] : v[1]; }
DEV int16_t *plane_ca(const DecWs &ws, int img, int comp) { return ws.buf<int16_t>(D_CA, img) + 1024 + (size_t)comp * (DQ + 2048); }

DEV int iabs(int v) { return v < 0 ? -v : v; }
DEV int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
DEV int bit_of(const uint8_t *bytes, int nbytes, int k) { return (k >> 3) < nbytes ? (bytes[k >> 3] >> (7 - (k & 7))) & 1 : 0; }

/* Wave-wide scans on the DPP network (an instruction each step; the shuffle forms go through the LDS crossbar: address arithmetic, a
 * ds_bpermute and its latency per step, and these scans sit on the serial path of every 64-byte step of the entropy kernels).
 * Inclusive, lane 0 first: four shifts inside the rows of 16 lanes, then the total of the row before into rows 1 and 3 and the total of
 * the first two rows into rows 2 and 3.  A lane that has no source reads 0, which every operator below takes as "nothing on my left". */
#define DPP_STEPS(S) S(0x111, 0xF, true) S(0x112, 0xF, true) S(0x114, 0xF, true) S(0x118, 0xF, true) S(0x142, 0xA, false) S(0x143, 0xC, false)
DEV int wscan_add(int v)
{
#define S_(CTRL, RM, BC) v += __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xF, BC);
	DPP_STEPS(S_)
#undef S_
	return v;
}
/* segmented sum: a flagged lane starts over; (flag, sum) -> flag: some lane at or before me is flagged, sum: from the nearest one on */
DEV void wscan_seg(int &flag, int &sum)
{
#define S_(CTRL, RM, BC) { const int lf = __builtin_amdgcn_update_dpp(0, flag, CTRL, RM, 0xF, BC), ls = __builtin_amdgcn_update_dpp(0, sum, CTRL, RM, 0xF, BC); if (!flag) { sum += ls; flag = lf; } }
	DPP_STEPS(S_)
#undef S_
}
/* the value of the last lane at or before me that has one (has = 0: none so far, val undefined) */
DEV void wscan_last(int &has, int &val)
{
#define S_(CTRL, RM, BC) { const int lh = __builtin_amdgcn_update_dpp(0, has, CTRL, RM, 0xF, BC), lv = __builtin_amdgcn_update_dpp(0, val, CTRL, RM, 0xF, BC); if (!has) { has = lh; val = lv; } }
	DPP_STEPS(S_)
#undef S_
}
/* Markers that take the byte behind them as a payload, whatever that byte is: inside a run of marker bytes every second one is a marker,
 * counted from the run's first byte (unless that one is the payload handed over by the block before).  T: the marker-valued bytes of a
 * 64-byte block; returns the real markers.  The markers of the runs that begin on an even position are the run's bytes on even positions,
 * likewise odd: an addition at the run's first bit carries through the run and picks it out. */
DEV uint64_t alt_starts(uint64_t T, bool pending)
{
	const uint64_t Tm = pending ? T & ~1ull : T, rs = Tm & ~(Tm << 1), ev = 0x5555555555555555ull;
	return (Tm & ~(Tm + (rs & ev)) & ev) | (Tm & ~(Tm + (rs & ~ev)) & ~ev);
}
DEV int from_left(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false); }   /* lane l: v of lane l - 1; lane 0: `first` */
DEV int last_lane(int v) { return __builtin_amdgcn_readlane(v, 63); }

/* ---------------------------------------------------------------------------------------------- parse (d1)
 * parse_file, nhw_decoder.c:1497-1659 */
struct Rd { const uint8_t *p; uint32_t n, at; int bad; };
DEV unsigned rd8(Rd &s) { if (s.at + 1 > s.n) { s.bad = 1; return 0; } return s.p[s.at++]; }
DEV unsigned rd16(Rd &s) { const unsigned a = rd8(s); return a | (rd8(s) << 8); }
DEV unsigned rd32(Rd &s) { const unsigned a = rd16(s); return a | (rd16(s) << 16); }
DEV uint32_t take(Rd &s, uint32_t n) { const uint32_t r = s.at; if (s.at + n > s.n || s.at + n < s.at) { s.bad = 1; return 0; } s.at += n; return r; }

#define HDR_STAGE 128                /* bytes of the file brought into LDS for parse_header (it reads fewer than 64) */
#define BOOK_STAGE 2048              /* bytes of a packed code book staged in LDS; a longer one (no encoder writes that) is read from memory from there on */
DEV void stage_hdr(uint8_t *dst, const uint8_t *f, uint64_t flen, int t) { dst[t] = (uint64_t)t < flen ? f[t] : 0; }
DEV void parse_header(const uint8_t *d, uint32_t len, DecMeta *m)
{
	Rd s = { d, len, 0, 0 };
	m->res_high = (int)rd8(s);
	const int q = m->q = (int)rd8(s);
	if (s.bad || m->res_high > 6 || q < 1 || q > 23) { m->status = NHW_E_FORMAT; return; }
	m->book1_len = (int)rd16(s); m->book2_len = (int)rd16(s);
	m->data1 = (int)rd32(s); m->data2 = (int)rd32(s);
	m->tree_end = (int)rd16(s); m->exw_len = (int)rd16(s);
	if (q > 12) m->res1_len = (int)rd16(s);
	if (q >= 19) { m->res3_len = (int)rd16(s); m->res3_bits = (int)rd16(s); }
	if (q > 17) m->res4_len = (int)rd16(s);
	if (q > 12) m->res1_bits = (int)rd16(s);
	if (q >= 21) { m->res5_len = (int)rd16(s); m->res5_bits = (int)rd16(s); }
	if (q > 21) { m->res6_len = (int)rd32(s); m->res6_bits = (int)rd16(s); m->char_res1_len = (int)rd16(s); }
	if (q > 22) m->qs3_len = (int)rd16(s);
	m->select1 = (int)rd16(s); m->select2 = (int)rd16(s);
	if (q > 15) m->ll_word_len = (int)rd16(s);
	m->ch_res_len = (int)rd16(s);
	if (s.bad || m->data1 < 0 || m->data2 < m->data1 || m->data2 > PK_WORDS - 8 || m->res6_len < 0) { m->status = NHW_E_FORMAT; return; }
	m->o_book1 = take(s, (uint32_t)m->book1_len); m->o_book2 = take(s, (uint32_t)m->book2_len);
	m->o_exw = take(s, (uint32_t)m->exw_len);
	if (q > 12) { m->o_res1 = take(s, (uint32_t)m->res1_len); m->o_res1_bit = take(s, (uint32_t)m->res1_bits); m->o_res1_word = take(s, (uint32_t)m->res1_bits); }
	if (q > 17) m->o_res4 = take(s, (uint32_t)m->res4_len);
	if (q >= 19) { m->o_res3 = take(s, (uint32_t)m->res3_len); m->o_res3_bit = take(s, (uint32_t)m->res3_bits); m->o_res3_word = take(s, (uint32_t)m->res3_bits * 2); }
	if (q >= 21) { m->o_res5 = take(s, (uint32_t)m->res5_len); m->o_res5_bit = take(s, (uint32_t)m->res5_bits); m->o_res5_word = take(s, (uint32_t)m->res5_bits); }
	if (q > 21) {
		m->o_res6 = take(s, (uint32_t)m->res6_len); m->o_res6_bit = take(s, (uint32_t)m->res6_bits); m->o_res6_word = take(s, (uint32_t)m->res6_bits);
		m->o_char = take(s, (uint32_t)m->char_res1_len * 2);
	}
	if (q > 22) m->o_qs3 = take(s, (uint32_t)m->qs3_len * 4);
	m->o_sel1 = take(s, (uint32_t)m->select1); m->o_sel2 = take(s, (uint32_t)m->select2);
	if (q > 15) { m->o_u64 = take(s, 2 * DH); m->o_v64 = take(s, 2 * DH); m->o_llword = take(s, (uint32_t)m->ll_word_len); }
	m->o_chres = take(s, (uint32_t)m->ch_res_len);
	m->o_packet1 = take(s, (uint32_t)m->data1 * 4);
	m->o_packet2 = take(s, (uint32_t)(m->data2 - m->data1) * 4);
	if (s.bad || m->res1_bits * 8 > P16_CAP - 64 || m->res3_bits * 8 > P16_CAP - 64 || m->res5_bits * 8 > P16_CAP - 64 || m->res6_bits * 8 > P6_CAP - 64)
		m->status = NHW_E_FORMAT;
}

/* A wavefront's view of a byte string, 64 bytes at a time (lane l holds byte i0 + l): the block in hand, and the next one already
 * on its way from memory while this one is being worked on.  Bytes behind the end read 0. */
struct BlockReader {
	const uint8_t *g; int len, i0, cur, nxt, n2, n3;                /* the block in hand and the three behind it: a 64-byte step is shorter than a memory round trip */
	DEV int ld(int i) const { return i < len ? (int)g[i] : 0; }
	DEV void init(const uint8_t *g_, int len_, int start, int lane) { g = g_; len = len_; i0 = start; cur = ld(i0 + lane); nxt = ld(i0 + 64 + lane); n2 = ld(i0 + 128 + lane); n3 = ld(i0 + 192 + lane); }
	DEV void advance(int lane) { i0 += 64; cur = nxt; nxt = n2; n2 = n3; n3 = ld(i0 + 192 + lane); }
	DEV int after(int lane) const { const int a = __builtin_amdgcn_update_dpp(0, cur, 0x130 /* wave_shl:1 */, 0xF, 0xF, false), b = __builtin_amdgcn_readlane(nxt, 0); return lane < 63 ? a : b; }   /* byte i0 + l + 1 */
};
/* bytes that a walk consumes a few at a time, at most 64 a step, by index: a 256-byte window in registers that slides 64 at a time */
struct WindowReader {
	const uint8_t *g; int len, base, w0, w1, w2, w3;
	DEV int ld(int i) const { return i < len ? (int)g[i] : 0; }
	DEV void init(const uint8_t *g_, int len_, int lane) { g = g_; len = len_; base = 0; w0 = ld(lane); w1 = ld(64 + lane); w2 = ld(128 + lane); w3 = ld(192 + lane); }
	DEV int at(int i) const { const int r = i - base, a = __shfl(w0, r & 63), b = __shfl(w1, r & 63); return r < 64 ? a : b; }   /* base <= i < base + 128 */
	DEV void consumed_up_to(int a, int lane) { if (a - base >= 64) { base += 64; w0 = w1; w1 = w2; w2 = w3; w3 = ld(base + 192 + lane); } }   /* a: first index still wanted */
};

/* LL2 samples (res_comp), nhw_decoder.c:1661-2026; unsigned char arithmetic.
 *
 * The reference expands the DPCM bytes one at a time, each sample relative to the one before.  The dependence has two parts,
 * and both are scans: (1) which bytes start a token -- every byte does, except the one after a 64..127 byte that itself
 * starts a token (its payload), an alternation inside runs of such bytes; (2) the sample values -- a token either sets an
 * absolute value or adds a few differences, so the value after each token is a segmented sum (mod 256).  One wavefront takes
 * 64 bytes per step: token starts from a ballot, sample offsets from a prefix sum of the tokens' lengths, values from a
 * segmented scan; then every lane writes its own token's samples.  The luma part ends at the first token that would start at
 * sample 16384; that byte is the first chroma sample, verbatim, and the chroma bytes (one-byte tokens) follow. */
struct LlTok { int copies, n, d0, d1, d2, abs_n, a0, a1; };   /* `copies` repeats of the value before, then n differences; or abs_n absolute samples */

DEV LlTok ll_token_luma(int b, int d, int mode, bool fine_on, int fine)
{
	LlTok t = { 0, 0, 0, 0, 0, 0, 0, 0 };
	if (b >= 128) { if (fine_on) { t.abs_n = 2; t.a0 = fine; t.a1 = (b - 128) << 1; } else { t.abs_n = 1; t.a0 = (b - 128) << 1; } }
	else if (b >= 64) { const int c = b - 64; t.n = 3; t.d0 = (((c >> 1) & 31) << 1) - 32; t.d1 = ((((c & 1) << 3) | (d >> 5)) << 1) - 16; t.d2 = ((d & 31) << 1) - 32; }
	else if (mode == 1) {
		if (b < 32) { t.copies = ((b >> 2) & 7) + 2; const int k = b & 3; if (k) { t.n = 1; t.d0 = k == 1 ? 2 : k == 2 ? -2 : 0; } }
		else { const int c = b - 32; t.n = 2; t.d0 = ((c >> 3) << 1) - 4; t.d1 = ((c & 7) << 1) - 8; }
	}
	else if (mode == 2) t.copies = (b & 63) + 2;
	else {
		if (b < 16) {
			t.copies = ((b >> 3) & 1) + 2;
			switch (b & 7) {
			case 1: t.n = 1; t.d0 = 2; break;
			case 2: t.n = 2; t.d0 = 2; t.d1 = -2; break;
			case 3: t.n = 2; t.d0 = 2; t.d1 = 0; break;
			case 4: t.n = 2; t.d0 = -2; t.d1 = 2; break;
			case 5: t.n = 2; t.d0 = -2; t.d1 = 0; break;
			case 6: t.n = 1; t.d0 = -2; break;
			case 7: t.n = 1; t.d0 = 4; break;
			default: break;
			}
		}
		else if (b < 32) { t.n = 2; t.d0 = b >= 24 ? 4 : 2; t.d1 = ((b & 7) << 1) - 8; }
		else { const int c = b - 32; t.n = 2; t.d0 = ((c >> 3) << 1) - 6; t.d1 = ((c & 7) << 1) - 8; }
	}
	return t;
}
DEV LlTok ll_token_chroma(int b)
{
	LlTok t = { 0, 0, 0, 0, 0, 0, 0, 0 };
	if (b >= 192) {
		const int c = b - 192, pr = c >> 2, k = c & 3;
		t.n = 3;
		t.d0 = pr == 2 || pr == 4 || pr == 5 ? 4 : pr == 3 || pr == 6 || pr == 7 ? -4 : 0;
		t.d1 = pr == 0 || pr == 4 || pr == 6 ? 4 : pr == 1 || pr == 5 || pr == 7 ? -4 : 0;
		t.d2 = k == 0 ? 0 : k == 1 ? 4 : k == 2 ? -4 : 8;
	}
	else if (b >= 128) { t.abs_n = 1; t.a0 = (b - 128) << 2; }
	else if (b >= 64) {
		const int run = (b >> 3) & 7;
		if (run == 7) t.copies = (b & 7) + 7 + 2;
		else {
			t.copies = run + 2;
			switch (b & 7) {
			case 1: t.n = 1; t.d0 = 4; break;
			case 2: t.n = 2; t.d0 = 4; t.d1 = -4; break;
			case 3: t.n = 3; t.d0 = 4; t.d1 = -4; t.d2 = 0; break;
			case 4: t.n = 3; t.d0 = -4; t.d1 = 4; t.d2 = 0; break;
			case 5: t.n = 2; t.d0 = -4; t.d1 = 4; break;
			case 6: t.n = 1; t.d0 = -4; break;
			case 7: t.n = 1; t.d0 = 8; break;
			default: break;
			}
		}
	}
	else { t.n = 2; t.d0 = ((b >> 3) << 2) - 16; t.d1 = ((b & 7) << 2) - 16; }
	return t;
}

/* one 64-byte step: lanes that start a token carry `t`; returns through j / prev the running sample offset and value */
DEV void ll_emit_block(const LlTok &t, bool is_tok, int lane, int &j, int &prev, int limit, uint8_t *ll)
{
	const int cnt = is_tok ? (t.abs_n ? t.abs_n : t.copies + t.n) : 0;
	int off = cnt;                                                  /* inclusive prefix of the sample counts */
	off = wscan_add(off);
	/* value after each token: (absolute?, value) pairs under "a later absolute token wins, otherwise differences add up" */
	int isabs = is_tok && t.abs_n ? 1 : 0;
	int val = !is_tok ? 0 : t.abs_n ? (t.abs_n == 2 ? t.a1 : t.a0) : (t.d0 + t.d1 + t.d2);
	wscan_seg(isabs, val);
	const int after = (isabs ? val : prev + val) & 255;             /* value after my token */
	const int before = from_left(after, prev);
	if (is_tok) {
		int at = j + off - cnt;
		if (t.abs_n) { if (at < limit) ll[at] = (uint8_t)t.a0; if (t.abs_n == 2 && at + 1 < limit) ll[at + 1] = (uint8_t)t.a1; }
		else {
			for (int k = 0; k < t.copies; k++, at++) if (at < limit) ll[at] = (uint8_t)before;
			int v = before;
			if (t.n > 0) { v += t.d0; if (at < limit) ll[at] = (uint8_t)v; at++; }
			if (t.n > 1) { v += t.d1; if (at < limit) ll[at] = (uint8_t)v; at++; }
			if (t.n > 2) { v += t.d2; if (at < limit) ll[at] = (uint8_t)v; at++; }
		}
	}
	j += last_lane(off);
	prev = last_lane(after);
}

/* whole wavefront */
DEV void ll_expand_wave(const uint8_t *code_g, int code_len, const uint8_t *fine, const DecMeta *m, uint8_t *ll, int lane)
{
	BlockReader code;
	const int mode = (m->res_high & 3) == 3 ? 0 : (m->res_high & 3), q = m->q;
	int prev = code_len > 0 ? code_g[0] : 0, j = 1, a = 0;
	code.init(code_g, code_len, 1, lane);
	WindowReader fines;
	fines.init(fine, q > 15 ? m->ll_word_len : 0, lane);
	bool pending = false;                                           /* byte i0 is the payload of a token that started in the block before */
	if (!lane) ll[0] = (uint8_t)prev;
	int split = -1;                                                 /* index of the byte that is sample 16384 */
	while (split < 0) {
		const int b = code.cur, d = code.after(lane), i0 = code.i0;
		const uint64_t T = __ballot(b >= 64 && b < 128);
		/* a 64..127 byte that is not itself a payload starts a two-byte token */
		const uint64_t starters = alt_starts(T, pending);
		const uint64_t pay = (starters << 1) | (pending ? 1ull : 0ull);
		const bool pend_out = (starters >> 63) != 0;
		const bool start = !((pay >> lane) & 1ull);
		const bool verb = start && b >= 128;
		/* fine bytes: one per verbatim token, in order */
		int vpre = verb ? 1 : 0;
		vpre = wscan_add(vpre);
		const int fidx = a + vpre - 1;
		const int fw = fines.at(verb ? fidx : a);                   /* (every lane takes part in the shuffle) */
		const int fv = (verb && q > 15 && fidx < m->ll_word_len) ? fw : 0;
		LlTok t = ll_token_luma(b, d, mode, q > 15, fv);
		/* where would my token start? the first token at or past sample 16384 is not a token but the chroma seed */
		const int cnt = start ? (t.abs_n ? t.abs_n : t.copies + t.n) : 0;
		int off = cnt;
		off = wscan_add(off);
		const uint64_t over = __ballot(start && j + off - cnt >= DQ / 4);
		bool live = start;
		if (over) { const int ls = __builtin_ctzll(over); split = i0 + ls; live = start && lane < ls; }
		ll_emit_block(t, live, lane, j, prev, DQ / 4, ll);
		a += last_lane(vpre);                                     /* (past the split this is no longer used) */
		fines.consumed_up_to(a, lane);
		pending = pend_out;
		code.advance(lane);
	}
	/* chroma: sample 16384 verbatim (:1878), then one-byte tokens (:1882-1979) */
	prev = split < code_len ? code_g[split] : 0; j = DQ / 4 + 1;
	if (!lane) ll[DQ / 4] = (uint8_t)prev;
	code.init(code_g, code_len, split + 1, lane);
	while (j < DQ / 4 + DQ / 8) {
		const int b = code.cur;
		LlTok t = ll_token_chroma(b);
		const int cnt = t.abs_n ? t.abs_n : t.copies + t.n;
		int off = cnt;
		off = wscan_add(off);
		const bool live = j + off - cnt < DQ / 4 + DQ / 8;         /* the walk stops at the first token that would start past the end */
		ll_emit_block(t, live, lane, j, prev, DQ / 4 + DQ / 8, ll);
		code.advance(lane);
	}
}

/* Position lists (nhw_decoder.c:93-137 and its three copies): list bytes -> (row | column) entries.
 *
 * Byte by byte the reference keeps a row counter, the column of the entry it wrote last and "the byte before was a row mark".
 * A byte below 127 is a column (absolute: it starts a segment; the row advances if the column went backwards); 127 is a row
 * mark; a byte from 128 carries two column steps from the entry before, and a step that would pass column 253 ends the row
 * instead -- after that, and after a row mark, further step bytes only advance the row, until the next absolute column.  (The
 * reference does that by patching list bytes to 127 as it goes.)
 * Per segment that is a running sum with one cut-off point, so a wavefront takes 64 bytes per step: columns from a segmented
 * sum, the cut-off from a "last non-neutral wins" scan of {column byte: alive, mark / overflow: dead}, the column written last
 * from another such scan, rows and entry offsets from prefix sums. */
struct PlCarry { int last, p127, row, n; };
DEV int scan_add(int v, int) { return wscan_add(v); }
/* inclusive "the last lane at or before me that has one" scan: has/val in, the winning val (or `none` if no lane has one) out */
DEV int scan_last(bool has, int val, int none, int lane)
{
	int h = has ? 1 : 0, v = val;
	wscan_last(h, v);
	return h ? v : none;
}
template <typename T>
DEV int poslist_wave(const uint8_t *list, int len, T *pos, int cap, int row_step, bool mask16, int lane)
{
	PlCarry c = { 0, 0, 0, 0 };
	BlockReader b;
	b.init(list, len, 0, lane);
	for (int i0 = 0; i0 < len; i0 += 64, b.advance(lane)) {
		const int i = i0 + lane;
		const bool valid = i < len;
		const int v = b.cur;
		const bool isM = valid && v == 127, isA = valid && !isM && (v < 127 || i == 0), isR = valid && !isM && !isA;
		const int d1 = ((v - 128) >> 4) << 1, d2 = (v & 15) << 1, D = isR ? d1 + d2 : 0;
		const int anew = (v << 1) & 255;
		/* running column assuming no cut-off: segmented sum, a column byte restarts it */
		int flag = isA ? 1 : 0, sum = isA ? anew : D;
		wscan_seg(flag, sum);
		const int run = flag ? sum : c.last + sum;
		const int c1 = run - D + d1, c2 = run;
		const bool ovf = isR && c2 >= 254;
		/* "the byte before was a row mark" after each byte: column byte -> 0, mark or overflow -> 1, otherwise unchanged */
		const int st_after = scan_last(isA || isM || ovf, isA ? 0 : 1, c.p127, lane);
		const int p127b = from_left(st_after, c.p127);
		const bool alive = isR && !p127b;
		const int nemit = isA ? 1 : alive ? (ovf ? (c1 < 254 ? 1 : 0) : 2) : 0;
		/* column of the entry written last, after each byte */
		const bool writes = isA || (alive && nemit > 0);
		const int last_after = scan_last(writes, isA ? anew : (ovf ? c1 : c2), c.last, lane);
		const int lastb = from_left(last_after, c.last);
		int inc = 0;
		if (isA) inc = (i != 0 && (v << 1) < lastb && !p127b) ? row_step : 0;
		else if (isM) inc = row_step;
		else if (isR) inc = !alive ? 2 * row_step : ovf ? (c1 >= 254 ? 2 * row_step : row_step) : 0;
		const int rsum = scan_add(inc, lane), esum = scan_add(nemit, lane);
		const int row_before = c.row + rsum - inc;
		int at = c.n + esum - nemit;
		if (isA) { const unsigned val = (unsigned)((v << 1) + row_before + inc); if (at < cap) pos[at] = (T)(mask16 ? (val & 0xFFFFu) : val); }
		else if (nemit > 0) {
			const unsigned v1 = (unsigned)(c1 + row_before), v2 = (unsigned)(c2 + row_before);
			if (at < cap) pos[at] = (T)(mask16 ? (v1 & 0xFFFFu) : v1);
			if (nemit > 1 && at + 1 < cap) pos[at + 1] = (T)(mask16 ? (v2 & 0xFFFFu) : v2);
		}
		c.last = last_lane(last_after); c.p127 = last_lane(st_after);
		c.row += last_lane(rsum); c.n += last_lane(esum);
	}
	return c.n;
}

/* One wavefront per workgroup and side stream -- role 0: the LL2 DPCM bytes (+ the chroma bit planes on top of them, + the file's
 * header record for the kernels behind); roles 1..3: the position lists res1, res3, res5 + res6 with their bit planes -- the long role
 * first for the whole batch.  (As four wavefronts of one workgroup the three list walks waited at a barrier for the LL2 walk.) */
__global__ __launch_bounds__(64) void k_dec_parse(DecWs ws)
{
	__shared__ DecMeta sm;
	__shared__ uint8_t hdr[HDR_STAGE];
	const int role = (int)blockIdx.x / ws.n, img = (int)blockIdx.x - role * ws.n, lane = threadIdx.x;
	const uint8_t *f = ws.blob + ws.blob_off[img];
	const uint64_t flen = ws.blob_len[img];
	for (int k = lane; k < HDR_STAGE; k += 64) stage_hdr(hdr, f, flen, k);   /* the header's bytes by the wavefront, then one lane reads them from LDS */
	__syncthreads();
	if (!lane) {
		memset(&sm, 0, sizeof sm);
		sm.size = (int)flen;
		if (flen > (1u << 24)) sm.status = NHW_E_FORMAT; else parse_header(hdr, (uint32_t)flen, &sm);
	}
	__syncthreads();
	DecMeta *gm = ws.buf<DecMeta>(D_META, img);
	if (sm.status) { if (!lane && !role) *gm = sm; return; }
	const int q = sm.q;

	/* a list: (row, col) entries by 64-byte scans, then the low bits from the bit plane; entries the list did not reach are 0 + their bit (the
	 * reference's calloc) */
#define LIST16(P, RES_OFF, RES_LEN, BITS, BIT_OFF) do { \
		const int c_ = poslist_wave(f + (RES_OFF), (RES_LEN), (P), (BITS) * 8, 256, true, lane); \
		__syncthreads(); \
		for (int k = lane; k < (BITS) * 8; k += 64) (P)[k] = (uint16_t)((k < c_ ? (P)[k] : 0) + bit_of(f + (BIT_OFF), (BITS), k)); \
	} while (0)
	if (role == 1) { if (q > 12) LIST16(ws.buf<uint16_t>(D_P1, img), sm.o_res1, sm.res1_len, sm.res1_bits, sm.o_res1_bit); return; }
	if (role == 2) { if (q >= 19) LIST16(ws.buf<uint16_t>(D_P3, img), sm.o_res3, sm.res3_len, sm.res3_bits, sm.o_res3_bit); return; }
	if (role == 3) {
		if (q >= 21) LIST16(ws.buf<uint16_t>(D_P5, img), sm.o_res5, sm.res5_len, sm.res5_bits, sm.o_res5_bit);
		if (q > 21) {
			uint32_t *p6 = ws.buf<uint32_t>(D_P6, img);
			const int c6 = poslist_wave(f + sm.o_res6, sm.res6_len, p6, sm.res6_bits * 8, 256, false, lane);
			__syncthreads();
			for (int k = lane; k < sm.res6_bits * 8; k += 64) p6[k] = (k < c6 ? p6[k] : 0u) + (uint32_t)bit_of(f + sm.o_res6_bit, sm.res6_bits, k);
		}
		return;
	}
#undef LIST16
	uint8_t *ll = ws.buf<uint8_t>(D_LL, img);
	ll_expand_wave(f + sm.o_chres, sm.ch_res_len, f + sm.o_llword, &sm, ll, lane);
	__syncthreads();
	/* bit-1 planes of the chroma LL2 samples (:1983-2026) */
	if (q > 15) {
		for (int k = lane; k < 2 * DH * 8; k += 64) {
			ll[DQ / 4 + k] = (uint8_t)(ll[DQ / 4 + k] + (bit_of(f + sm.o_u64, 2 * DH, k) << 1));
			ll[DQ / 4 + DQ / 16 + k] = (uint8_t)(ll[DQ / 4 + DQ / 16 + k] + (bit_of(f + sm.o_v64, 2 * DH, k) << 1));
		}
	}
	if (!lane) {
		/* the reference's `count` as decode_image reaches :571 */
		int carry = 4 * DQ;
		if (q > 12) carry = sm.res1_bits > 0 ? (sm.res1_bits - 1) * 8 : 0;
		if (q >= 21) carry = sm.res5_bits > 0 ? (sm.res5_bits - 1) * 8 : 0;
		if (q > 21) carry = sm.res6_bits > 0 ? (sm.res6_bits - 1) * 8 : 0;
		if (q >= 19) carry = sm.res3_bits > 0 ? (sm.res3_bits * 2 - 2) * 4 : 0;
		sm.carry = carry;
		*gm = sm;
	}
}

/* ---------------------------------------------------------------------------------------------- VLC (d2)
 * the 290-word prefix code (encoder/tree.h:58-140; decoder/tables.h:59,125 hold it as two lookup tables) */
struct VlcRun { uint32_t first; uint8_t len; uint16_t count; };
__constant__ VlcRun k_runs[26] = {
	{0x0000,2,1},{0x0002,3,1},{0x0004,3,1},{0x000a,4,2},{0x0006,4,2},{0x0018,5,3},{0x0036,6,2},{0x0070,7,2},
	{0x00e8,8,12},{0x01c8,9,8},{0x01e8,9,8},{0x03e8,10,8},{0x03e4,10,4},{0x07c0,11,2},{0x07e0,11,2},
	{0x07f0,11,16},{0x07e8,11,8},{0x0f88,12,8},{0x0fc8,12,8},{0x1f08,13,4},{0x3f10,14,8},
	{0x1f0c0,17,64},{0x1f8c0,17,46},{0x3f1dc,18,12},{0x7e3d0,19,38},{0xfc7ec,20,20}
};

/* Two-level code table.  lut[v], v = the next 8 bits: (len << 8) | rank for code words of up to 8 bits; 0x8000 | s when v is one of
 * the sixteen 8-bit prefixes of longer words, s selecting a 64-entry sub-table indexed by the following 6 bits: (len << 10) | rank for
 * words of 9..14 bits, 0 for the 17..20-bit tail (ranks >= 110), which is searched by its {first, length, count} runs. */
DEV void vlc_fill_lut(uint16_t *lut, uint16_t *lut2, int lane)
{
	for (int v = lane; v < 256; v += 64) {
		unsigned e = 0; int rank = 0;
		for (int r = 0; r < 9; r++) {
			const unsigned c = (unsigned)v >> (8 - k_runs[r].len);
			if (c >= k_runs[r].first && c < k_runs[r].first + k_runs[r].count) { e = ((unsigned)k_runs[r].len << 8) | (unsigned)(rank + (int)(c - k_runs[r].first)); break; }
			rank += k_runs[r].count;
		}
		if (!e) e = 0x8000u | (unsigned)(v < 0xe8 ? v - 0xe4 : v < 0xf8 ? 4 + v - 0xf4 : 8 + v - 0xf8);   /* the 16 prefixes no short word covers: e4-e7, f4-f7, f8-ff */
		lut[v] = (uint16_t)e;
	}
	for (int idx = lane; idx < 16 * 64; idx += 64) {
		const int s = idx >> 6;
		const unsigned pre = (unsigned)(s < 4 ? 0xe4 + s : s < 8 ? 0xf4 + s - 4 : 0xf8 + s - 8);
		const unsigned v14 = (pre << 6) | (unsigned)(idx & 63);
		unsigned e = 0; int rank = 26;
		for (int r = 9; r < 21; r++) {
			const unsigned c = v14 >> (14 - k_runs[r].len);
			if (c >= k_runs[r].first && c < k_runs[r].first + k_runs[r].count) { e = ((unsigned)k_runs[r].len << 10) | (unsigned)(rank + (int)(c - k_runs[r].first)); break; }
			rank += k_runs[r].count;
		}
		lut2[idx] = (uint16_t)e;
	}
}
/* books, compress_pixel.c:86-117 / :456-478: entry = (run length << 8) | symbol, 354 entries (ranks 0..353); one lane, scr = 1440 bytes of LDS scratch */
DEV int build_book_small(const uint8_t *raw_g, int raw_len, const uint8_t *raw_s, int staged, bool chroma, int tree_end, uint16_t *book, uint8_t *scr)
{
	uint8_t *flat = scr, *inter = scr + 720;
	const int rep = chroma ? 128 : 3;
	int e = 0, n = 0;
	for (int i = 0; i < 720; i++) { flat[i] = 0; inter[i] = 0; }
#define RAW(i) ((i) < staged ? raw_s[i] : raw_g[i])
	for (int i = 0; i < raw_len && e < 708; i++) {                 /* (once 708 bytes are out nothing more is kept) */
		const int b = RAW(i);
		if (b == rep) { const int cnt = i + 1 < raw_len ? RAW(i + 1) : 0; for (int j = 0; j < cnt && e < 708; j++) flat[e++] = (uint8_t)rep; i++; }
		else flat[e++] = (uint8_t)b;
	}
#undef RAW
	if (chroma) e = tree_end;
	if (e > 708) e = 708;
	int j = 0;
	for (int i = 0; i < e; i += 2) inter[i] = flat[j++];
	for (int i = 1; i < e; i += 2) inter[i] = flat[j++];
	for (int i = 0; i < e; i++) {
		uint16_t v;
		if (!chroma) {
			if (inter[i] == 3) { v = (uint16_t)((inter[i + 1] << 8) | 128); i++; }
			else v = (uint16_t)(256 | inter[i]);
		} else {
			if (!(inter[i] & 1)) { v = (uint16_t)((inter[i + 1] << 8) | inter[i]); i++; }
			else v = (uint16_t)(256 | (inter[i] & 0xfe));
		}
		if (n < 354) book[n] = v;
		n++;
	}
	for (int i = n; i < 354; i++) book[i] = 0;
	return n;
}

/* the same by a whole wavefront, the packed book in LDS (raw_len <= BOOK_STAGE): both byte-serial parts are "a marker takes the next byte
 * as its payload" walks (the 3 / 128 repeat marker of the packing, the two-byte entries of the book), 64 bytes a step */
DEV void build_book_wave(const uint8_t *raw, int raw_len, bool chroma, int tree_end, uint16_t *book, uint8_t *scr, int lane)
{
	uint8_t *flat = scr, *inter = scr + 720;
	const int rep = chroma ? 128 : 3;
	for (int k = lane; k < 1440; k += 64) scr[k] = 0;
	__syncthreads();
	int e = 0;
	bool pend = false;
	for (int i0 = 0; i0 < raw_len && e < 708; i0 += 64) {
		const int i = i0 + lane;
		const bool valid = i < raw_len;
		const int b = valid ? raw[i] : -1, cnt = i + 1 < raw_len ? raw[i + 1] : 0;
		const uint64_t st = alt_starts(__ballot(b == rep), pend), pay = (st << 1) | (pend ? 1ull : 0ull);
		const bool is_st = (st >> lane) & 1ull, is_pay = (pay >> lane) & 1ull;
		const int len = !valid || is_pay ? 0 : is_st ? cnt : 1;
		const int inc = wscan_add(len), at = e + inc - len;
		if (is_st) { for (int z = 0; z < cnt && at + z < 708; z++) flat[at + z] = (uint8_t)rep; }
		else if (len && at < 708) flat[at] = (uint8_t)b;
		e = min(708, e + last_lane(inc));
		pend = (st >> 63) != 0;
	}
	if (chroma) e = tree_end;
	if (e > 708) e = 708;
	if (e < 0) e = 0;
	__syncthreads();
	for (int k = lane; k < e; k += 64) inter[k] = flat[(k & 1) ? ((e + 1) >> 1) + (k >> 1) : (k >> 1)];
	__syncthreads();
	int n = 0;
	pend = false;
	for (int i0 = 0; i0 < e; i0 += 64) {
		const int i = i0 + lane;
		const bool valid = i < e;
		const int b = valid ? inter[i] : 1, nx = valid ? inter[i + 1] : 0;      /* (1: a marker of neither kind) */
		const uint64_t st = alt_starts(__ballot(valid && (chroma ? !(b & 1) : b == 3)), pend), pay = (st << 1) | (pend ? 1ull : 0ull);
		const bool is_st = (st >> lane) & 1ull, entry = valid && !((pay >> lane) & 1ull);
		const int v = is_st ? (chroma ? (nx << 8) | b : (nx << 8) | 128) : (chroma ? 256 | (b & 0xfe) : 256 | b);
		const int inc = wscan_add(entry ? 1 : 0), idx = n + inc - (entry ? 1 : 0);
		if (entry && idx < 354) book[idx] = (uint16_t)v;
		n += last_lane(inc);
		pend = (st >> 63) != 0;
	}
	for (int r = (n < 354 ? n : 354) + lane; r < 354; r += 64) book[r] = 0;
}

DEV int extra_level(int word)          /* decoder/tables.h:51 */
{
	const int off = word & 7;
	if (word < 10 || word > 108 || (off != 2 && off != 4 && off != 6)) return 0;
	const int n = ((word >> 3) - 1) * 3 + (off >> 1);
	return n <= 19 ? n : -(n - 19);
}
DEV int plain_level(int word)
{
	const int x = word < 110 ? extra_level(word) : 0;
	if (x > 0) return 123 + (x << 3);
	if (x < 0) return (x << 3) - 123;
	return word > 128 ? word - 125 : word - 131;
}

/* The prefix-code walk, in parallel (compress_pixel.c:49-444 luma, :446-640 chroma).
 *
 * Read serially, a stream costs about a microsecond per symbol on this machine: a lone lane of a lone wavefront, some 120
 * dependent instructions per symbol.  Both halves of the walk are really short-memory chains:
 *
 *  A. Parsing.  Where a code word starts depends on all words before it -- but a prefix code re-synchronises within a few
 *     words.  A chunk of 4096 bits is cut into 64 spans of 64 bits, one per lane.  Every lane parses its span from a guessed
 *     start and reports where it left the span; that is the next lane's true start, so lanes whose start moved parse again,
 *     until nothing moves (lane 0's start is known; in practice two or three rounds).  A last pass stores the ranks of the
 *     chunk in order (prefix sum of the per-span counts).
 *  B. Placing.  Luma only: the encoder folded isolated +-8 values into the zero runs around them, and the decoder re-inserts one
 *     in front of a run when a handful of conditions on the recent past hold (mem, mem2, nhw_ac1, the five values before, the
 *     distance to the last 254-run: compress_pixel.c:280-320).  That past is a 14-bit state; a lane takes one symbol, computes
 *     its transition from a guessed entry state, passes the exit state to the right, and the chain is iterated to its fixed
 *     point as in A.  Stream positions and the indices into the two sign-bit strings are then prefix sums.
 *
 * One wavefront (workgroup) per stream.  Values leave as a list in stream order (k_dec_expand / k_dec_chroma
 * puts them in place). */
enum { VCH_WORDS = 128, VCH_SYMS = 2048 + 64 };

DEV unsigned peek20(const uint32_t *cw, int rel)
{
	const int w = rel >> 5, sh = rel & 31;
	const uint64_t v = ((uint64_t)cw[w] << 32) | cw[w + 1];
	return (unsigned)((v << sh) >> 44);
}
/* the code word at bit `rel` of the chunk: its length, and through `rank` its rank (after the zone shift), -1 if no word matches */
DEV int code_at(const uint32_t *cw, int rel, bool zoned, const uint16_t *lut, const uint16_t *lut2, int &rank)
{
	const unsigned look = peek20(cw, rel);
	if (zoned && (look >> 11) == 1u) { rank = 110 + (int)((look >> 5) & 63u); return 15; }      /* 000000001 + 6 bits (:127-142) */
	int len, r;
	const unsigned e = lut[look >> 12];
	if (!(e & 0x8000u)) { len = (int)(e >> 8); r = (int)(e & 255u); }
	else {
		const unsigned e2 = lut2[(e & 15u) * 64 + ((look >> 6) & 63u)];
		unsigned d;
		if (e2) { len = (int)(e2 >> 10); r = (int)(e2 & 1023u); }
		else if ((d = (look >> 3) - 0x1f0c0u) < 64u) { len = 17; r = 110 + (int)d; }
		else if ((d = (look >> 3) - 0x1f8c0u) < 46u) { len = 17; r = 174 + (int)d; }
		else if ((d = (look >> 2) - 0x3f1dcu) < 12u) { len = 18; r = 220 + (int)d; }
		else if ((d = (look >> 1) - 0x7e3d0u) < 38u) { len = 19; r = 232 + (int)d; }
		else if ((d = look - 0xfc7ecu) < 20u) { len = 20; r = 270 + (int)d; }
		else { len = 20; r = -1; }
	}
	if (zoned && r >= 110) r += 64;                                  /* :277 */
	rank = r;
	return len;
}

/* A: ranks of the code words that start in chunk `c` of the stream, in order, into syms[]; returns their number.
 * start0: bit offset (relative to the chunk) of the first word, updated to the one of the next chunk. */
DEV int vlc_parse_chunk(const uint8_t *g, int nwords, int c, int &start0, bool zoned, const uint16_t *lut, const uint16_t *lut2,
                        uint32_t *cw, uint16_t *syms, int lane, int &bad)
{
	for (int k = lane; k < VCH_WORDS + 2; k += 64) {
		const int w = c * VCH_WORDS + k;
		uint32_t v = 0;
		if (w < nwords) { const uint8_t *p = g + 4 * (size_t)w; v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
		cw[k] = v;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	__builtin_amdgcn_wave_barrier();
	const int total_bits = (nwords - c * VCH_WORDS) * 32;           /* bits of the stream from the start of this chunk */
	const int lo = 64 * lane, hi = min(64 * (lane + 1), total_bits);
	int start = lane ? lo : start0, exitp = 0, cnt = 0;
	for (;;) {
		int pos = start, n = 0, rk;
		while (pos < hi) { pos += code_at(cw, pos, zoned, lut, lut2, rk); n++; }
		exitp = pos; cnt = n;
		int nxt = from_left(exitp, start0);
		if (nxt < lo) nxt = lo;                                        /* (a span past the end of the stream: nothing starts in it) */
		if (!__any(nxt != start)) break;
		start = nxt;
	}
	int off = cnt;
	off = wscan_add(off);
	{
		int pos = start, at = off - cnt, rk;
		while (pos < hi) { pos += code_at(cw, pos, zoned, lut, lut2, rk); if (rk < 0) { bad = 1; rk = 0; } syms[at++] = (uint16_t)rk; }
	}
	start0 = last_lane(exitp) - 64 * 64;
	if (start0 < 0) start0 = 0;
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	__builtin_amdgcn_wave_barrier();
	return last_lane(off);
}

/* B: the luma placement state.  bits 0-1 mem (3 = three or more), 2 mem2, 3 nhw_ac1, 4-8 which of the five values before are
 * non-zero (bit 4: the last one), 9-10 values written since the last 254-run (3 = three or more, or no such run yet), 11-13 min(e, 5) */
#define ST_NEUTRAL ((3u << 9) | (5u << 11) | (31u << 4))
DEV unsigned luma_step(unsigned s, bool is_run, int rle, bool lit5, bool mark, int &put, int &which)
{
	unsigned mem = s & 3u, mem2 = (s >> 2) & 1u, ac1 = (s >> 3) & 1u, hist = (s >> 4) & 31u, btw = (s >> 9) & 3u, e5 = (s >> 11) & 7u;
	put = 0; which = 0;
	int adv;
	if (is_run) {
		const bool z1 = !(hist & 1u), z2 = !(hist & 2u), z3 = !(hist & 4u), z4 = !(hist & 8u), z5 = !(hist & 16u);
		const bool far = (int)btw + rle >= 3;
		mem = mem < 3u ? mem + 1u : 3u;
		if (mem2) {
			if ((e5 >= 5u && z2 && z3 && z4 && z5) || (rle >= 4 && z2)) { put = 1; which = 2; }
			mem2 = 0;
		} else {
			const bool room = rle >= 4 && e5 > 0u && z1 && !ac1 && far;
			if (mem == 2u && !ac1) { if ((e5 >= 4u && z1 && z2 && z3 && z4 && far) || room) { put = 1; which = 1; mem = 1; } }
			else if (room) { put = 1; which = 1; mem = 1; }
		}
		if (put) hist = ((hist << 1) | 1u) & 31u;
		adv = put + rle;
		if (rle == 254) { ac1 = 1; mem = 0; btw = 0; } else { ac1 = 0; btw = min(3u, btw + (unsigned)adv); }
		hist = rle >= 5 ? 0u : (hist << rle) & 31u;
	} else {
		mem = 0; mem2 = mark ? 1u : 0u; ac1 = 0;
		adv = lit5 ? 5 : 1;
		hist = lit5 ? 17u : ((hist << 1) | 1u) & 31u;
		btw = min(3u, btw + (unsigned)adv);
	}
	e5 = min(5u, e5 + (unsigned)adv);
	return mem | (mem2 << 2) | (ac1 << 3) | (hist << 4) | (btw << 9) | (e5 << 11);
}
DEV int luma_literal(int word, int lvl, int &second)                /* value of a non-run symbol (:324-386); `second`: the value four cells on, for 132..135 */
{
	second = 0;
	switch (word) {
	case 136: return 11;   case 120: return -11;
	case 132: second = 11; return 11;   case 133: second = -11; return 11;
	case 134: second = 11; return -11;  case 135: second = -11; return -11;
	case 127: return 1008; case 129: return 1009; case 125: return 1006; case 126: return 1007;
	case 121: return 1010; case 122: return 1011; case 124: return 11;   case 123: return -11;
	default: return lvl;
	}
}

/* index of the first entry at or behind the start of each of the nseg segments (1 << shift stream positions each) of a list the calling
 * wavefront has just written, and the total behind them: entry k opens every segment from the one after its predecessor's up to its own */
DEV void segment_index(const uint32_t *ent, int n, int shift, int nseg, uint32_t *segt, int lane)
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          /* the entries are this wavefront's own stores */
	__builtin_amdgcn_wave_barrier();
	for (int k0 = 0; k0 < n + 1; k0 += 64) {
		const int k = k0 + lane;
		if (k > n) continue;
		const int cur = k < n ? ENT_POS(ent[k]) >> shift : nseg, prev = k ? ENT_POS(ent[k - 1]) >> shift : -1;
		for (int sg = prev + 1; sg <= cur; sg++) segt[sg] = (uint32_t)k;
	}
}

/* the code table is the same for every file: built once per handle into device memory, copied into LDS by every workgroup */
__global__ __launch_bounds__(64) void k_dec_vlc_table(uint16_t *tab /* [256 + 1024] */)
{
	vlc_fill_lut(tab, tab + 256, threadIdx.x);
}

/* One wavefront per workgroup and stream: the luma streams of the batch first (they are the long ones), then the chroma streams -- as two
 * wavefronts of one workgroup the short chroma walk kept its half of the workgroup's LDS until the luma walk was through, and LDS is what
 * bounds the number of resident walks. */
__global__ __launch_bounds__(64) void k_dec_vlc(DecWs ws, const uint16_t *__restrict__ table)
{
	__shared__ uint16_t lut[256 + 16 * 64];
	__shared__ uint16_t book1[354];
	__shared__ int16_t level1[354];
	__shared__ __attribute__((aligned(16))) uint32_t cw1[VCH_WORDS + 4];
	__shared__ __attribute__((aligned(16))) uint16_t syms1[VCH_SYMS];
	/* The kernel runs next to k_dec_parse on a stream of its own, so it reads the file header itself (a few dozen bytes) instead of the
	 * workspace copy, and reports into a word of its own (D_SPARE[0] / [1]: one per stream) that k_dec_verdict folds into the file's status. */
	__shared__ DecMeta hm;
	const int part = (int)blockIdx.x >= ws.n ? 1 : 0, img = (int)blockIdx.x - part * ws.n, lane = threadIdx.x;
	const uint16_t *lut2 = lut + 256;
	uint16_t *book = book1, *syms = syms1;
	int16_t *level = level1;
	uint32_t *cw = cw1;
	const uint8_t *f = ws.blob + ws.blob_off[img];
	int *verdict = ws.buf<int>(D_SPARE, img) + part;
	/* One lane reads the header and unpacks the code book: byte-serial work, so the bytes are brought into LDS by the wavefront first */
	const uint64_t flen = ws.blob_len[img];
	uint8_t *stage = reinterpret_cast<uint8_t *>(syms);             /* 4224 bytes, free until the first chunk is parsed */
	for (int k = lane; k < HDR_STAGE; k += 64) stage_hdr(stage, f, flen, k);
	for (int k = lane; k < (256 + 16 * 64) / 2; k += 64) reinterpret_cast<uint32_t *>(lut)[k] = reinterpret_cast<const uint32_t *>(table)[k];
	__syncthreads();
	if (!lane) {
		memset(&hm, 0, sizeof hm);
		if (flen > (1u << 24)) hm.status = NHW_E_FORMAT; else parse_header(stage, (uint32_t)flen, &hm);
		*verdict = hm.status;
	}
	__syncthreads();
	const DecMeta *m = &hm;
	if (m->status) return;
	{
		const uint8_t *raw = part ? f + m->o_book2 : f + m->o_book1;
		const int raw_len = part ? m->book2_len : m->book1_len, staged = min(raw_len, BOOK_STAGE);
		for (int k = lane; k < staged; k += 64) stage[1440 + k] = raw[k];
		__syncthreads();
		if (raw_len <= BOOK_STAGE) build_book_wave(stage + 1440, raw_len, part != 0, m->tree_end, book, stage /* 1440 bytes of scratch */, lane);
		else if (!lane) build_book_small(raw, raw_len, stage + 1440, staged, part != 0, m->tree_end, book, stage);   /* (no encoder writes a book that long) */
	}
	__syncthreads();
	for (int r = lane; r < 354; r += 64) level[r] = (int16_t)plain_level(book[r] & 255);
	__syncthreads();
	const uint16_t *bk = book;
	const int16_t *lv = level;
	const uint8_t *g = f + (part ? m->o_packet2 : m->o_packet1);
	const int nwords = part ? m->data2 - m->data1 : m->data1;
	const int nchunks = (nwords + VCH_WORDS - 1) / VCH_WORDS + 1;       /* one more: zero bits behind the stream decode as words too, as in the reference */
	int bad = 0, start0 = 0;
	if (!part) {
		/* The symbols leave as a list of (position in the stream, value) in stream order -- a q20 file has a few thousand values for its
		 * 262 144 luma cells -- with the index of the first entry of every segment of the stream that the un-zig-zag takes in one piece: it
		 * clears its tile in LDS and drops the entries of the tile's segments in.  (Before: the stream itself, 512 KB per file, zeroed by a
		 * fill, written here value by value and read back whole.) */
		uint32_t *ent = ws.buf<uint32_t>(D_B, img);
		int nE = 0;
		const uint8_t *s1 = f + m->o_sel1, *s2 = f + m->o_sel2;
		const int n1 = m->select1, n2 = m->select2;
		const bool zoned = m->res_high < 4;
		const int limit = 4 * DQ - 1;
		int e = 0, t1 = 0, t2 = 0;
		unsigned carry = (0u) | (3u << 9);                            /* nothing before the start: mem 0, no 254-run yet, e = 0, history zero */
		bool done = false;
		for (int c = 0; c < nchunks + 64 && !done; c++) {
			const int nsym = vlc_parse_chunk(g, nwords, c, start0, zoned, lut, lut2, cw, syms, lane, bad);
			for (int base = 0; base < nsym && !done; base += 64) {
				const bool have = base + lane < nsym;
				const int rank = have ? syms[base + lane] : 0;
				const int bkv = bk[rank], word = bkv & 255, rle = bkv >> 8;
				const bool is_run = word == 128, lit5 = word >= 132 && word <= 135, mark = word == 136 || word == 120;
				unsigned sin = lane ? ST_NEUTRAL : carry, sout;
				int put, which;
				for (;;) {
					sout = luma_step(sin, is_run, rle, lit5, mark, put, which);
					const unsigned nxt = (unsigned)from_left((int)sout, (int)carry);
					if (!__any(have && nxt != sin)) break;
					sin = nxt;
				}
				const int adv = have ? put + (is_run ? rle : (lit5 ? 5 : 1)) : 0;
				int pe = adv, p1 = have && put && which == 1 ? 1 : 0, p2 = have && put && which == 2 ? 1 : 0;
				const int q1 = p1, q2 = p2;
				const int nw = have ? put + (is_run ? 0 : lit5 ? 2 : 1) : 0;   /* list entries of this symbol */
				int pw = nw;
				{                                                           /* four prefix sums as two: a batch advances at most 64 x 255 cells and writes at most 192 entries */
					int sa = pe | (pw << 16), sb = p1 | (p2 << 16);
					sa = wscan_add(sa); sb = wscan_add(sb);
					pe = sa & 0xFFFF; pw = sa >> 16; p1 = sb & 0xFFFF; p2 = sb >> 16;
				}
				const int at = e + pe - adv;
				const bool live = have && at < limit;                       /* the reference's loop test: a symbol is taken while e < limit */
				if (live) {                                                 /* (at < limit: every position but the last of a 132..135 symbol is inside the stream) */
					int pos = at, k = nE + pw - nw;
#define ENTRY(P, V) do { ent[k++] = ENT_MAKE(P, V); } while (0)
					if (put) {
						int neg;
						if (which == 1) { const int kk = t1 + p1 - q1; neg = (kk >> 3) < n1 ? (s1[kk >> 3] >> (7 - (kk & 7))) & 1 : 0; }
						else { const int kk = t2 + p2 - q2; neg = !((kk >> 3) < n2 ? (s2[kk >> 3] >> (7 - (kk & 7))) & 1 : 0); }
						ENTRY(pos, neg ? -11 : 11);
						pos++;
					}
					if (!is_run) {
						int second;
						const int v = luma_literal(word, lv[rank], second);
						ENTRY(pos, v);
						if (lit5) { if (pos + 4 < 4 * DQ) ENTRY(pos + 4, second); else ENTRY(pos, v); }   /* past the end: the slot repeats the value */
					}
#undef ENTRY
				}
				const uint64_t lm = __ballot(live);
				if (lm != __ballot(have)) done = true;
				if (lm) {
					const int last = 63 - __builtin_clzll(lm);
					e += __builtin_amdgcn_readlane(pe, last); t1 += __builtin_amdgcn_readlane(p1, last); t2 += __builtin_amdgcn_readlane(p2, last); nE += __builtin_amdgcn_readlane(pw, last);
					carry = (unsigned)__builtin_amdgcn_readlane((int)sout, last);
				}
				if (e >= limit) done = true;
			}
			if (c >= nchunks - 1 && !done && (c + 1) * VCH_WORDS > nwords + 8) { bad = 1; break; }   /* ran out of stream before the last cell */
		}
		segment_index(ent, nE, 11, 128, ws.buf<uint32_t>(D_SEG, img), lane);   /* the strips of the luma stream: k_dec_expand follows each with a cursor */
	} else {
		uint32_t *ent = ws.buf<uint32_t>(D_CB, img);                  /* U on even, V on odd stream positions */
		int nE = 0;
		const int limit = 2 * DQ - 2;
		int e = 0;
		bool done = false;
		for (int c = 0; c < nchunks + 64 && !done; c++) {
			const int nsym = vlc_parse_chunk(g, nwords, c, start0, false, lut, lut2, cw, syms, lane, bad);
			for (int base = 0; base < nsym && !done; base += 64) {
				const bool have = base + lane < nsym;
				const int rank = have ? syms[base + lane] : 0;
				const int bkv = bk[rank], word = bkv & 255;
				const bool is_run = word == 128;
				const int adv = have ? (is_run ? bkv >> 8 : 1) : 0, nw = have && !is_run ? 1 : 0;
				int sa = adv | (nw << 16);
				sa = wscan_add(sa);
				const int pe = sa & 0xFFFF, pw = sa >> 16;
				const int at = e + pe - adv;
				const bool live = have && at < limit;
				if (live && !is_run) {
					const int v = word == 124 ? 5005 : word == 126 ? 5006 : word == 122 ? 5003 : word == 130 ? 5004 : lv[rank];
					ent[nE + pw - 1] = ENT_MAKE(at, v);
				}
				const uint64_t lm = __ballot(live);
				if (lm != __ballot(have)) done = true;
				if (lm) { const int last = 63 - __builtin_clzll(lm); e += __builtin_amdgcn_readlane(pe, last); nE += __builtin_amdgcn_readlane(pw, last); }
				if (e >= limit) done = true;
			}
			if (c >= nchunks - 1 && !done && (c + 1) * VCH_WORDS > nwords + 8) { bad = 1; break; }
		}
		if (!lane) ws.buf<uint32_t>(D_SEG, img)[SEG_CHROMA + 128] = (uint32_t)nE;     /* k_dec_chroma takes its component's entries from the whole list */
	}
	if (__any(bad) && !lane) atomicExch(verdict, (int)NHW_E_FORMAT);
}

/* ---------------------------------------------------------------------------------------------- expand (d3)
 * nhw_decoder.c:493-668 (luma) and the chroma LL2 / exception samples (:943-981, :1231-1267): one wavefront per image.
 *
 * The reference walks the plane in raster order and lets a pattern symbol write constants over its neighbours, so
 * whether a symbol is still there when the walk reaches it depends on the symbols before it.  Rows run in order here; a
 * row is loaded by the whole wavefront (cell = lane + 64k) and its symbols gathered with ballots.
 *   loops 1-2 (rows 0..255, and the left half of rows 256..511): symbols are sparse, so a row with none is one load and a
 *     ballot; a row with some is replayed on an LDS copy, symbol by symbol in column order (not cell by cell).
 *   loop 3 (HH quadrant): besides its symbols every 9..15 coefficient is nudged by what its four neighbours hold at the
 *     moment of the visit.  Which symbols are live comes from the same sparse replay (only 1008/1009 kill their right
 *     neighbour); everything else is a per-cell function of the loaded row, the live masks, the finished row above and
 *     the untouched row below, evaluated by all lanes at once. */
/* stores of this wavefront visible to its own later loads (one CU, one L1): a workgroup-scope fence; an agent-scope one would write the L2 back */
DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }

struct Mask4 { uint64_t w[4]; };
DEV Mask4 m4_prev(const Mask4 &a) { return Mask4{ { a.w[0] << 1, (a.w[1] << 1) | (a.w[0] >> 63), (a.w[2] << 1) | (a.w[1] >> 63), (a.w[3] << 1) | (a.w[2] >> 63) } }; }   /* bit of column j-1 at j */
DEV Mask4 m4_next(const Mask4 &a) { return Mask4{ { (a.w[0] >> 1) | (a.w[1] << 63), (a.w[1] >> 1) | (a.w[2] << 63), (a.w[2] >> 1) | (a.w[3] << 63), a.w[3] >> 1 } }; }   /* bit of column j+1 at j */
DEV Mask4 m4_and(const Mask4 &a, const Mask4 &b) { return Mask4{ { a.w[0] & b.w[0], a.w[1] & b.w[1], a.w[2] & b.w[2], a.w[3] & b.w[3] } }; }
/* this lane's bit of a wave-uniform mask: the mask IS a lane-select operand (v_cndmask), no 64-bit shift per lane needed */
DEV int m4_bit(const Mask4 &a, int k, int) { return (int)__builtin_amdgcn_inverse_ballot_w64(a.w[k]); }

/* replay the pattern symbols of the left half of one of the rows 256..511 (loop 2, :529-560), staged in LDS with its HH half behind it, in
 * column order.  Lane 0 only.  A 1008/1009 in column 0 writes the last cell of the row above: done up front, see the caller. */
DEV void replay_lower(int16_t *st, const uint64_t *mk)
{
	for (int k = 0; k < 4; k++) {
		uint64_t m = mk[k];
		while (m) {
			const int j = 64 * k + __builtin_ctzll(m);
			m &= m - 1;
			int16_t *p = st + j;
			const int s = *p;
			if (s == 1008) { if (j > 0) p[-1] = 5; p[0] = 6; p[1] = 5; }
			else if (s == 1009) { if (j > 0) p[-1] = -5; p[0] = -7; p[1] = -5; }
			else if (s == 1006) { p[0] = -7; p[1] = -7; }
			else if (s == 1007) { p[0] = 7; p[1] = 7; }
		}
	}
}

/* the two entropy branches meet: a file the prefix-code walk gave up on is skipped by everything that follows */
__global__ __launch_bounds__(256) void k_dec_verdict(DecWs ws)
{
	const int img = blockIdx.x * 256 + threadIdx.x;
	if (img >= ws.n) return;
	const int v = walk_verdict(ws, img);
	DecMeta *m = ws.buf<DecMeta>(D_META, img);
	if (v && !m->status) m->status = v;
}

/* one pattern symbol of loop 1 (:493-527) at column j of a row staged in LDS; the rows around it are staged contiguously, so the cell in
 * front of column 0 and the cell behind column 511 of the row below are where the reference's raster arithmetic puts them */
DEV void replay_upper(int16_t *p, int j)
{
	switch (*p) {
	case 1008: p[-1] = 5; p[1] = 5; p[0] = (int16_t)(j < DH ? 5 : 6); break;
	case 1009: p[-1] = -5; p[1] = -5; p[0] = (int16_t)(j < DH ? -6 : -7); break;
	case 1010: p[0] = 5; p[1] = 5; p[DW] = 5; p[DW + 1] = 5; break;
	case 1011: p[0] = -5; p[1] = -5; p[DW] = -5; p[DW + 1] = -5; break;
	case 1006: p[0] = -6; p[1] = -6; break;
	case 1007: p[0] = 6; p[1] = 6; break;
	default: break;                                               /* a symbol an earlier one wrote over */
	}
}
DEV unsigned symbols_of(const uint4 &r)                          /* which of a lane's eight cells hold a pattern symbol */
{
	const unsigned w[4] = { r.x, r.y, r.z, r.w };
	unsigned sm = 0;
#pragma unroll
	for (int e = 0; e < 4; e++) {
		sm |= ((int)(int16_t)(w[e] & 0xFFFFu) > 1000 ? 1u : 0u) << (2 * e);
		sm |= ((int)(int16_t)(w[e] >> 16) > 1000 ? 1u : 0u) << (2 * e + 1);
	}
	return sm;
}

#define XC 8                          /* rows per chunk of loop 1 */
#define XD 4                          /* rows per chunk of loops 2 and 3: their row work needs more registers, the rows in flight fewer */
#define X_CELLS (8 + (XC + 1) * DW + 8)
#define X_NONE 0x7fff
__global__ __launch_bounds__(256) void k_dec_expand(DecWs ws)
{
	__shared__ __attribute__((aligned(16))) int16_t stage[4][X_CELLS];
	__shared__ uint8_t symb[4][XC + 1][64];
	const int img = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (img >= ws.n) return;
	DecMeta *m = ws.buf<DecMeta>(D_META, img);
	if (m->status) return;
	const int q = m->q;
	int16_t *a = plane_a(ws, img);
	int16_t *st = stage[threadIdx.x >> 6];
	const uint8_t *f = ws.blob + ws.blob_off[img];
	/* The level-1 detail bands -- three quarters of the plane -- are mostly zeros, and the only reader of those cells is k_dec_final: a
	 * 16-byte group of a row that holds nothing is NOT stored (2.1 GB of zeros per 4096 files went out here and came back there); nzg[row]
	 * says which of the row's 64 groups are in memory (the level-1 LL quadrant, rows and columns below 256, always is: the level-2 kernel
	 * and the passes at the end of this one work in it).  What a later symbol patches into a row that has left goes through put_late. */
	uint64_t *nzg = ws.buf<uint64_t>(D_NZG, img);
	uint64_t last_mask = 0;                                       /* the groups of the row stored last */
	auto store_row = [&](int row, const int16_t *src, bool upper) {
		const uint4 v = *(const uint4 *)(src + 8 * lane);
		const bool keep = (upper && lane < 32) || (v.x | v.y | v.z | v.w) != 0 || ws.dense;
		if (keep) *(uint4 *)(a + (size_t)row * DW + 8 * lane) = v;
		last_mask = __ballot(keep);
		if (!lane) nzg[row] = last_mask;
	};
	/* cells of the row stored last (flag: this lane has one, at column col) behind its stores: groups that were left out come into being as zeros first */
	auto put_late = [&](int row, bool flag, int col, int val) {
		uint64_t need = __ballot(flag), groups = 0;
		while (need) { const int l = __builtin_ctzll(need); need &= need - 1; groups |= 1ull << (__builtin_amdgcn_readlane(col, l) >> 3); }
		const uint64_t missing = groups & ~last_mask;
		wave_sync();
		if ((missing >> lane) & 1ull) *(uint4 *)(a + (size_t)row * DW + 8 * lane) = make_uint4(0, 0, 0, 0);
		if (missing) { last_mask |= missing; if (!lane) nzg[row] = last_mask; wave_sync(); }
		if (flag) a[(size_t)row * DW + col] = (int16_t)val;
	};

	/* Where the rows come from: the prefix-code walk left the file's values as a list in stream order -- 128 strips of 4 columns, a strip row
	 * after row -- so the values of any range of rows are, for every strip, the next few entries of the strip's part of the list.  A lane
	 * follows two strips (lane, lane + 64) with a cursor and the next four entries of each already in registers: feeding rows into the LDS
	 * buffer is zeros, then each lane drops the entries of its strips that lie in those rows into their cells and asks for the next four
	 * (which arrive while the rows are being worked on).  The plane in memory is only ever written here, row by row, finished. */
	struct __attribute__((aligned(4))) Ent4 { uint32_t w[4]; };
	const uint32_t *ent = ws.buf<uint32_t>(D_B, img);
	int ecur0, ecur1, eend0, eend1;
	Ent4 ewin0, ewin1;
	{
		const uint32_t *sidx = ws.buf<uint32_t>(D_SEG, img);
		ecur0 = (int)sidx[lane]; eend0 = (int)sidx[lane + 1]; ecur1 = (int)sidx[lane + 64]; eend1 = (int)sidx[lane + 65];
		ewin0 = *reinterpret_cast<const Ent4 *>(ent + ecur0); ewin1 = *reinterpret_cast<const Ent4 *>(ent + ecur1);
	}
	auto feed_strip = [&](int strip, int &ecur, int eend, Ent4 &ewin, int hi, int base) {
		const int limit = strip * 2048 + 4 * (hi + 1);
		for (;;) {
			int used = 0;
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const uint32_t en = ewin.w[t];
				if (ecur + t < eend && ENT_POS(en) < limit) {
					const int w = ENT_POS(en) - strip * 2048, idx = w & 7;
					st[8 + (2 * (w >> 3) + (idx >> 2) - base) * DW + 4 * strip + ((idx & 4) ? 7 - idx : idx)] = (int16_t)ENT_VAL(en);
					used = t + 1;
				}
			}
			if (!used) break;
			ecur += used;
			ewin = *reinterpret_cast<const Ent4 *>(ent + ecur);
			if (used < 4) break;
		}
	};
	auto feed = [&](int lo, int hi, int base) {                   /* rows lo .. hi into slots lo - base .. hi - base */
		for (int r = lo; r <= hi; r++) *(uint4 *)(st + 8 + (r - base) * DW + 8 * lane) = make_uint4(0, 0, 0, 0);
		__builtin_amdgcn_wave_barrier();
		feed_strip(lane, ecur0, eend0, ewin0, hi, base);
		feed_strip(lane + 64, ecur1, eend1, ewin1, hi, base);
	};
	int pend_lower = X_NONE;

	/* loop 1: rows 0..255, all columns (:493-527).  A symbol writes into its own row, the row below, the last cell of the row above and the
	 * first cell of the row after next; the walk is serial, but only over the symbols.  Rows pass through LDS XC at a time -- slot s of the
	 * buffer is row i0+s, slot XC is the row below the chunk (the next chunk's slot 0) -- lane 0 replays the symbols of slots 0..XC-1 in
	 * raster order while the next XC rows are already on their way from HBM, and the chunk goes back with one 16-byte store per lane and row.
	 * Which cells are symbols is taken from the rows as loaded: a replay only ever writes small values, it cannot make a new symbol, and a
	 * symbol that was written over is seen as such when the walk reads it. */
	{
		int16_t *buf = st + 8;
		uint8_t *sm = &symb[threadIdx.x >> 6][0][0];
		uint64_t any_halo;
		feed(0, 0, 0);
		__builtin_amdgcn_wave_barrier();
		{
			const unsigned s0 = symbols_of(*(const uint4 *)(buf + 8 * lane));
			sm[lane] = (uint8_t)s0;
			any_halo = __ballot(s0 != 0);
		}
		int pend = X_NONE;                                        /* what a column-511 symbol of the chunk's last row left for column 0 of the row after next */
		for (int i0 = 0; i0 < DH; i0 += XC) {
			uint64_t any[XC + 1]; uint64_t any_all = any_halo;          /* any[s]: slot s holds a symbol */
			any[0] = any_halo;
			feed(i0 + 1, i0 + XC, i0);
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int g = 0; g < XC; g++) {
				const unsigned sg = symbols_of(*(const uint4 *)(buf + (g + 1) * DW + 8 * lane));
				sm[(g + 1) * 64 + lane] = (uint8_t)sg;
				any[g + 1] = __ballot(sg != 0);
				if (g + 1 < XC) any_all |= any[g + 1];
			}
			if (!lane) {
				if (pend != X_NONE) buf[DW] = (int16_t)pend;
				buf[-1] = X_NONE; buf[(XC + 1) * DW] = X_NONE;
			}
			__builtin_amdgcn_wave_barrier();
			if (any_all) {
				if (!lane) {
#pragma unroll
					for (int s = 0; s < XC; s++) {
						uint64_t mm = any[s];
						while (mm) {
							const int l = __builtin_ctzll(mm);
							mm &= mm - 1;
							unsigned bits = sm[s * 64 + l];
							while (bits) {
								const int e = __builtin_ctz(bits);
								bits &= bits - 1;
								replay_upper(buf + s * DW + 8 * l + e, 8 * l + e);
							}
						}
					}
				}
				__builtin_amdgcn_wave_barrier();
				const int up = buf[-1];
				if (up != X_NONE) {                                     /* the last cell of the row above, behind the rows the last chunk stored (in front of row 0: the pad) */
					if (i0) put_late(i0 - 1, lane == 0, DW - 1, up);
					else { wave_sync(); if (!lane) a[-1] = (int16_t)up; }
				}
			}
			pend = buf[(XC + 1) * DW];
#pragma unroll
			for (int s = 0; s < XC; s++) store_row(i0 + s, buf + s * DW, true);
			{
				const uint4 h = *(const uint4 *)(buf + XC * DW + 8 * lane);
				const uint8_t hs = sm[XC * 64 + lane];
				__builtin_amdgcn_wave_barrier();
				*(uint4 *)(buf + 8 * lane) = h;
				sm[lane] = hs;
				any_halo = any[XC];
			}
		}
		pend_lower = pend;                                          /* row 256 stays in slot 0, with what rows 254 and 255 wrote into it */
	}

	/* loops 2 and 3, row by row: the left half's pattern symbols (:529-560), then the HH half (:562-616).  The reference finishes
	 * loop 2 before loop 3 starts; going row by row instead only differs where the two loops touch each other's cells across a row
	 * end: a 1008/1009 in column 0 (loop 2) writes the last HH cell of the row above -- applied here before anything else -- and a
	 * 1008/1009 in column 511 (loop 3) writes column 0 of the next row -- held back until that row's loop-2 part is done.
	 * The rows pass through the same LDS buffer as in loop 1, XD at a time with the next XD on their way: a row reads the untouched HH
	 * half of the row below (slot s+1), writes itself and one cell class into the left half of the row above (slot s-1; for slot 0 that
	 * row has left already and is patched in memory), and the finished HH row above stays in registers. */
	{
		int16_t *buf = st + 8;
		{                                                           /* row 256 is in slot 0: a 1008/1009 in its column 0 writes the last cell of row 255, which has left */
			const int s256 = buf[0];
			if (s256 == 1008 || s256 == 1009) put_late(DH - 1, lane == 0, DW - 1, s256 == 1008 ? 5 : -5);
			wave_sync();
		}
		int carry = m->carry;
		int pend0 = X_NONE;                                       /* value a column-511 symbol of the previous row writes to column 0 of this one */
		int upf[4];
#pragma unroll
		for (int k = 0; k < 4; k++)                                   /* row 255's HH half: a group that was not stored is zeros, whatever an earlier batch left in the plane (last_mask is row 255's: loop 1's last store and the put_late above) */
			upf[k] = ((last_mask >> (32 + 8 * k + (lane >> 3))) & 1ull) ? (int)a[(size_t)(DH - 1) * DW + DH + lane + 64 * k] : 0;
		for (int i0 = DH; i0 < DW; i0 += XD) {
			if (i0 + 1 < DW) feed(i0 + 1, i0 + XD < DW ? i0 + XD : DW - 1, i0);
			__builtin_amdgcn_wave_barrier();
			if (i0 == DH && pend_lower != X_NONE && !lane) buf[DW] = (int16_t)pend_lower;   /* what loop 1's last row left for column 0 of row 257 */
			__builtin_amdgcn_wave_barrier();
			if (lane < XD && i0 + 1 + lane < DW) {                  /* a 1008/1009 in column 0 of a row writes the last HH cell of the row above: before anything else */
				const int s0 = buf[(1 + lane) * DW];
				if (s0 == 1008 || s0 == 1009) buf[lane * DW + DW - 1] = (int16_t)(s0 == 1008 ? 5 : -5);
			}
			__builtin_amdgcn_wave_barrier();
#pragma unroll 1
			for (int s = 0; s < XD; s++) {
				const int i = i0 + s;
				int16_t *row = buf + s * DW;
				{
					int v[4]; uint64_t mk[4]; uint64_t any = 0;
#pragma unroll
					for (int k = 0; k < 4; k++) v[k] = row[lane + 64 * k];
#pragma unroll
					for (int k = 0; k < 4; k++) { mk[k] = __ballot(v[k] > 1000); any |= mk[k]; }
					if (any) {
						if (!lane) replay_lower(row, mk);
						__builtin_amdgcn_wave_barrier();
					}
				}
				if (pend0 != X_NONE) { if (!lane) row[0] = (int16_t)pend0; pend0 = X_NONE; }
				/* HH half of the row: columns 256..511 */
				int cur[4], dn[4];
#pragma unroll
				for (int k = 0; k < 4; k++) { const int c = DH + lane + 64 * k; cur[k] = row[c]; dn[k] = i + 1 < DW ? row[c + DW] : 0; }
				Mask4 k8, k9, s67, liveK = { { 0, 0, 0, 0 } }, live67 = { { 0, 0, 0, 0 } };
				uint64_t any = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					k8.w[k] = __ballot(cur[k] == 1008); k9.w[k] = __ballot(cur[k] == 1009); s67.w[k] = __ballot(cur[k] == 1006 || cur[k] == 1007);
					any |= k8.w[k] | k9.w[k] | s67.w[k];
				}
				if (!any && q >= 23) {
#pragma unroll
					for (int k = 0; k < 4; k++) upf[k] = cur[k];
					continue;
				}
				if (any) {                                           /* which symbols are still there when the walk reaches them */
					int dead_at = -1;
#pragma unroll
					for (int k = 0; k < 4; k++) {
						uint64_t mm = k8.w[k] | k9.w[k] | s67.w[k];
						while (mm) {
							const int b = __builtin_ctzll(mm), col = 64 * k + b;
							mm &= mm - 1;
							if (col == dead_at) continue;
							if ((s67.w[k] >> b) & 1ull) live67.w[k] |= 1ull << b;
							else { liveK.w[k] |= 1ull << b; dead_at = col + 1; }
						}
					}
				}
				const Mask4 live8 = m4_and(liveK, k8);
				const Mask4 kL = m4_prev(liveK), kL2 = m4_prev(kL), kR = m4_next(liveK), p8L = m4_prev(live8), p8R = m4_next(live8), s67L = m4_prev(live67);
				unsigned cand = 0, hit[4];
				int fin[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int c = DH + lane + 64 * k;
					const int lw = k ? __builtin_amdgcn_readlane(cur[k - 1], 63) : 0, rw = k < 3 ? __builtin_amdgcn_readlane(cur[k + 1], 0) : 0;
					const int lv = from_left(cur[k], lw);                   /* lane 0 takes the word seam */
					const int r = __builtin_amdgcn_update_dpp(0, cur[k], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
					const int rv = lane < 63 ? r : rw;
					const int lk = m4_bit(liveK, k, lane), l67 = m4_bit(live67, k, lane), lkL = m4_bit(kL, k, lane), lkR = m4_bit(kR, k, lane);
					const bool cd = !lkL && !lk && !l67 && cur[k] <= 1000 && iabs(cur[k]) > 8 && iabs(cur[k]) < 16 && c > DH && c < DW - 1 && q < 23;
					const bool lsmall = m4_bit(kL2, k, lane) || m4_bit(s67L, k, lane) || iabs(lv) < 8;
					hit[k] = (unsigned)((lsmall ? 1 : 0) + (iabs(rv) < 8) + (iabs(upf[k]) < 8) + (iabs(dn[k]) < 8));
					cand |= cd ? 1u << k : 0u;
					int v = cur[k];
					if (lkR) v = m4_bit(p8R, k, lane) ? 5 : -5;
					else if (lk) v = cur[k] == 1008 ? 6 : -7;
					else if (l67) v = 0;
					else if (lkL) v = m4_bit(p8L, k, lane) ? 5 : -5;
					fin[k] = v;
				}
				if (carry) {                                        /* the very first candidate of the walk also gets the left-over count */
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint64_t bm = __ballot((cand >> k) & 1);
						if (bm && carry) { if (lane == __builtin_ctzll(bm)) hit[k] += (unsigned)carry; carry = 0; }
					}
				}
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int lkR = m4_bit(kR, k, lane);
					if (((cand >> k) & 1) && hit[k] >= 2 && !lkR) fin[k] = cur[k] > 0 ? cur[k] + 1 : cur[k] - 1;
					if (fin[k] != cur[k]) row[DH + lane + 64 * k] = (int16_t)fin[k];
					upf[k] = fin[k];
				}
				if (any) {                                           /* what the symbols write outside the HH half of this row */
					const bool up67 = (live67.w[0] | live67.w[1] | live67.w[2] | live67.w[3]) != 0;
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const int c = DH + lane + 64 * k;
						const bool l67 = m4_bit(live67, k, lane);
						const int16_t val = (int16_t)(cur[k] == 1006 ? -7 : 7);
						if (l67) { row[c - DH] = val; if (s) row[c - 3 * DH] = val; }
						if (up67 && s == 0 && live67.w[k]) put_late(i - 1, l67, c - DH, val);   /* the row above went to memory with the chunk before: patched there, behind those stores */
						if (m4_bit(liveK, k, lane)) {
							const int16_t val = (int16_t)(cur[k] == 1008 ? 5 : -5);
							if (c == DH) row[DH - 1] = val;
						}
					}
					if ((liveK.w[3] >> 63) & 1ull) pend0 = ((k8.w[3] >> 63) & 1ull) ? 5 : -5;
				}
			}
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int s = 0; s < XD; s++) store_row(i0 + s, buf + s * DW, false);
			{
				const uint4 h = *(const uint4 *)(buf + XD * DW + 8 * lane);
				__builtin_amdgcn_wave_barrier();
				*(uint4 *)(buf + 8 * lane) = h;
			}
		}
	}
	wave_sync();

	/* LL2 samples (:618-625): 16 of them per lane and step */
	{
		const uint4 *l4 = (const uint4 *)ws.buf<uint8_t>(D_LL, img);
		for (int k = lane; k < DQ / 64; k += 64) {
			const uint4 b = l4[k];
			const unsigned w[4] = { b.x, b.y, b.z, b.w };
			unsigned o[8];
#pragma unroll
			for (int e = 0; e < 4; e++) { o[2 * e] = (w[e] & 0xFFu) | ((w[e] & 0xFF00u) << 8); o[2 * e + 1] = ((w[e] >> 16) & 0xFFu) | ((w[e] >> 24) << 16); }
			uint4 *dst = (uint4 *)(a + (size_t)(k >> 3) * DW + (k & 7) * 16);
			dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
			dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
		}
	}
	wave_sync();
	if (q > 17) {                                                  /* odd-LL tags (:627-654): a tag makes four cells odd, which is the same done once or twice, in any order; the row a tag
	                                                                * is in is the number of row-advancing bytes in front of it */
		const uint8_t *r4 = f + m->o_res4;
		const int n = m->res4_len;
		int rowbase = 0;
		for (int i0 = 0; i0 < n; i0 += 64) {
			const bool valid = i0 + lane < n;
			const int v = valid ? r4[i0 + lane] : 0;
			const uint64_t adv = __ballot(valid && v >= 128);
			const int rowi = rowbase + __builtin_popcountll(adv & ((1ull << lane) - 1ull));
			rowbase += __builtin_popcountll(adv);
			if (valid && v != 128) {
				const int at = (rowi << 9) + (v > 128 ? v - 129 : v - 1);
				if (at >= 0 && at + 3 < 4 * DQ) {
					int c4[4];
#pragma unroll
					for (int k = 0; k < 4; k++) c4[k] = a[at + k];
#pragma unroll
					for (int k = 0; k < 4; k++) if (!(c4[k] & 1)) a[at + k] = (int16_t)(c4[k] + 1);
				}
			}
		}
		wave_sync();
	}
	if (!lane) {
		/* exception samples of the luma plane (:656-668); U and V follow on the same cursor (k_dec_chroma) */
		const uint8_t *x = f + m->o_exw;
		const int n = m->exw_len;
#define XB(k) ((k) < n ? (int)x[k] : 0)
		for (int i = 0; i < n; i += 3) {
			if (!XB(i) && !XB(i + 1)) break;
			const int hi = XB(i + 1) >= 128, lo = XB(i + 1) & 127;
			a[(XB(i) << 9) + lo] = (int16_t)(hi ? XB(i + 2) + 255 : -XB(i + 2));
		}
#undef XB
	}
}

/* add to element idx of an int16 array in LDS (dword-aligned base): compare-and-swap on the dword, the pointer stays a typed offset of
 * the base so that it compiles to the LDS instruction */
DEV void add_i16_at(int16_t *base, int idx, int delta)
{
	unsigned *w = reinterpret_cast<unsigned *>(base) + (idx >> 1);
	const int sh = (idx & 1) << 4;
	unsigned old = *w, seen;
	do {
		seen = old;
		const unsigned v = ((((seen >> sh) & 0xFFFFu) + (unsigned)delta) & 0xFFFFu) << sh;
		old = atomicCAS(w, seen, (seen & ~(0xFFFFu << sh)) | v);
	} while (old != seen);
}

/* ---------------------------------------------------------------------------------------------- synthesis (d4)
 * The 5/3 synthesis over rows of [low half | high half] (decoder/filters.c:143-194), driven as in decoder/wavelet_filterbank.c:52-357:
 * along the rows, transpose, along the rows again (normalised), which leaves every level's result transposed.
 */
/* Both directions of a level run on one LDS residency of the block: filtered along the rows in place (a wavefront owns a row: it has read
 * it before it writes it), then along the columns, and column c of the result is row c of the plane -- the transposed orientation the
 * reference's transposes leave behind and everything downstream expects.  The plane in between never travels.  The block kernels below
 * (k_dec_luma_l2, k_dec_chroma) fill a CU's LDS with one 1024-thread workgroup, which therefore works through several blocks. */
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   /* orders LDS traffic only: global loads stay in flight across it */
template <int S>
DEV void synth_pair(const int16_t *x, int st, int k, bool norm, int &ev, int &od)
{
	constexpr int M = S / 2;
	const int16_t *lo = x, *hi = x + M * st;
	const int l0 = lo[k * st], ln = k + 1 < M ? lo[(k + 1) * st] : l0;
	const int h0 = hi[k * st], hp = k > 0 ? hi[(k - 1) * st] : h0, hn = k + 1 < M ? hi[(k + 1) * st] : h0;
	ev = (int16_t)((int16_t)(l0 << 3) - ((h0 + hp) << 1));
	od = (int16_t)((int16_t)((l0 + ln) << 2) + (6 * h0 - hp - hn));
	if (norm) { if (ev > 0) ev = (int16_t)(ev + 32); ev >>= 6; if (od > 0) od = (int16_t)(od + 32); od >>= 6; }
}
/* ---------------------------------------------------------------------------------------------- level 2 of the luma (:670-787)
 * Three passes of the reference on one LDS residency of the 256 x 256 block (plane A's top-left quarter), one launch:
 *   shrink    isolated level-2 coefficients shrink by one (:670-721): a stencil on the values as they were -- a cell that shrinks cannot
 *             have a neighbour that does, but its new value would read differently to that neighbour, so all decisions are taken (a
 *             thread slides a 3 x 3 window down 64 rows of a column, a bit per row) before any is applied;
 *   synthesis both directions, in place (see k_dec_synth2d): afterwards LDS holds sample (row c, column j) of the level-1 LL, in the
 *             transposed orientation the plane keeps, at [j][c];
 *   residuals the three residual lists onto it (:731-787): positions repeat, the steps commute: compare-and-swap adds on the LDS words;
 * then column c of the LDS block leaves as row c of the plane.  Before: three kernels, each a round trip of the block (and the residual
 * kernel read every list eight times, once per band of rows).  `upto`: the debug stop (1: write the block back after the shrink, 2: after
 * the synthesis; 3: everything). */
__global__ __launch_bounds__(1024) void k_dec_luma_l2(DecWs ws, int items, int upto)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	constexpr int S = DH, LS = S + 2, HLF = S / 2, PPL = HLF / 64, NT_ = 1024, NPRE = S * (S / 8) / NT_;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	uint4 pre[NPRE];
	if ((int)blockIdx.x < items) {
		const int16_t *src = plane_a(ws, blockIdx.x);
#pragma unroll
		for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(v / (S / 8)) * DW + 8 * (v % (S / 8))); }
	}
	for (int img = blockIdx.x; img < items; img += gridDim.x) {
		int16_t *pl = plane_a(ws, img);
		const DecMeta *m = ws.buf<DecMeta>(D_META, img);
		const bool skip = m->status != 0;
		const int q = m->q;
#pragma unroll
		for (int u = 0; u < NPRE; u++) {
			const int v = t + u * NT_, row = v / (S / 8), o = v % (S / 8);
			uint32_t *d = reinterpret_cast<uint32_t *>(smem + row * LS + 8 * o);
			d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
		}
		lds_barrier();
		if (img + (int)gridDim.x < items) {
			const int16_t *src = plane_a(ws, img + gridDim.x);
#pragma unroll
			for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(v / (S / 8)) * DW + 8 * (v % (S / 8))); }
		}
		{
			/* shrink: a wavefront takes 16 rows, its lanes the columns (lane + 64k).  "Loud" (|v| > 8, > diag) is one compare per cell into a
			 * row mask; the 3 x 3 rule is then mask algebra on the three rows around a cell (the scalar unit's work, 256 columns at a time), and
			 * a lane only keeps its own 4 x 16 verdicts, as bits, until every wavefront has taken its decisions. */
			const int diag = q <= 16 ? 16 : 8, i_first = 16 * wv;
			auto loud = [&](int r, Mask4 &m8, Mask4 &md) {
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int a = r >= 0 && r < S ? iabs((int)smem[r * LS + lane + 64 * k]) : 0;
					m8.w[k] = __ballot(a > 8); md.w[k] = __ballot(a > diag);
				}
			};
			Mask4 p8, pd, c8, cd, n8, nd;
			loud(i_first - 1, p8, pd); loud(i_first, c8, cd);
			unsigned hb0 = 0, hb1 = 0;                               /* rows 0..7, 8..15 of mine: four bits a row */
#pragma unroll 1
			for (int t = 0; t < 16; t++) {
				const int i = i_first + t;
				loud(i + 1, n8, nd);
				Mask4 h;
				const Mask4 c8l = m4_prev(c8), c8r = m4_next(c8), pdl = m4_prev(pd), pdr = m4_next(pd), ndl = m4_prev(nd), ndr = m4_next(nd);
#pragma unroll
				for (int k = 0; k < 4; k++) h.w[k] = c8.w[k] & ~(c8l.w[k] | c8r.w[k] | p8.w[k] | n8.w[k] | pdl.w[k] | pdr.w[k] | ndl.w[k] | ndr.w[k]);
				h.w[0] &= ~1ull; h.w[3] &= ~(1ull << 63);                 /* columns 1 .. 254 */
				if (i < HLF) { h.w[0] = 0; h.w[1] = 0; }                  /* not the LL2 quadrant */
				if (i < 1 || i > S - 2) { h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0; }
				unsigned nib = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) nib |= (unsigned)m4_bit(h, k, lane) << k;
				if (t < 8) hb0 |= nib << (4 * t); else hb1 |= nib << (4 * t - 32);
				p8 = c8; pd = cd; c8 = n8; cd = nd;
			}
			lds_barrier();
			for (int half = 0; half < 2; half++) {
				unsigned hb = half ? hb1 : hb0;
				while (hb) {
					const int b = __builtin_ctz(hb);
					hb &= hb - 1;
					int16_t *cell = smem + (i_first + 8 * half + (b >> 2)) * LS + lane + 64 * (b & 3);
					*cell = (int16_t)(*cell > 0 ? *cell - 1 : *cell + 1);
				}
			}
			lds_barrier();
		}
		if (upto == 1) {
			if (!skip)
				for (int v = t; v < S * (S / 8); v += NT_) {
					const int row = v / (S / 8), o = v % (S / 8);
					const uint32_t *d = reinterpret_cast<const uint32_t *>(smem + row * LS + 8 * o);
					*reinterpret_cast<uint4 *>(pl + (size_t)row * DW + 8 * o) = make_uint4(d[0], d[1], d[2], d[3]);
				}
			lds_barrier();
			continue;
		}
		for (int i = 0; i < 16; i++) {                               /* along the rows, un-normalised */
			int16_t *x = smem + (wv * 16 + i) * LS;
			int e[PPL], o[PPL];
#pragma unroll
			for (int u = 0; u < PPL; u++) synth_pair<S>(x, 1, lane + 64 * u, false, e[u], o[u]);
#pragma unroll
			for (int u = 0; u < PPL; u++) reinterpret_cast<uint32_t *>(x)[lane + 64 * u] = (uint32_t)(uint16_t)e[u] | ((uint32_t)(uint16_t)o[u] << 16);
		}
		lds_barrier();
		for (int i = 0; i < 16; i++) {                               /* along the columns, normalised, in place: sample 2k, 2k+1 of column c */
			int16_t *x = smem + wv * 16 + i;
			int e[PPL], o[PPL];
#pragma unroll
			for (int u = 0; u < PPL; u++) synth_pair<S>(x, LS, lane + 64 * u, true, e[u], o[u]);
#pragma unroll
			for (int u = 0; u < PPL; u++) { const int k = lane + 64 * u; x[(2 * k) * LS] = (int16_t)e[u]; x[(2 * k + 1) * LS] = (int16_t)o[u]; }
		}
		lds_barrier();
		if (upto >= 3 && !skip) {                                    /* residual lists: plane cell (row, col) sits at [col][row] */
			const uint8_t *f = ws.blob + ws.blob_off[img];
#define ACC(row, col, d) add_i16_at(smem, (col) * LS + (row), (d))
			if (q >= 21) {
				const uint16_t *p5 = ws.buf<uint16_t>(D_P5, img);
				const int cnt = (m->res5_bits - 1) * 8;
				for (int k = t; k < cnt; k += NT_) { const int p = p5[k]; ACC(p >> 8, p & 255, bit_of(f + m->o_res5_word, m->res5_bits, k) ? -3 : 3); }
			}
			if (q > 12) {
				const uint16_t *p1 = ws.buf<uint16_t>(D_P1, img);
				const int amp = q >= 18 ? 5 : q >= 15 ? 7 : 9, cnt = (m->res1_bits - 1) * 8;
				for (int k = t; k < cnt; k += NT_) { const int p = p1[k]; ACC(p >> 8, p & 255, bit_of(f + m->o_res1_word, m->res1_bits, k) ? -amp : amp); }
			}
			if (q >= 19) {
				const uint16_t *p3 = ws.buf<uint16_t>(D_P3, img);
				const uint8_t *w = f + m->o_res3_word;
				const int cnt = (m->res3_bits * 2 - 2) * 4;
				for (int k = t; k < cnt; k += NT_) {
					const int p = p3[k], row = p >> 8, col = p & 255;
					const int sel = (w[k >> 2] >> (6 - 2 * (k & 3))) & 3;
					/* rows 254/255 reach below the level-1 LL: in the reference those cells are scratch that the next pass overwrites; here they are
					 * the level-1 detail bands, so those adds are dropped */
					const int d0 = sel == 1 ? -4 : sel == 0 ? 4 : sel == 2 ? 2 : -2, d1 = sel == 1 ? -3 : sel == 0 ? 3 : sel == 2 ? 2 : -2, d2 = sel == 2 ? 2 : sel == 3 ? -2 : 0;
					ACC(row, col, d0);
					if (row + 1 < DH) ACC(row + 1, col, d1);
					if (d2 && row + 2 < DH) ACC(row + 2, col, d2);
				}
			}
#undef ACC
		}
		lds_barrier();
		if (!skip)
		for (int i = 0; i < 16; i++) {                               /* column c of the block is row c of the plane */
			const int c = wv * 16 + i;
			uint32_t *dst = reinterpret_cast<uint32_t *>(pl + (size_t)c * DW);
#pragma unroll
			for (int u = 0; u < PPL; u++) {
				const int k = lane + 64 * u;
				dst[k] = (uint32_t)(uint16_t)smem[(2 * k) * LS + c] | ((uint32_t)(uint16_t)smem[(2 * k + 1) * LS + c] << 16);
			}
		}
		lds_barrier();                                               /* the block is done with before the next one moves in */
	}
}

/* The same three passes, a QUARTER of the block to a 256-thread workgroup (production; the debug stops keep k_dec_luma_l2): quarter p owns the
 * 64 columns 64 p .. 64 p + 63 of the row pass's result = rows 64 p .. of the plane.  Its row pass reads, of every row, the low-band cells
 * 32 p .. 32 p + 32 and the high-band cells 32 p - 1 .. 32 p + 32; the shrink before it looks one cell further: two windows of 36 cells a row
 * (dword-aligned: block columns 32 p - 2 .. and 126 + 32 p ..), 38 KB of LDS, four workgroups a CU -- whose load, filter and store phases
 * overlap where the 1024-thread block kernel's follow one another (0.71 ms for 1.3 GB).  Each quarter takes the shrink's decisions for the cells
 * it reads (from the values as they were, all before any is applied), filters in place, adds the residuals whose plane row is its own and
 * writes 64 whole rows of the plane.  Block b -> file ((b >> 5) << 3) | (b & 7), quarter (b >> 3) & 3 (a file's quarters on one XCD). */
#define LQ_LS 74                      /* pitch of a row of the tile in shorts (37 dwords: column walks on 32 banks) */
__global__ __launch_bounds__(256) void k_dec_luma_l2q(DecWs ws, int items)
{
	__shared__ __attribute__((aligned(16))) int16_t T[DH * LQ_LS];
	constexpr int S = DH, HLF = S / 2;
	const int b = blockIdx.x, img = ((b >> 5) << 3) | (b & 7), p = (b >> 3) & 3;
	if (img >= items) return;
	const DecMeta *m = ws.buf<DecMeta>(D_META, img);
	if (m->status) return;
	const int q = m->q;
	int16_t *pl = plane_a(ws, img);
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const int a_lo = 32 * p - 2, a_hi = 126 + 32 * p;             /* the windows' first block columns */
	{                                                                /* all of a thread's 36 loads in flight at once (a clamped address where the window leaves the block: no branch in front of a load) */
		uint32_t v[36];
#pragma unroll
		for (int j = 0; j < 36; j++) {
			const int idx = t + 256 * j, row = idx / 36, d = idx - row * 36;
			const int col = d < 18 ? a_lo + 2 * d : a_hi + 2 * (d - 18);
			const int cc = col < 0 ? 0 : col > S - 2 ? S - 2 : col;
			v[j] = *reinterpret_cast<const uint32_t *>(pl + (size_t)row * DW + cc);
		}
#pragma unroll
		for (int j = 0; j < 36; j++) {
			const int idx = t + 256 * j, row = idx / 36, d = idx - row * 36;
			const int col = d < 18 ? a_lo + 2 * d : a_hi + 2 * (d - 18);
			reinterpret_cast<uint32_t *>(T)[row * (LQ_LS / 2) + d] = (col >= 0 && col < S) ? v[j] : 0u;
		}
	}
	lds_barrier();
	{
		/* shrink (:670-721): a lane a ROW (row 64 wv + lane; the tile's pitch of 37 dwords puts 64 rows on different banks): it sorts its row's 72 cells
		 * into "loud" masks (|v| > 8, > diag; a bit a cell of either window), takes the masks of the rows above and below from its neighbour lanes
		 * (the wavefront's first and last row: from the next wavefront, through LDS), and the 3 x 3 rule is mask algebra in registers.  All masks are
		 * made from the values as they were: nothing is applied before every wavefront has read its rows. */
		__shared__ uint64_t s_edge[4][2][4];                           /* a wavefront's first / last row: m8[0], m8[1], md[0], md[1] */
		const int diag = q <= 16 ? 16 : 8, r = 64 * wv + lane;
		uint64_t ok[2], ll[2];                                       /* per window: cells with both neighbours in it and a block column 1 .. 254; cells of the LL2 quadrant's columns */
		for (int w = 0; w < 2; w++) {
			ok[w] = 0; ll[w] = 0;
			for (int x = 1; x < 35; x++) { const int c = (w ? a_hi : a_lo) + x; if (c >= 1 && c <= S - 2) ok[w] |= 1ull << x; if (c < HLF) ll[w] |= 1ull << x; }
		}
		uint64_t m8[2] = { 0, 0 }, md[2] = { 0, 0 };
		{
			const uint32_t *rw = reinterpret_cast<const uint32_t *>(T) + r * (LQ_LS / 2);
#pragma unroll
			for (int d = 0; d < 36; d++) {
				const uint32_t v = rw[d];
				const int a0 = iabs((int)(int16_t)(v & 0xFFFFu)), a1 = iabs((int)(int16_t)(v >> 16));
				const int w = d >= 18, x = 2 * (d - 18 * w);
				m8[w] |= (uint64_t)(a0 > 8) << x | (uint64_t)(a1 > 8) << (x + 1);
				md[w] |= (uint64_t)(a0 > diag) << x | (uint64_t)(a1 > diag) << (x + 1);
			}
		}
		if (lane == 0 || lane == 63) { uint64_t *e = s_edge[wv][lane ? 1 : 0]; e[0] = m8[0]; e[1] = m8[1]; e[2] = md[0]; e[3] = md[1]; }
		lds_barrier();
		uint64_t h[2];
		for (int w = 0; w < 2; w++) {
			auto from = [&](uint64_t v, int dpp_up) -> uint64_t {       /* the value of the lane above (dpp_up) or below */
				const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
				const uint32_t l2 = dpp_up ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x138, 0xF, 0xF, false) : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xF, 0xF, false);
				const uint32_t h2 = dpp_up ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x138, 0xF, 0xF, false) : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x130, 0xF, 0xF, false);
				return (uint64_t)l2 | ((uint64_t)h2 << 32);
			};
			uint64_t p8 = from(m8[w], 1), pd = from(md[w], 1), n8 = from(m8[w], 0), nd = from(md[w], 0);
			if (lane == 0) { p8 = wv ? s_edge[wv - 1][1][w] : 0; pd = wv ? s_edge[wv - 1][1][2 + w] : 0; }
			if (lane == 63) { n8 = wv < 3 ? s_edge[wv + 1][0][w] : 0; nd = wv < 3 ? s_edge[wv + 1][0][2 + w] : 0; }
			const uint64_t c8 = m8[w];
			uint64_t hh = c8 & ~((c8 << 1) | (c8 >> 1) | p8 | n8 | (pd << 1) | (pd >> 1) | (nd << 1) | (nd >> 1)) & ok[w];
			if (r < HLF) hh &= ~ll[w];                               /* not the LL2 quadrant */
			if (r < 1 || r > S - 2) hh = 0;
			h[w] = hh;
		}
		for (int w = 0; w < 2; w++) {
			uint64_t hb = h[w];
			while (hb) {
				const int bt = __builtin_ctzll(hb);
				hb &= hb - 1;
				int16_t *cell = T + r * LQ_LS + 36 * w + bt;
				*cell = (int16_t)(*cell > 0 ? *cell - 1 : *cell + 1);
			}
		}
		lds_barrier();
	}
	{                                                                /* along the rows, un-normalised: half a wavefront a row, a lane the outputs 2 k, 2 k + 1 of k = 32 p + (lane & 31) */
		const int rsub = lane >> 5, kk = lane & 31, k = 32 * p + kk;
#pragma unroll 2
		for (int it = 0; it < 32; it++) {
			int16_t *x = T + (it * 8 + wv * 2 + rsub) * LQ_LS;
			const int l0 = x[kk + 2], ln = k + 1 < HLF ? x[kk + 3] : l0;
			const int h0 = x[36 + kk + 2], hp = k > 0 ? x[36 + kk + 1] : h0, hn = k + 1 < HLF ? x[36 + kk + 3] : h0;
			const int ev = (int16_t)((int16_t)(l0 << 3) - ((h0 + hp) << 1));
			const int od = (int16_t)((int16_t)((l0 + ln) << 2) + (6 * h0 - hp - hn));
			__builtin_amdgcn_wave_barrier();                          /* every lane of the row has read its taps */
			reinterpret_cast<uint32_t *>(x)[kk] = (uint32_t)(uint16_t)ev | ((uint32_t)(uint16_t)od << 16);
		}
	}
	lds_barrier();
	for (int i = 0; i < 16; i++) {                                   /* along the columns, normalised, in place: sample 2k, 2k+1 of column j */
		int16_t *x = T + wv * 16 + i;
		int e[2], o[2];
#pragma unroll
		for (int u = 0; u < 2; u++) synth_pair<S>(x, LQ_LS, lane + 64 * u, true, e[u], o[u]);
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int u = 0; u < 2; u++) { const int k = lane + 64 * u; x[(2 * k) * LQ_LS] = (int16_t)e[u]; x[(2 * k + 1) * LQ_LS] = (int16_t)o[u]; }
	}
	lds_barrier();
	{                                                                /* residual lists (:731-787): plane cell (row, col) sits at [col][row - 64 p]; a quarter takes the cells of its own plane rows */
		const uint8_t *f = ws.blob + ws.blob_off[img];
#define ACC(row, col, d) do { const int r_ = (row); if ((r_ >> 6) == p) add_i16_at(T, (col) * LQ_LS + (r_ & 63), (d)); } while (0)
		/* a thread's entries of a list eight at a time: their loads (the entry, the byte its sign bits sit in) go out together, then the adds */
		auto scan = [&](const uint16_t *pp, int cnt, const uint8_t *bits, int nbytes, int per_byte_shift, auto &&apply) {
			for (int k0 = t; k0 < cnt; k0 += 256 * 8) {
				int e[8], by[8];
#pragma unroll
				for (int j = 0; j < 8; j++) {
					const int k = k0 + 256 * j, kc = k < cnt ? k : cnt - 1, bi = kc >> per_byte_shift;
					e[j] = pp[kc]; by[j] = bi < nbytes ? bits[bi] : 0;
				}
#pragma unroll
				for (int j = 0; j < 8; j++) { const int k = k0 + 256 * j; if (k < cnt) apply(k, e[j], by[j]); }
			}
		};
		if (q >= 21) scan(ws.buf<uint16_t>(D_P5, img), (m->res5_bits - 1) * 8, f + m->o_res5_word, m->res5_bits, 3,
		                  [&](int k, int ps, int byte) { ACC(ps >> 8, ps & 255, ((byte >> (7 - (k & 7))) & 1) ? -3 : 3); });
		if (q > 12) {
			const int amp = q >= 18 ? 5 : q >= 15 ? 7 : 9;
			scan(ws.buf<uint16_t>(D_P1, img), (m->res1_bits - 1) * 8, f + m->o_res1_word, m->res1_bits, 3,
			     [&](int k, int ps, int byte) { ACC(ps >> 8, ps & 255, ((byte >> (7 - (k & 7))) & 1) ? -amp : amp); });
		}
		if (q >= 19)
			scan(ws.buf<uint16_t>(D_P3, img), (m->res3_bits * 2 - 2) * 4, f + m->o_res3_word, 1 << 30, 2,
			     [&](int k, int ps, int byte) {
				const int row = ps >> 8, col = ps & 255;
				const int sel = (byte >> (6 - 2 * (k & 3))) & 3;
				const int d0 = sel == 1 ? -4 : sel == 0 ? 4 : sel == 2 ? 2 : -2, d1 = sel == 1 ? -3 : sel == 0 ? 3 : sel == 2 ? 2 : -2, d2 = sel == 2 ? 2 : sel == 3 ? -2 : 0;
				ACC(row, col, d0);
				if (row + 1 < DH) ACC(row + 1, col, d1);                /* (rows 254 / 255 reach below the level-1 LL: dropped, see k_dec_luma_l2) */
				if (d2 && row + 2 < DH) ACC(row + 2, col, d2);
			     });
#undef ACC
	}
	lds_barrier();
	for (int i = 0; i < 16; i++) {                                   /* column j of the tile is row 64 p + j of the plane */
		const int j = wv * 16 + i;
		uint32_t *dst = reinterpret_cast<uint32_t *>(pl + (size_t)(64 * p + j) * DW);
#pragma unroll
		for (int u = 0; u < 2; u++) {
			const int k = lane + 64 * u;
			dst[k] = (uint32_t)(uint16_t)T[(2 * k) * LQ_LS + j] | ((uint32_t)(uint16_t)T[(2 * k + 1) * LQ_LS + j] << 16);
		}
	}
}

/* ---------------------------------------------------------------------------------------------- a chroma plane up to its level-1 synthesis
 * One launch, one LDS residency of the 256 x 256 block per (file, component): the block is BUILT in LDS -- zeros, the component's entries of
 * the walk's value list (stream order -> cells: strips of 8 columns, serpentine, U on even and V on odd positions, nhw_decoder.c:904-932 /
 * :1192-1220), the LL2 samples and the exception samples (:943-981, :1231-1267) -- then level 2 of the filterbank on its 128 x 128 corner in
 * place, the pair / single corrections carried as symbols in the level-1 detail bands (:992-1083) onto the level-1 LL, level 1, and column
 * c of the result leaves as row c of the plane.  Before: five kernels and four round trips of the plane.  After the in-place level 2 the
 * corner holds sample (row r, column c) of the level-1 LL at [c][r] -- which is how level 1 wants it (its row pass reads line c of the
 * transposed LL) and where the corrections, given in plane coordinates, are added.  `upto`: the debug stop (1: the block as built, 2: after
 * level 2, 3: after the corrections -- written back in the plane's layout; 4: everything). */
__global__ __launch_bounds__(1024) void k_dec_chroma(DecWs ws, int items, int upto)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	constexpr int S = DH, LS = S + 2, HLF = S / 2, NT_ = 1024;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	for (int item = blockIdx.x; item < items; item += gridDim.x) {
		const int img = item >> 1, comp = item & 1;
		const DecMeta *m = ws.buf<DecMeta>(D_META, img);
		if (m->status) continue;                                     /* (uniform: the whole workgroup skips the file) */
		const int q = m->q;
		int16_t *pl = plane_ca(ws, img, comp);
		const uint8_t *f = ws.blob + ws.blob_off[img];
#pragma unroll 4
		for (int k = t; k < S * LS / 2; k += NT_) reinterpret_cast<uint32_t *>(smem)[k] = 0;
		lds_barrier();
		{                                                            /* the component's values */
			const uint32_t *ent = ws.buf<uint32_t>(D_CB, img);
			const int n_ent = (int)ws.buf<uint32_t>(D_SEG, img)[SEG_CHROMA + 128];
			for (int k = t; k < n_ent; k += NT_) {
				const uint32_t en = ent[k];
				const int pos = ENT_POS(en);
				if ((pos & 1) != comp) continue;
				const int w = pos >> 1, strip = w >> 11, rp = (w & 2047) >> 4, idx = w & 15;
				smem[(2 * rp + (idx >> 3)) * LS + 8 * strip + ((idx & 8) ? 15 - idx : (idx & 7))] = (int16_t)ENT_VAL(en);
			}
		}
		lds_barrier();
		{                                                            /* LL2 samples (64 x 64) */
			const uint8_t *l = ws.buf<uint8_t>(D_LL, img) + DQ / 4 + (comp ? DQ / 16 : 0);
#pragma unroll 1
			for (int k = t; k < DQ / 16; k += NT_) smem[(k >> 6) * LS + (k & 63)] = (int16_t)(l[k] + (q > 15 ? 0 : 1));
		}
		lds_barrier();
		if (!t) {                                                    /* exception samples: luma, then U, then V share one cursor */
			const uint8_t *x = f + m->o_exw;
			const int n = m->exw_len;
#define XB(k) ((k) < n ? (int)x[k] : 0)
			int i = 0;
			for (; i < n; i += 3) if (!XB(i) && !XB(i + 1)) break;
			i += 2;
			if (comp) { for (; i < n; i += 3) if (!XB(i) && !XB(i + 1)) break; i += 2; }
			for (; i < n; i += 3) {
				if (!comp && !XB(i) && !XB(i + 1)) break;
				const int hi = XB(i + 1) >= 128, lo = XB(i + 1) & 127;
				smem[XB(i) * LS + lo] = (int16_t)(hi ? XB(i + 2) + 255 : -XB(i + 2));
			}
#undef XB
		}
		lds_barrier();
		/* the block back to the plane for the debug stop; llt: the 128 x 128 corner sits transposed in LDS */
		auto store_block = [&](bool llt) {
#pragma unroll 1
			for (int v = t; v < S * S; v += NT_) {
				const int r = v >> 8, c = v & 255;
				pl[(size_t)r * DH + c] = (llt && r < HLF && c < HLF) ? smem[c * LS + r] : smem[r * LS + c];
			}
		};
		if (upto == 1) { store_block(false); lds_barrier(); continue; }
#pragma unroll 1
		for (int i = 0; i < 8; i++) {                                /* level 2, along the rows of the corner (un-normalised) */
			int16_t *x = smem + (wv * 8 + i) * LS;
			int e, o;
			synth_pair<HLF>(x, 1, lane, false, e, o);
			reinterpret_cast<uint32_t *>(x)[lane] = (uint32_t)(uint16_t)e | ((uint32_t)(uint16_t)o << 16);
		}
		lds_barrier();
#pragma unroll 1
		for (int i = 0; i < 8; i++) {                                /* along its columns, normalised, in place */
			int16_t *x = smem + wv * 8 + i;
			int e, o;
			synth_pair<HLF>(x, LS, lane, true, e, o);
			x[(2 * lane) * LS] = (int16_t)e; x[(2 * lane + 1) * LS] = (int16_t)o;
		}
		lds_barrier();
		if (upto == 2) { store_block(true); lds_barrier(); continue; }
		{                                                            /* the corrections: a symbol 5003..5006 in a detail band steps the level-1 LL cell(s) it sits over.  Such a symbol can only
		                                                              * have come from the value list (LL2 and exception samples are small), so the list is gone through again instead of
		                                                              * the 49 152 detail cells; a cell an exception sample has overwritten since no longer holds its symbol and is left alone */
			const uint32_t *ent = ws.buf<uint32_t>(D_CB, img);
			const int n_ent = (int)ws.buf<uint32_t>(D_SEG, img)[SEG_CHROMA + 128];
			for (int k = t; k < n_ent; k += NT_) {
				const uint32_t en = ent[k];
				const int pos = ENT_POS(en), sy = ENT_VAL(en);
				if ((pos & 1) != comp || sy < 5003 || sy > 5006) continue;
				const int w = pos >> 1, strip = w >> 11, rp = (w & 2047) >> 4, idx = w & 15;
				const int i = 2 * rp + (idx >> 3), j = 8 * strip + ((idx & 8) ? 15 - idx : (idx & 7));
				if ((i < HLF && j < HLF) || smem[i * LS + j] != sy) continue;
				const int ti = i < HLF ? i : i - HLF, tj = j < HLF ? j : j - HLF;   /* plane cell (ti, tj) of the LL sits at [tj][ti] */
				const bool two = tj < HLF - 1;                            /* the second cell of a pair at the last LL column is scratch in the reference */
				const int d = sy == 5005 ? -4 : sy == 5006 ? 4 : sy == 5003 ? -6 : 6;
				add_i16_at(smem, tj * LS + ti, d);
				if (two && (sy == 5005 || sy == 5006)) add_i16_at(smem, (tj + 1) * LS + ti, d);
				smem[i * LS + j] = 0;
			}
		}
		lds_barrier();
		if (upto == 3) { store_block(true); lds_barrier(); continue; }
#pragma unroll 1
		for (int i = 0; i < 16; i++) {                               /* level 1, along the rows */
			int16_t *x = smem + (wv * 16 + i) * LS;
			int e[2], o[2];
#pragma unroll
			for (int u = 0; u < 2; u++) synth_pair<S>(x, 1, lane + 64 * u, false, e[u], o[u]);
#pragma unroll
			for (int u = 0; u < 2; u++) reinterpret_cast<uint32_t *>(x)[lane + 64 * u] = (uint32_t)(uint16_t)e[u] | ((uint32_t)(uint16_t)o[u] << 16);
		}
		lds_barrier();
#pragma unroll 1
		for (int i = 0; i < 16; i++) {                               /* along the columns, normalised: column c is row c of the plane */
			const int c = wv * 16 + i;
			uint32_t *dst = reinterpret_cast<uint32_t *>(pl + (size_t)c * DH);
#pragma unroll
			for (int u = 0; u < 2; u++) {
				int e, o;
				synth_pair<S>(smem + c, LS, lane + 64 * u, true, e, o);
				dst[lane + 64 * u] = (uint32_t)(uint16_t)e | ((uint32_t)(uint16_t)o << 16);
			}
		}
		lds_barrier();                                               /* the block is done with before the next one is built */
	}
}

#define SYNTH_WGS 256                /* one resident workgroup per CU for the 256 x 256 blocks */
static int synth2d_attrs()
{
	int rc = NHW_OK;                                               /* per device: every handle sets it for its own */
	const int big = 256 * 258 * (int)sizeof(int16_t);
	if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dec_chroma), hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess ||
	    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dec_luma_l2), hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) rc = NHW_E_HIP;
	return rc;
}

/* ---------------------------------------------------------------------------------------------- smooth-edge marks (:789-848)
 * The reference marks a sample by adding 16000 to it in place while it walks the level-1 LL in raster order, so a
 * mark shows up (as -16000) in the Laplacians of the pairs visited after it: the row above (three cells) and the
 * cell to the left.  One wavefront per image walks the rows in order; a lane owns two pairs (cells 4l+1 .. 4l+4),
 * evaluates them for both states of the cell on its left, and the chain of "my last cell is marked" along the row
 * is settled on the ballots.  Output: the marked positions in raster order. */
DEV void pair_decide(int r0, int r1, int &m0, int &m1)
{
	/* (as an if / else-if chain that sets one of the two, the compiler made the pair an array in scratch memory) */
	m0 = (r0 > 41 && r0 < 108 && r1 < 16) || (r0 < -41 && r0 > -108 && r1 > -16);
	m1 = !m0 && ((r1 > 41 && r1 < 108 && r0 < 16) || (r1 < -41 && r1 > -108 && r0 > -16));
}
__global__ __launch_bounds__(256) void k_dec_marks(DecWs ws)
{
	const int img = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (img >= ws.n) return;
	DecMeta *m = ws.buf<DecMeta>(D_META, img);
	if (m->status) return;
	const int16_t *c = plane_a(ws, img);
	uint16_t *marks = ws.buf<uint16_t>(D_MARKS, img);
	uint16_t *rowstart = mark_rows(ws, img);                        /* [257]: index of the first mark of row i (the list is in row order) */
	int total = 0;
	if (lane < 2) rowstart[lane] = 0;
	unsigned above = 0;                                             /* marks of the row above on my cells 4l+1..4l+4 (bits 0..3) */
	/* A lane's Laplacians need columns 4l .. 4l+5 of three rows: one 16-byte load per row (4l .. 4l+7), the rows rolling through registers
	 * and MK_AHEAD more on their way -- the marks chain runs down the rows, the loads do not depend on it.  (Nine 2-byte loads per cell and
	 * a wait per row were 2 us a row.) */
	struct __attribute__((aligned(8))) Row8 { uint32_t w[4]; };
	auto ld = [&](int r) { return *reinterpret_cast<const Row8 *>(c + (size_t)(r < DW ? r : DW - 1) * DW + 4 * lane); };
	auto cells = [](const Row8 &r, int *v) {
#pragma unroll
		for (int e = 0; e < 3; e++) { v[2 * e] = (int16_t)(r.w[e] & 0xFFFFu); v[2 * e + 1] = (int16_t)(r.w[e] >> 16); }
	};
#define MK_AHEAD 4
	int up[6], mid[6], dn[6];
	{ const Row8 r0 = ld(0), r1 = ld(1); cells(r0, up); cells(r1, mid); }
	Row8 fly[MK_AHEAD];
#pragma unroll
	for (int g = 0; g < MK_AHEAD; g++) fly[g] = ld(2 + g);
	for (int i = 1; i < DH - 1; i++) {
		cells(fly[0], dn);
#pragma unroll
		for (int g = 0; g + 1 < MK_AHEAD; g++) fly[g] = fly[g + 1];
		fly[MK_AHEAD - 1] = ld(i + 1 + MK_AHEAD);
		int cs[6], base[4];
#pragma unroll
		for (int x = 0; x < 6; x++) cs[x] = up[x] + mid[x] + dn[x];
#pragma unroll
		for (int k = 0; k < 4; k++) base[k] = (lane < 63 || k < 2) ? 9 * mid[k + 1] - (cs[k] + cs[k + 1] + cs[k + 2]) : 0;
#pragma unroll
		for (int x = 0; x < 6; x++) { up[x] = mid[x]; mid[x] = dn[x]; }
		/* marks above cells 4l .. 4l+5 as bits 0..5 */
		const unsigned fromL = (unsigned)from_left((int)above, 0), fromR = (unsigned)__builtin_amdgcn_update_dpp(0, (int)above, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
		const unsigned ab = (lane ? (fromL >> 3) & 1u : 0u) | (above << 1) | (lane < 63 ? (fromR & 1u) << 5 : 0u);
		int cont[4];
#pragma unroll
		for (int k = 0; k < 4; k++) cont[k] = (int)((ab >> k) & 1u) + (int)((ab >> (k + 1)) & 1u) + (int)((ab >> (k + 2)) & 1u);
		int res[2][4];                                               /* [left cell marked][m0A, m1A, m0B, m1B] */
#pragma unroll
		for (int ml = 0; ml < 2; ml++) {
			int a0, a1, b0, b1;
			pair_decide(base[0] - 16000 * (cont[0] + ml), base[1] - 16000 * cont[1], a0, a1);
			pair_decide(base[2] - 16000 * (cont[2] + a1), base[3] - 16000 * cont[3], b0, b1);
			if (lane == 63) { b0 = 0; b1 = 0; }                       /* pair 127 (cells 255, 256) does not exist */
			res[ml][0] = a0; res[ml][1] = a1; res[ml][2] = b0; res[ml][3] = b1;
		}
		const uint64_t o0 = __ballot(res[0][3]), o1 = __ballot(res[1][3]);
		uint64_t x = o0;
		for (;;) { const uint64_t need = x << 1, xn = (o0 & ~need) | (o1 & need); if (xn == x) break; x = xn; }
		const int ml = lane ? (int)((x >> (lane - 1)) & 1ull) : 0;
		const unsigned mine0 = (unsigned)res[0][0] | ((unsigned)res[0][1] << 1) | ((unsigned)res[0][2] << 2) | ((unsigned)res[0][3] << 3);
		const unsigned mine1 = (unsigned)res[1][0] | ((unsigned)res[1][1] << 1) | ((unsigned)res[1][2] << 2) | ((unsigned)res[1][3] << 3);
		const unsigned mine = ml ? mine1 : mine0;                       /* (indexing res[] with ml put the array in scratch memory) */
		/* ordered output */
		const int cnt = __popc(mine);
		int pre = cnt;
		pre = wscan_add(pre);
		int at = total + pre - cnt;
#pragma unroll
		for (int k = 0; k < 4; k++) if ((mine >> k) & 1u) marks[at++] = (uint16_t)(i * DH + 4 * lane + 1 + k);
		total += last_lane(pre);
		above = mine;
		if (!lane) rowstart[i + 1] = (uint16_t)total;
	}
	if (!lane) { m->nmarks = total; rowstart[DH] = (uint16_t)total; }
#undef MK_AHEAD
}

/* ---------------------------------------------------------------------------------------------- chroma
 */
/* sharpen (:1097-1121): in place and in raster order -- a cell sees the new values of its left and upper neighbours.
 * One wavefront per plane, rows in order, a lane owns four consecutive cells; along a row the only thing that travels is
 * the change (0, +-2, +-3) of the cell on the left, settled by re-evaluating until no lane's outgoing change moves. */
__global__ __launch_bounds__(256) void k_dec_sharpen(DecWs ws)
{
	const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int img = wv >> 1, comp = wv & 1;
	if (img >= ws.n) return;
	const DecMeta *m = ws.buf<DecMeta>(D_META, img);
	if (m->status) return;
	const int thr = m->q <= 14 ? 35 : 60;
	int16_t *b = plane_ca(ws, img, comp);
	uint8_t *out = ws.buf<uint8_t>(D_CU, img) + (size_t)comp * DQ;
	int up[6], cur[6], dn[6];                                       /* cells 4l-1 .. 4l+4 of rows i-1 (already sharpened), i, i+1 */
	/* a lane loads its own four cells of a row with one 8-byte load, SH_AHEAD rows ahead of the one in work (the walk down the rows is a
	 * chain, the loads are not part of it), and takes the cell on either side from its neighbours */
	auto ld = [&](int r) { return *reinterpret_cast<const uint2 *>(b + (size_t)(r < DH ? r : DH - 1) * DH + 4 * lane); };
	auto spread = [&](const uint2 &w, int *dst) {
		dst[1] = (int16_t)(w.x & 0xFFFFu); dst[2] = (int16_t)(w.x >> 16); dst[3] = (int16_t)(w.y & 0xFFFFu); dst[4] = (int16_t)(w.y >> 16);
		const int l = from_left(dst[4], 0), r = __builtin_amdgcn_update_dpp(0, dst[1], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
		dst[0] = lane ? l : 0; dst[5] = lane < 63 ? r : 0;
	};
	auto put = [&](int r, const int *v /* [4] */) {
		*reinterpret_cast<uint32_t *>(out + (size_t)r * DH + 4 * lane) = (uint32_t)clip8(v[0]) | ((uint32_t)clip8(v[1]) << 8) | ((uint32_t)clip8(v[2]) << 16) | ((uint32_t)clip8(v[3]) << 24);
	};
#define SH_AHEAD 4
	{ const uint2 r0 = ld(0), r1 = ld(1); spread(r0, up); spread(r1, cur); }
	uint2 fly[SH_AHEAD];
#pragma unroll
	for (int g = 0; g < SH_AHEAD; g++) fly[g] = ld(2 + g);
	put(0, up + 1);
	for (int i = 1; i < DH - 1; i++) {
		spread(fly[0], dn);
#pragma unroll
		for (int g = 0; g + 1 < SH_AHEAD; g++) fly[g] = fly[g + 1];
		fly[SH_AHEAD - 1] = ld(i + 1 + SH_AHEAD);
		int nw[4], din = 0;
		for (;;) {
			int left = cur[0] + din;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int c = 4 * lane + k;
				const int x = cur[k + 1];
				int v = x;
				if (c >= 1 && c <= DH - 2) {
					const int r = (x << 3) - left - cur[k + 2] - up[k + 1] - dn[k + 1] - up[k] - dn[k] - up[k + 2] - dn[k + 2];
					if (r > thr) v = x + (r > 160 ? 3 : 2); else if (r < -thr) v = x - (r < -160 ? 3 : 2);
				}
				nw[k] = v; left = v;
			}
			const int ndin = from_left(nw[3] - cur[4], 0);
			if (!__any(ndin != din)) break;
			din = ndin;
		}
		/* the sharpened row becomes `up`: own cells, plus the edge cells of the neighbours */
		const int l = from_left(nw[3], 0), r = __builtin_amdgcn_update_dpp(0, nw[0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
		up[0] = lane ? l : 0; up[5] = lane < 63 ? r : 0;
#pragma unroll
		for (int k = 0; k < 4; k++) up[k + 1] = nw[k];
		put(i, nw);
#pragma unroll
		for (int k = 0; k < 6; k++) cur[k] = dn[k];
	}
	put(DH - 1, cur + 1);
#undef SH_AHEAD
}

/* ---------------------------------------------------------------------------------------------- colour (d5 tail + d6)
 * x2 bilinear chroma (:1150-1196: columns of rows first, each rounded to a byte, then along the rows) and the colour
 * matrix of write_image_bmp (nhw_decoder_cli.c:135-286); output bytes in the order the reference writes them */
__constant__ float k_inv_low[17] = { 0.0f, 2.060881f, 1.985939f, 1.916257f, 1.820444f, 1.741126f, 1.665887f, 1.587597f, 1.521263f,
	1.392014f, 1.281502f, 1.190611f, 1.177434f, 1.186945f, 1.138331f, 1.048174f, 1.012139f };
/* bits 32..47 of a 24 x 24 bit product on the full-rate multiplier (operands must fit 24 bits) */
DEV unsigned mulhi_u24(unsigned a, unsigned b) { unsigned r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV void yuv_to_bytes(int q, int yv, int uv, int vv, int &R, int &G, int &B)
{
	if (q >= 20) {
		/* The reference evaluates Y + 1.402 V + 0.5 and its two sisters in double and truncates.  The exact values are multiples of 1/1000
		 * (1/100000 for G) and the rounding errors are some 1e-13, so the truncation can only differ from the integer quotient below where
		 * the exact value IS an integer: never for R, harmlessly for B, and for G at 493 of the 2^24 triples -- those take the double path
		 * (checked against the double evaluation over all 2^24 triples).  Negative values and values from 256 clip either way. */
		/* every product fits the full-rate 24-bit multiplier; the -128 of U and V sit in the constants; one v_med3 clamps a numerator to the
		 * range whose quotient is 0 .. 255; n / 1000 = (n * 4294968) >> 32 for n < 256000 and n / 100000 = ((n >> 5) * 1374390) >> 32 for
		 * n < 25.6e6 (floor(floor(n / 32) / 3125)): the quotient is the multiplier's high half, no shift behind it
		 * (tests/test_dec_colour_formula.py evaluates exactly this arithmetic against the double form on all 2^24 triples) */
		const int nR = __mul24(vv, 1402) + (__mul24(yv, 1000) + (500 - 1402 * 128));
		const int nB = __mul24(uv, 1772) + (__mul24(yv, 1000) + (500 - 1772 * 128));
		const int nG = __mul24(yv, 100000) - __mul24(uv, 34414) - __mul24(vv, 71414) + (50000 + (34414 + 71414) * 128);
		R = (int)mulhi_u24((unsigned)min(max(nR, 0), 255999), 4294968u);
		B = (int)mulhi_u24((unsigned)min(max(nB, 0), 255999), 4294968u);
		const int gq = (int)mulhi_u24((unsigned)min(max(nG, 0), 25599999) >> 5, 1374390u);
		G = gq;
		if (nG > 0 && __mul24(gq, 100000) == nG) G = clip8((int)(yv - 0.34414 * (uv - 128) - 0.71414 * (vv - 128) + 0.5f));
	}
	else if (q >= 18) {
		const float yinv = q == 19 ? 1.025641f : 1.075269f;
		const float Yq = (float)(yv * yinv);
		const int U = uv - 128, V = vv - 128;
		R = (int)(Yq + 1.402 * V + 0.5f); G = (int)(Yq - 0.34414 * U - 0.71414 * V + 0.5f); B = (int)(Yq + 1.772 * U + 0.5f);
	}
	else if (q == 17) {
		const float yinv = 1.063830f;
		const int Y = yv, U = uv - 128, V = vv - 128;
		R = (int)((Y + 1.402 * V) * yinv + 0.5f); G = (int)((Y - 0.34414 * U - 0.71414 * V) * yinv + 0.5f); B = (int)((Y + 1.772 * U) * yinv + 0.5f);
	}
	else {
		const float yinv = k_inv_low[q];
		const int Y = yv * 298, U = uv, V = vv;
		R = ((int)((Y + 409 * V + (-56992 - 128)) * yinv + 128.5f)) >> 8;
		G = ((int)((Y - 100 * U - 208 * V + (34784 - 128)) * yinv + 128.5f)) >> 8;
		B = ((int)((Y + 516 * U + (-70688 - 128)) * yinv + 128.5f)) >> 8;
	}
	if (q < 20) { R = clip8(R); G = clip8(G); B = clip8(B); }
}
/* Quality >= 20, eight pixels at a time, in single precision -- exact by construction, like the encoder's conversion (nhw_front_image.h):
 *   R = trunc(Y + 1.402 V' + 0.5) is floor(nR / 1000), nR = 1000 Y + 1402 V + c an integer below 2^24; v_cvt_pk_u8_f32 rounds to nearest even and
 *   saturates to 0 .. 255, so the floor is rne(nR * 0.001f + (c / 1000 - 0.5 + 5e-4)): the true fractions are multiples of 1e-3, the float error
 *   stays below 7e-5, a negative numerator clips to 0 either way.  B likewise; R and B run as one packed fma pair.
 *   G = floor(Y + 0.5 - w / 100000), w = 34414 U' + 71414 V' (|w| < 2^24 with the -128 taken first): its fractions are multiples of 1e-5, less than
 *   the float error (2.5e-5), so a value that comes out within 1e-4 of an integer boundary is done again in the integer / double form of
 *   yuv_to_bytes (7 triples in 100 000); everything else is the float's rne(. - 0.5).
 * One v_cvt_pk_u8_f32 converts, clips and drops a byte into its place of the 24 output bytes.  y: 8 luma bytes; u / v: 8 bytes each (x2 chroma done). */
typedef float f32x2_ __attribute__((ext_vector_type(2)));
DEV float ubf_(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }            /* v_cvt_f32_ubyte<k> */
DEV void colour8_q20(uint2 y8, uint2 u8, uint2 v8, uint32_t w[6])
{
#pragma unroll
	for (int e = 0; e < 6; e++) w[e] = 0;
	float worst = 0.f;                                                  /* how close a G of the eight comes to an integer boundary */
#pragma unroll
	for (int px = 0; px < 8; px++) {
		const uint32_t yw = px < 4 ? y8.x : y8.y, uw = px < 4 ? u8.x : u8.y, vw = px < 4 ? v8.x : v8.y;
		const float yf = ubf_(yw, px & 3), uf = ubf_(uw, px & 3), vf = ubf_(vw, px & 3);
		/* R, B: Y + 1.402 V' + 0.5 with everything constant in the addend: the product's error (1.3e-5), the fma's rounding below 512 (3e-5) and the
		 * addend's (2.3e-5) stay a factor of seven inside the 5e-4 between the values' fractions (multiples of 1e-3) and the rounding boundary */
		const f32x2_ add = (f32x2_){ yf, yf } + (f32x2_){ (500.f - 1402.f * 128.f) / 1000.f - 0.4995f, (500.f - 1772.f * 128.f) / 1000.f - 0.4995f };
		const f32x2_ xrb = __builtin_elementwise_fma((f32x2_){ vf, uf }, (f32x2_){ 1.402f, 1.772f }, add);
		const f32x2_ uvc = (f32x2_){ uf, vf } - (f32x2_){ 128.f, 128.f };
		const float wg = __builtin_fmaf(uvc.y, 71414.f, uvc.x * 34414.f);
		const float xg = __builtin_fmaf(wg, -1e-5f, yf);                    /* the value the floor is taken of, minus 0.5 */
		const int b0 = 3 * px;
		w[b0 >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(xrb.x, b0 & 3, w[b0 >> 2]);
		w[(b0 + 2) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(xrb.y, (b0 + 2) & 3, w[(b0 + 2) >> 2]);
		w[(b0 + 1) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(xg, (b0 + 1) & 3, w[(b0 + 1) >> 2]);
		worst = __builtin_fmaxf(worst, __builtin_fabsf(xg - __builtin_rintf(xg)));
	}
	if (worst > 0.5f - 1e-4f) {                                         /* one of the eight is too close to a boundary for single precision (7 triples in 100 000): all eight G again, in the integer / double form */
#pragma unroll
		for (int px = 0; px < 8; px++) {
			const uint32_t yw = px < 4 ? y8.x : y8.y, uw = px < 4 ? u8.x : u8.y, vw = px < 4 ? v8.x : v8.y;
			int R, G, B;
			yuv_to_bytes(20, (int)((yw >> (8 * (px & 3))) & 255u), (int)((uw >> (8 * (px & 3))) & 255u), (int)((vw >> (8 * (px & 3))) & 255u), R, G, B);
			const int b1 = 3 * px + 1;
			w[b1 >> 2] = (w[b1 >> 2] & ~(0xFFu << (8 * (b1 & 3)))) | ((uint32_t)G << (8 * (b1 & 3)));
		}
	}
}
/* byte-wise (a + b + 1) >> 1 on four bytes (v_lerp_u8) */
DEV uint32_t avg_up4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
/* x2 along a row (nhw_decoder.c:1150-1196): four chroma samples c0 (and the one behind them, low byte of c1) -> eight: the sample, then its mean with the next */
DEV uint2 chroma_x2(uint32_t c0, uint32_t c1)
{
	const uint32_t odd = avg_up4(c0, __builtin_amdgcn_alignbyte(c1, c0, 1));
	return make_uint2(__builtin_amdgcn_perm(odd, c0, 0x05010400u), __builtin_amdgcn_perm(odd, c0, 0x07030602u));
}
/* developer / test hook: the colour matrix of quality q on n (Y, U, V) byte triples, eight to a thread, through the functions k_dec_final runs */
__global__ void k_dec_colour_probe(const uint8_t *__restrict__ yuv, uint8_t *__restrict__ rgb, int n8, int q)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n8) return;
	const uint8_t *p = yuv + (size_t)i * 24;
	uint8_t y[8], u[8], v[8];
	for (int e = 0; e < 8; e++) { y[e] = p[3 * e]; u[e] = p[3 * e + 1]; v[e] = p[3 * e + 2]; }
	uint32_t w[6];
	if (q >= 20) {
		auto pk = [](const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); };
		colour8_q20(make_uint2(pk(y), pk(y + 4)), make_uint2(pk(u), pk(u + 4)), make_uint2(pk(v), pk(v + 4)), w);
	} else {
		for (int e = 0; e < 6; e++) w[e] = 0;
		for (int px = 0; px < 8; px++) {
			int R, G, B;
			yuv_to_bytes(q, y[px], u[px], v[px], R, G, B);
			const int b0 = 3 * px;
			w[b0 >> 2] |= (uint32_t)R << (8 * (b0 & 3)); w[(b0 + 1) >> 2] |= (uint32_t)G << (8 * ((b0 + 1) & 3)); w[(b0 + 2) >> 2] |= (uint32_t)B << (8 * ((b0 + 2) & 3));
		}
	}
	for (int e = 0; e < 6; e++) reinterpret_cast<uint32_t *>(rgb + (size_t)i * 24)[e] = w[e];
}

/* ---------------------------------------------------------------------------------------------- final reconstruction, one kernel
 * Level-1 synthesis in both directions (decoder/wavelet_filterbank.c:52-357 as driven by nhw_decoder.c), the q > 21 corrections on the
 * plane between them (wavelet_filterbank.c:301-347), the 5-tap smoothing at the marked samples (nhw_decoder.c:859-876), the doubling of
 * the chroma planes and the colour matrix (nhw_decoder_cli.c:133-283) for a band of FR output rows: the intermediate plane and the luma
 * bytes never travel.
 *
 * The second direction is pointwise in the output row r': Y[r'][2j], Y[r'][2j+1] come from T[j][r'], T[256+j][r'] and their neighbours in j,
 * where T is the first direction's output (the reference's transposed plane).  T[k][2m], T[k][2m+1] in turn come from line k of plane A:
 * its low half (for k < 256 the level-1 LL, kept transposed: A[m][k]) and high half around m.  So a band needs, of every one of the 512
 * lines of A, the dozen coefficients around m = r0/2 -- 24-byte pieces, loaded straight into the registers of the thread that owns the
 * line; the four bands that share a 64-byte sector run next to each other on one XCD (same workgroup order as the encoder's front
 * kernel), so the sector is fetched from HBM once.  Only T (18 KB), the band's luma bytes and nine rows of either chroma plane live in LDS.
 * The smoothing reads the rows above and below a mark; marks sit in even rows only (k_dec_marks), and the rows next to them are never
 * marked, so a band needs one extra row (r0 - 1) and sees it as the corrections left it.  The list is in row order and k_dec_marks
 * leaves the index of every row's first mark, so a band knows its share. */
#define FR 16                        /* output rows per workgroup */
#define FBP (FR + 2)                 /* LDS pitch (shorts) of a line of T: local index l <-> row r0 - 2 + l (l = 0 unused); 9 dwords: no bank conflicts across lines */
#define FM (FR / 2 + 1)              /* values of m a band computes: r0/2 - 1 .. r0/2 + FR/2 - 1 */
#define F_T_BYTES (2 * DH * FBP * 2)
#define F_LDS_BYTES (F_T_BYTES + FR * DW + 2 * (FR / 2 + 1) * DH)   /* T, the luma bytes, the chroma rows: 31 KB, five bands to a CU */
__global__ __launch_bounds__(256) void k_dec_final(DecWs ws, uint8_t *out, int dev_stop /* developer builds: end every band after phase dev_stop (0: run it all) */)
{
#ifdef NHW_DEV
#define F_STOP(i) do { if (dev_stop == (i)) return; } while (0)
#else
#define F_STOP(i) do { } while (0)
#endif
	extern __shared__ __attribute__((aligned(16))) uint8_t fl[];
	int16_t *T = reinterpret_cast<int16_t *>(fl);                               /* [512][FBP] */
	uint8_t *ybuf = fl + F_T_BYTES;                                             /* [FR][512] */
	uint8_t *crow = ybuf + FR * DW;                                             /* [2][FR / 2 + 1][256] */
	const int tid = threadIdx.x;
	int band, img;
	{
		const int nb = DW / FR, total = nb * ws.n, w = blockIdx.x, per = total >> 3;
		const int item = (total & 7) ? w : (w & 7) * per + (w >> 3);            /* workgroup w runs on XCD w % 8: consecutive items stay on one XCD */
		img = item / nb; band = item % nb;
	}
	const DecMeta *m = ws.buf<DecMeta>(D_META, img);
	if (m->status) return;
	const int q = m->q, r0 = FR * band, m0 = r0 / 2 - 1;                       /* m0: first m of the band (-1 in band 0: skipped) */
	const int16_t *A = plane_a(ws, img);

	for (int k = tid; k < 2 * (FR / 2 + 1) * (DH / 16); k += 256) {
		const int pl = k / ((FR / 2 + 1) * (DH / 16)), rem = k % ((FR / 2 + 1) * (DH / 16)), rr = rem / (DH / 16), o = rem % (DH / 16);
		const int src = r0 / 2 + rr < DH ? r0 / 2 + rr : DH - 1;
		reinterpret_cast<uint4 *>(crow + (pl * (FR / 2 + 1) + rr) * DH)[o] = reinterpret_cast<const uint4 *>(ws.buf<uint8_t>(D_CU, img) + (size_t)pl * DQ + (size_t)src * DH)[o];
	}
	/* first direction (decoder/filters.c:143-194 on line k): T[k][2m], T[k][2m + 1].  A thread takes lines tid and tid + 256 and loads what
	 * they need of A straight into registers: of a line's high half (and of the low half of the lines from 256) the twelve coefficients
	 * r0/2 - 2 .. r0/2 + 9 -- one 16-byte load at r0/2 (16-byte aligned) and a dword on either side; the low half of line k < 256 is column
	 * k of rows r0/2 - 1 .. of the level-1 LL, consecutive lanes on consecutive columns.  The few coefficients outside a line's half that
	 * this touches at the first and the last band are never used (row -1 lies in the pad in front of the plane). */
	const uint64_t *nzg = ws.buf<uint64_t>(D_NZG, img);
	/* a line's three groups around column c0 (a multiple of 8): the dword in front, the 16 bytes at c0, the dword behind -- each only if
	 * k_dec_expand stored its group (gm: the line's groups in memory); a group it left out is zeros, and most of the detail bands' are */
	auto piece = [&](const int16_t *line, uint64_t gm, int c0, uint32_t w[6]) {
		const int g = c0 >> 3;
		const int16_t *src = line + c0;
		const bool b0 = g > 0 && ((gm >> (g - 1)) & 1), b1 = (gm >> g) & 1, b2 = g < 63 && ((gm >> (g + 1)) & 1);   /* (what lies outside the line is never used) */
		const uint32_t w0 = b0 ? *reinterpret_cast<const uint32_t *>(src - 2) : 0u, w5 = b2 ? *reinterpret_cast<const uint32_t *>(src + 8) : 0u;
		const uint4 wm = b1 ? *reinterpret_cast<const uint4 *>(src) : make_uint4(0, 0, 0, 0);
		w[0] = w0; w[1] = wm.x; w[2] = wm.y; w[3] = wm.z; w[4] = wm.w; w[5] = w5;
	};
#pragma unroll
	for (int half = 0; half < 2; half++) {
		const int k = tid + DH * half;
		const uint64_t gm = nzg[k];
		int hi[12], lo[12];                                                     /* hi[e], lo[e]: coefficient m0 - 1 + e */
		{
			uint32_t w[6];
			piece(A + (size_t)k * DW, gm, DH + r0 / 2, w);
#pragma unroll
			for (int e = 0; e < 6; e++) { hi[2 * e] = (int16_t)(w[e] & 0xFFFF); hi[2 * e + 1] = (int16_t)(w[e] >> 16); }
		}
		if (half) {
			uint32_t w[6];
			piece(A + (size_t)k * DW, gm, r0 / 2, w);
#pragma unroll
			for (int e = 0; e < 6; e++) { lo[2 * e] = (int16_t)(w[e] & 0xFFFF); lo[2 * e + 1] = (int16_t)(w[e] >> 16); }
		} else {
			lo[0] = 0; lo[11] = 0;
#pragma unroll
			for (int e = 0; e <= FM; e++) lo[1 + e] = A[(ptrdiff_t)(m0 + e) * DW + k];
		}
#pragma unroll
		for (int mi = 0; mi < FM; mi++) {
			const int j = m0 + mi, M = DH;
			if (j < 0) continue;
			const int l0 = lo[1 + mi], l1 = lo[2 + mi], h0 = hi[1 + mi], hm = hi[mi], hp = hi[2 + mi];
			int ev = (int16_t)(l0 << 3), od = j < M - 1 ? (int16_t)((l1 + l0) << 2) : (int16_t)(l0 << 3);
			if (j == 0) { ev -= h0 << 2; od += 5 * h0 - hp; }
			else if (j < M - 1) { ev -= (h0 + hm) << 1; od += 6 * h0 - hp - hm; }
			else { ev -= (h0 + hm) << 1; od += 5 * h0 - hm; }
			*reinterpret_cast<uint32_t *>(T + k * FBP + 2 * mi) = (uint32_t)(uint16_t)ev | ((uint32_t)(uint16_t)od << 16);
		}
	}
	__syncthreads();
	F_STOP(2);

	/* q > 21: the corrections that land in rows r0 - 1 .. r0 + FR - 1 (wavelet_filterbank.c:301-347); positions may repeat */
	if (q > 21) {
		const uint8_t *f = ws.blob + ws.blob_off[img];
		const uint32_t *p6 = ws.buf<uint32_t>(D_P6, img);
		const int cnt = (m->res6_bits - 1) * 8;
#define F_ADD(at, delta) do { const int rr_ = (int)((at) & (DW - 1)) - (r0 - 2); if (rr_ >= 1 && rr_ <= FR + 1) add_i16_at(T, (int)((at) >> 9) * FBP + rr_, (delta)); } while (0)
		for (int k = tid; k < cnt; k += 256) if (p6[k] < 4u * DQ) F_ADD(p6[k], bit_of(f + m->o_res6_word, m->res6_bits, k) ? -32 : 32);
		const uint8_t *cr = f + m->o_char;
		for (int k = tid; k < m->char_res1_len; k += 256) {
			const int v = cr[2 * k] | (cr[2 * k + 1] << 8);
			const int t = v & 3;
			const int at = t == 0 ? (v << 1) + DH - 2 : t == 1 ? ((v - 1) << 1) + DH - 2 : t == 2 ? ((v - 2) << 1) + DH - 1 : ((v - 3) << 1) + DH - 1;
			if (at >= 0 && at < 4 * DQ) F_ADD(at, (t & 1) ? -32 : 32);
		}
		if (q > 22) {
			const uint8_t *qs = f + m->o_qs3;
			for (int k = tid; k < m->qs3_len; k += 256) {
				const uint32_t v = (uint32_t)qs[4 * k] | ((uint32_t)qs[4 * k + 1] << 8) | ((uint32_t)qs[4 * k + 2] << 16) | ((uint32_t)qs[4 * k + 3] << 24);
				if ((v >> 1) < 4u * DQ) F_ADD(v >> 1, (v & 1) ? -56 : 56);
			}
		}
#undef F_ADD
		__syncthreads();
	}

	/* 5-tap smoothing at the marked samples of my rows (:859-876).  List order matters only inside a run of adjacent marks: the head of a run walks it. */
	{
		const int nmarks = m->nmarks;
		const uint16_t *marks = ws.buf<uint16_t>(D_MARKS, img);
		const uint16_t *rows = mark_rows(ws, img);
		const int k_lo = rows[r0 / 2], k_hi = rows[r0 / 2 + FR / 2];
		for (int k = k_lo + tid; k < k_hi; k += 256) {
			if (k > 0 && marks[k - 1] + 1 == marks[k]) continue;
			for (int t = k; t < nmarks && (t == k || marks[t - 1] + 1 == marks[t]); t++) {
				const int l = ((marks[t] >> 8) << 1) - (r0 - 2), col = marks[t] & 255;   /* cell (row, col) of the transposed plane = T[col][row] */
#define TP(dr, dc) ((int)T[(col + (dc)) * FBP + l + (dr)])
				const int ctr = TP(0, 0);
				const int lap = (ctr << 3) - TP(0, -1) - TP(0, 1) - TP(-1, 0) - TP(1, 0) - TP(-1, -1) - TP(1, -1) - TP(-1, 1) - TP(1, 1);
				if (iabs(lap) < 116) T[col * FBP + l] = (int16_t)(((ctr << 2) + TP(0, -1) + TP(0, 1) + TP(-1, 0) + TP(1, 0) + 4) >> 3);
#undef TP
			}
		}
	}
	__syncthreads();
	F_STOP(3);

	/* second direction: two output rows (one dword of every line of T) per step, a thread per j.  Both rows at once in packed 16-bit
	 * arithmetic: the reference's values are int16 with wrap-around at every step, which is what the packed instructions compute; at the two
	 * ends of a line the missing neighbour is the sample itself (that is what the end rules of filters.c:143-194 amount to). */
	for (int lp = 1; lp <= FR / 2; lp++) {                                       /* local rows 2 lp, 2 lp + 1 <-> r0 + 2 (lp - 1), + 1 */
		typedef short s16x2 __attribute__((ext_vector_type(2)));
		const int j = tid, M = DH;
#define TW(k) (*reinterpret_cast<const uint32_t *>(T + (k) * FBP + 2 * lp))
		const uint32_t lo0 = TW(j), h0 = TW(M + j);
		const uint32_t lo1 = j < M - 1 ? TW(j + 1) : lo0, hm = j ? TW(M + j - 1) : h0, hp = j < M - 1 ? TW(M + j + 1) : h0;
#undef TW
		const s16x2 L0 = __builtin_bit_cast(s16x2, lo0), L1 = __builtin_bit_cast(s16x2, lo1);
		const s16x2 C0 = __builtin_bit_cast(s16x2, h0), CM = __builtin_bit_cast(s16x2, hm), CP = __builtin_bit_cast(s16x2, hp);
		s16x2 ev = (L0 << 3) - ((C0 + CM) << 1);
		s16x2 od = ((L1 + L0) << 2) + (C0 * (s16x2)(6) - CP - CM);
		const s16x2 zero = (s16x2)(0), one = (s16x2)(1), top = (s16x2)(255);
		ev = (ev + (__builtin_elementwise_min(__builtin_elementwise_max(ev, zero), one) << 5)) >> 6;   /* + 32 where positive, then the arithmetic shift */
		od = (od + (__builtin_elementwise_min(__builtin_elementwise_max(od, zero), one) << 5)) >> 6;
		ev = __builtin_elementwise_min(__builtin_elementwise_max(ev, zero), top);
		od = __builtin_elementwise_min(__builtin_elementwise_max(od, zero), top);
		*reinterpret_cast<uint16_t *>(ybuf + (2 * lp - 2) * DW + 2 * j) = (uint16_t)((unsigned)ev.x | ((unsigned)od.x << 8));
		*reinterpret_cast<uint16_t *>(ybuf + (2 * lp - 1) * DW + 2 * j) = (uint16_t)((unsigned)ev.y | ((unsigned)od.y << 8));
	}
	__syncthreads();
	F_STOP(4);

	/* colour (nhw_decoder_cli.c:133-283).  Quality >= 20: a thread = EIGHT pixels of one row, four rows a step; the chroma is doubled four bytes
	 * at a time (byte-wise means: v_lerp_u8), the matrix runs in packed single precision (colour8_q20). */
	if (q >= 20) {
		const int t8 = tid & 63;
		const uint8_t *cU = crow, *cV = crow + (FR / 2 + 1) * DH;
		for (int it = 0; it < FR / 4; it++) {
			const int lr = 4 * it + (tid >> 6), r = r0 + lr, ci = lr >> 1, j = 4 * t8;
			const uint2 y8 = *reinterpret_cast<const uint2 *>(ybuf + lr * DW + 8 * t8);
			const bool mean = r < 2 * DH - 2 && (r & 1);            /* odd rows: the mean of the two chroma rows (rows 510, 511: chroma row 255) */
			uint2 c8[2];
#pragma unroll
			for (int pl = 0; pl < 2; pl++) {
				const uint8_t *pc = (pl ? cV : cU) + ci * DH + j;
				uint32_t a0 = *reinterpret_cast<const uint32_t *>(pc), a1 = j + 4 < DH ? *reinterpret_cast<const uint32_t *>(pc + 4) : a0 >> 24;   /* behind the row: its last sample again */
				if (mean) {
					const uint32_t b0 = *reinterpret_cast<const uint32_t *>(pc + DH), b1 = j + 4 < DH ? *reinterpret_cast<const uint32_t *>(pc + DH + 4) : b0 >> 24;
					a0 = avg_up4(a0, b0); a1 = avg_up4(a1, b1);
				}
				c8[pl] = chroma_x2(a0, a1);
			}
			uint32_t w[6];
			colour8_q20(y8, c8[0], c8[1], w);
			uint2 *o = reinterpret_cast<uint2 *>(out + (size_t)img * NHW_IMG_BYTES + (size_t)r * DW * 3 + 24 * t8);   /* consecutive lanes, consecutive 24-byte pieces */
			o[0] = make_uint2(w[0], w[1]); o[1] = make_uint2(w[2], w[3]); o[2] = make_uint2(w[4], w[5]);
		}
	} else {
		const int t = tid & 127;
		const uint8_t *cU = crow, *cV = crow + (FR / 2 + 1) * DH;
		for (int it = 0; it < FR / 2; it++) {
			const int lr = 2 * it + (tid >> 7), r = r0 + lr;
			const uint32_t y4 = *reinterpret_cast<const uint32_t *>(ybuf + lr * DW + 4 * t);
			const int j = 2 * t;
			int tu[3], tv[3];                                         /* the vertically doubled chroma rows at columns j, j+1, j+2 */
#pragma unroll
			for (int c = 0; c < 3; c++) {
				const int jj = j + c < DH ? j + c : DH - 1;
				const int u0 = cU[it * DH + jj], u1 = cU[(it + 1) * DH + jj], v0 = cV[it * DH + jj], v1 = cV[(it + 1) * DH + jj];
				if (r >= 2 * DH - 2 || !(r & 1)) { tu[c] = u0; tv[c] = v0; }       /* rows 510, 511: chroma row 255 */
				else { tu[c] = (u0 + u1 + 1) >> 1; tv[c] = (v0 + v1 + 1) >> 1; }
			}
			uint32_t w[3] = { 0, 0, 0 };
#pragma unroll
			for (int px = 0; px < 4; px++) {
				int uv, vv;
				/* (the last two columns repeat column 255: tu[1] = tu[2] = column 255 there, and both rules give it) */
				if (px & 1) { uv = (tu[px >> 1] + tu[(px >> 1) + 1] + 1) >> 1; vv = (tv[px >> 1] + tv[(px >> 1) + 1] + 1) >> 1; }
				else { uv = tu[px >> 1]; vv = tv[px >> 1]; }
				int R, G, B;
				yuv_to_bytes(q, (int)((y4 >> (8 * px)) & 255u), uv, vv, R, G, B);
				const int b0 = 3 * px;
				w[b0 >> 2] |= (uint32_t)R << (8 * (b0 & 3));
				w[(b0 + 1) >> 2] |= (uint32_t)G << (8 * ((b0 + 1) & 3));
				w[(b0 + 2) >> 2] |= (uint32_t)B << (8 * ((b0 + 2) & 3));
			}
			/* consecutive lanes, consecutive 12-byte pieces: one store instruction writes 768 contiguous bytes of the row */
			*reinterpret_cast<uint3 *>(out + (size_t)img * NHW_IMG_BYTES + (size_t)r * DW * 3 + 12 * t) = make_uint3(w[0], w[1], w[2]);
		}
	}
}

__global__ __launch_bounds__(256) void k_dec_status(DecWs ws, int32_t *status, int32_t *quality)
{
	const int img = blockIdx.x * 256 + threadIdx.x;
	if (img >= ws.n) return;
	const DecMeta *m = ws.buf<DecMeta>(D_META, img);
	status[img] = m->status;
	if (quality) quality[img] = m->q;
}

} /* namespace */

/* ---------------------------------------------------------------------------------------------- host side */
static thread_local std::string g_derr;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { char b_[256]; snprintf(b_, sizeof b_, "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); g_derr = b_; return NHW_E_HIP; } } while (0)
extern "C" const char *nhw_dec_last_error(void) { return g_derr.c_str(); }

struct nhw_dec {
	int device, max_batch;
	DecWs ws;
	size_t slab_bytes;
	hipStream_t own_stream;
	hipStream_t chroma_stream;   /* the chroma sequence runs here, next to the luma one (NHW_CHROMA_FORK=0: behind it, on the caller's stream) */
	hipEvent_t fork_ev, join_ev;
	uint16_t *vlc_table;         /* the prefix code's two-level lookup table (k_dec_vlc_table), 2.5 KB */
	int chroma_fork;
	int stop_after;
	hipEvent_t ev[4];         /* start, after the entropy stages, around the final reconstruction kernel (= end) */
	bool timed;
	/* host convenience path */
	uint8_t *d_blob; size_t blob_cap;
	uint64_t *d_off; uint32_t *d_len; uint8_t *d_out; int32_t *d_status; int32_t *d_quality;
};

extern "C" void nhw_dec_destroy(nhw_dec *d);
extern "C" int nhw_dec_create(int device, int max_batch, nhw_dec **out)
{
	if (!out || max_batch < 1) { g_derr = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(device));
	nhw_dec *d = new nhw_dec();
	memset(d, 0, sizeof *d);
	d->device = device; d->max_batch = max_batch;
	size_t at = 0;
	for (int b = 0; b < D_COUNT; b++) { d->ws.off[b] = at; at += k_dec_bytes[b] * (size_t)max_batch; at = (at + 255) & ~(size_t)255; }
	d->slab_bytes = at;
	const int rc = [&]() -> int {                                  /* a failure half-way leaves nothing behind: the handle is destroyed below */
		size_t free_b = 0, total_b = 0;
		HIPCHK(hipMemGetInfo(&free_b, &total_b));
		if (at > free_b) { char b[160]; snprintf(b, sizeof b, "decoder workspace for max_batch %d needs %zu MiB, %zu MiB of HBM are free", max_batch, at >> 20, free_b >> 20); g_derr = b; return NHW_E_ARG; }
		HIPCHK(hipMalloc(&d->ws.base, at));
		HIPCHK(hipMemset(d->ws.base, 0, at));
		HIPCHK(hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking));
		HIPCHK(hipStreamCreateWithFlags(&d->chroma_stream, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&d->fork_ev, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&d->join_ev, hipEventDisableTiming));
		for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&d->ev[i]));
		HIPCHK(hipMalloc(&d->vlc_table, (256 + 16 * 64) * sizeof(uint16_t)));
		k_dec_vlc_table<<<1, 64, 0, d->own_stream>>>(d->vlc_table);
		HIPCHK(hipStreamSynchronize(d->own_stream));
		if (synth2d_attrs() != NHW_OK) { g_derr = "hipFuncSetAttribute(129 KB of LDS for the block kernels) failed"; return NHW_E_HIP; }
		return NHW_OK;
	}();
	if (rc != NHW_OK) { nhw_dec_destroy(d); return rc; }
	d->chroma_fork = 4;                                              /* bit 0: the entropy branches side by side, bit 1: the chroma sequence beside the luma one.  Both bought 2.2 ms in round 2; with the kernels of round 3 either one only stretches the kernels it overlaps (a 4096-file batch: 7.03 ms with both, 6.66 with neither), so both are off unless asked for; bit 2: the chroma sequence behind the luma's level 2, beside the smooth-edge marks only (6.51 ms): on */
	if (const char *p = getenv("NHW_CHROMA_FORK")) d->chroma_fork = atoi(p) != 0 ? 3 : 0;
	if (const char *p = getenv("NHW_DEC_FORK")) d->chroma_fork = atoi(p) & 7;
	*out = d;
	return NHW_OK;
}

extern "C" void nhw_dec_destroy(nhw_dec *d)
{
	if (!d) return;
	(void)hipSetDevice(d->device);
	if (d->ws.base) (void)hipFree(d->ws.base);
	if (d->d_blob) (void)hipFree(d->d_blob);
	if (d->d_off) (void)hipFree(d->d_off);
	if (d->d_len) (void)hipFree(d->d_len);
	if (d->d_out) (void)hipFree(d->d_out);
	if (d->d_status) (void)hipFree(d->d_status);
	if (d->d_quality) (void)hipFree(d->d_quality);
	if (d->own_stream) (void)hipStreamDestroy(d->own_stream);
	if (d->chroma_stream) (void)hipStreamDestroy(d->chroma_stream);
	if (d->vlc_table) (void)hipFree(d->vlc_table);
	if (d->fork_ev) (void)hipEventDestroy(d->fork_ev);
	if (d->join_ev) (void)hipEventDestroy(d->join_ev);
	for (int i = 0; i < 4; i++) if (d->ev[i]) (void)hipEventDestroy(d->ev[i]);
	delete d;
}

extern "C" void nhw_dec_debug_stop_after(nhw_dec *d, int stage) { if (d) d->stop_after = stage; }

/* debug: copy a workspace buffer of one image to the host (what = D_* index) */
extern "C" int nhw_dec_debug_colour(int quality, const void *d_yuv, void *d_rgb, int n)
{
	if (!d_yuv || !d_rgb || n < 8 || (n & 7) || quality < 1 || quality > 23) return NHW_E_ARG;
	k_dec_colour_probe<<<(n / 8 + 255) / 256, 256>>>((const uint8_t *)d_yuv, (uint8_t *)d_rgb, n / 8, quality);
	return hipGetLastError() == hipSuccess ? NHW_OK : NHW_E_HIP;
}
extern "C" int nhw_dec_debug_read(nhw_dec *d, int what, int img, void *dst, size_t bytes)
{
	if (!d || what < 0 || what >= D_COUNT || img < 0 || img >= d->max_batch || bytes > k_dec_bytes[what]) { g_derr = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(d->device));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(dst, d->ws.base + d->ws.off[what] + (size_t)img * k_dec_bytes[what], bytes, hipMemcpyDeviceToHost));
	return NHW_OK;
}

extern "C" int nhw_dec_batch_device(nhw_dec *d, const void *d_nhw, const uint64_t *d_off, const uint32_t *d_len, int n, void *d_bgr, int32_t *d_status,
                                    int32_t *d_quality, void *stream)
{
	if (!d || !d_nhw || !d_off || !d_len || !d_bgr || !d_status || n < 1 || n > d->max_batch) { g_derr = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(d->device));                              /* the handle's device, whatever the calling thread had current */
	hipStream_t s = stream ? (hipStream_t)stream : d->own_stream;
	DecWs ws = d->ws;
	ws.n = n; ws.blob = (const uint8_t *)d_nhw; ws.blob_off = d_off; ws.blob_len = d_len; ws.dense = d->stop_after != 0;
	int stage = 0;
	d->timed = false;
	const bool fork = (d->chroma_fork & 2) && !d->stop_after;          /* the chroma sequence beside the luma one */
	const bool fork_e = (d->chroma_fork & 1) && !d->stop_after;        /* the two entropy branches side by side */
	hipStream_t cs = fork ? d->chroma_stream : s;
	hipStream_t es = fork_e ? d->chroma_stream : s;
	const bool fork_late = (d->chroma_fork & 4) && !fork && !d->stop_after;   /* the chroma sequence beside the smooth-edge marks only: behind the luma's level 2 */
#define STAGE_END() do { if (d->stop_after && ++stage >= d->stop_after) goto done; } while (0)
#define EV(i) HIPCHK(hipEventRecord(d->ev[i], s))
	EV(0);
	/* Two entropy branches that only meet at the expansion: the side streams (LL2 DPCM, position lists; latency-bound scans) on the caller's
	 * stream, the prefix-code walk and the un-zig-zag on the second one. */
	if (fork_e) { HIPCHK(hipEventRecord(d->fork_ev, s)); HIPCHK(hipStreamWaitEvent(es, d->fork_ev, 0)); }
	k_dec_parse<<<4 * n, 64, 0, s>>>(ws);
	STAGE_END();                                                                  /* 1 */
	k_dec_vlc<<<2 * n, 64, 0, es>>>(ws, d->vlc_table);
	if (fork_e) { HIPCHK(hipEventRecord(d->join_ev, es)); HIPCHK(hipStreamWaitEvent(s, d->join_ev, 0)); }
	if (fork) { HIPCHK(hipEventRecord(d->fork_ev, s)); HIPCHK(hipStreamWaitEvent(cs, d->fork_ev, 0)); }   /* chroma goes on once both branches are in */
	k_dec_verdict<<<(n + 255) / 256, 256, 0, s>>>(ws);
	EV(1);
	STAGE_END();                                                                  /* 2 */
	/* From here the luma and the chroma sequences share nothing until the colour kernel: chroma goes to a stream of its own.  With the debug
	 * stop it stays in line. */
#define CHROMA(UPTO, STREAM) k_dec_chroma<<<2 * n < SYNTH_WGS ? 2 * n : SYNTH_WGS, 1024, 256 * 258 * sizeof(int16_t), STREAM>>>(ws, 2 * n, UPTO)
	k_dec_expand<<<(n + 3) / 4, 256, 0, s>>>(ws);
	if (fork) {
		CHROMA(4, cs);
		k_dec_sharpen<<<(2 * n + 3) / 4, 256, 0, cs>>>(ws);
		HIPCHK(hipEventRecord(d->join_ev, cs));
	}
	else if (d->stop_after == 3) CHROMA(1, s);                                    /* the chroma planes as the expansion leaves them */
	STAGE_END();                                                                  /* 3 */
	{
		/* level 2 of the luma: shrink, synthesis, residual lists on A's top-left 256 x 256 -> the level-1 LL in the same place */
		const int upto = d->stop_after == 4 ? 1 : d->stop_after == 5 ? 2 : 3;
		static const int quarters = getenv("NHW_DEC_L2Q") ? atoi(getenv("NHW_DEC_L2Q")) : 1;
		if (upto == 3 && quarters) k_dec_luma_l2q<<<4 * ((n + 7) & ~7), 256, 0, s>>>(ws, n);
		else k_dec_luma_l2<<<n < SYNTH_WGS ? n : SYNTH_WGS, 1024, 256 * 258 * sizeof(int16_t), s>>>(ws, n, upto);
	}
	STAGE_END();                                                                  /* 4 (the block as the shrink leaves it) */
	STAGE_END();                                                                  /* 5 (after the synthesis) */
	STAGE_END();                                                                  /* 6 */
	if (fork_late) {
		HIPCHK(hipEventRecord(d->fork_ev, s)); HIPCHK(hipStreamWaitEvent(d->chroma_stream, d->fork_ev, 0));
		CHROMA(4, d->chroma_stream);
		k_dec_sharpen<<<(2 * n + 3) / 4, 256, 0, d->chroma_stream>>>(ws);
		HIPCHK(hipEventRecord(d->join_ev, d->chroma_stream));
	}
	k_dec_marks<<<(n + 3) / 4, 256, 0, s>>>(ws);
	STAGE_END();                                                                  /* 7 */
	if (fork || fork_late) HIPCHK(hipStreamWaitEvent(s, d->join_ev, 0));
	else {
		CHROMA(d->stop_after == 8 ? 2 : d->stop_after == 9 ? 3 : 4, s);
		STAGE_END();                                                              /* 8 (after level 2) */
		STAGE_END();                                                              /* 9 (after the corrections) */
		STAGE_END();                                                              /* 10 */
		k_dec_sharpen<<<(2 * n + 3) / 4, 256, 0, s>>>(ws);
		STAGE_END();                                                              /* 11 */
	}
	/* level-1 synthesis both ways + corrections + smoothing + colour: one kernel, one band of FR output rows per workgroup */
	EV(2);
	{
		int dev_stop = 0;
#ifdef NHW_DEV
		if (const char *e = getenv("NHW_FINAL_STOP")) dev_stop = atoi(e);
#endif
		k_dec_final<<<(DW / FR) * n, 256, F_LDS_BYTES, s>>>(ws, (uint8_t *)d_bgr, dev_stop);
	}
	EV(3);
	d->timed = true;
done:
	k_dec_status<<<(n + 255) / 256, 256, 0, s>>>(ws, d_status, d_quality);
	HIPCHK(hipGetLastError());
	return NHW_OK;
#undef STAGE_END
#undef EV
#undef CHROMA
}

extern "C" int nhw_dec_last_timing(nhw_dec *d, nhw_dec_timing *t)
{
	if (!d || !t || !d->timed) return NHW_E_ARG;
	HIPCHK(hipEventSynchronize(d->ev[3]));
	HIPCHK(hipEventElapsedTime(&t->total_ms, d->ev[0], d->ev[3]));
	HIPCHK(hipEventElapsedTime(&t->entropy_ms, d->ev[0], d->ev[1]));
	HIPCHK(hipEventElapsedTime(&t->recon_ms, d->ev[2], d->ev[3]));
	return NHW_OK;
}

/* host convenience: H2D of the files, decode, D2H of the pixels.  nhw: the files back to back, off[n+1]. */
extern "C" int nhw_dec_batch(nhw_dec *d, const uint8_t *nhw, const uint64_t *off, int n, uint8_t *bgr, int32_t *status, int32_t *quality)
{
	if (!d || !nhw || !off || !bgr || !status || n < 1 || n > d->max_batch) { g_derr = "bad argument"; return NHW_E_ARG; }
	HIPCHK(hipSetDevice(d->device));
	const size_t total = (size_t)(off[n] - off[0]);
	if (total + 64 > d->blob_cap) {
		if (d->d_blob) (void)hipFree(d->d_blob);
		d->d_blob = nullptr; d->blob_cap = 0;                      /* nothing dangling if the allocation below fails */
		const size_t want = total + (total >> 2) + (1u << 20);
		HIPCHK(hipMalloc(&d->d_blob, want));
		d->blob_cap = want;
	}
	if (!d->d_off) {                                               /* all five or none: a half-made set would hand null pointers to the next call */
		void *b[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
		const size_t bytes[5] = { ((size_t)d->max_batch + 1) * 8, ((size_t)d->max_batch + 1) * 4, (size_t)d->max_batch * NHW_IMG_BYTES, (size_t)d->max_batch * 4, (size_t)d->max_batch * 4 };
		hipError_t err = hipSuccess;
		for (int i = 0; i < 5 && err == hipSuccess; i++) err = hipMalloc(&b[i], bytes[i]);
		if (err != hipSuccess) { for (int i = 0; i < 5; i++) if (b[i]) (void)hipFree(b[i]); HIPCHK(err); }
		d->d_off = (decltype(d->d_off))b[0]; d->d_len = (decltype(d->d_len))b[1]; d->d_out = (decltype(d->d_out))b[2];
		d->d_status = (decltype(d->d_status))b[3]; d->d_quality = (decltype(d->d_quality))b[4];
	}
	uint64_t *rel = (uint64_t *)malloc(((size_t)n + 1) * 12);
	if (!rel) return NHW_E_ARG;
	uint32_t *len = (uint32_t *)(rel + n + 1);
	for (int i = 0; i < n; i++) {
		rel[i] = off[i] - off[0];
		const uint64_t l = off[i + 1] - off[i];
		len[i] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
	}
	hipError_t e1 = hipMemcpyAsync(d->d_blob, nhw + off[0], total, hipMemcpyHostToDevice, d->own_stream);
	hipError_t e2 = hipMemcpyAsync(d->d_off, rel, (size_t)n * 8, hipMemcpyHostToDevice, d->own_stream);
	hipError_t e4 = hipMemcpyAsync(d->d_len, len, (size_t)n * 4, hipMemcpyHostToDevice, d->own_stream);
	hipError_t e3 = hipStreamSynchronize(d->own_stream);
	free(rel);
	HIPCHK(e1); HIPCHK(e2); HIPCHK(e4); HIPCHK(e3);
	const int rc = nhw_dec_batch_device(d, d->d_blob, d->d_off, d->d_len, n, d->d_out, d->d_status, d->d_quality, d->own_stream);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(bgr, d->d_out, (size_t)n * NHW_IMG_BYTES, hipMemcpyDeviceToHost, d->own_stream));
	HIPCHK(hipMemcpyAsync(status, d->d_status, (size_t)n * 4, hipMemcpyDeviceToHost, d->own_stream));
	if (quality) HIPCHK(hipMemcpyAsync(quality, d->d_quality, (size_t)n * 4, hipMemcpyDeviceToHost, d->own_stream));
	HIPCHK(hipStreamSynchronize(d->own_stream));
	return NHW_OK;
}

/* the 54-byte header nhw-dec writes in front of the pixels (nhw_decoder_cli.c:61-65, :293-312) */
extern "C" void nhw_dec_bmp_header(uint8_t h[54])
{
	static const uint8_t base[54] = { 66,77,54,0,12,0,0,0,0,0, 54,0,0,0,40,0,0,0,0,2, 0,0,0,2,0,0,1,0,24,0, 0,0,0,0,0,0,12,0,0,0 };
	memcpy(h, base, 54);
}
