/*
 * nhw_tail_dev.h -- what the tail kernels of the NHW encode path share (gfx950): the per-image view of the workspace (Ctx), the small
 * value predicates of the reference's coefficient heuristics, the format tables (escape codes, the prefix code), the position-list
 * packer.
 *
 * The passes themselves -- everything between the filterbank calls of encode_image (rcanut/nhwcodec encoder/nhw_encoder.c:103-2878,
 * encoder/image_processing.c:108-521, 2600-3353, encoder/compress_pixel.c:53-1022) -- are in nhw_tail_par.h (a 256-thread workgroup
 * per image) and nhw_tail_wave.h (a wavefront per image).  Pass ids (Y5..Y31) are those of SURVEY.md Appendix A; file:line
 * citations are into the reference encoder.  Quality 1..23.
 */
#ifndef NHW_TAIL_DEV_H
#define NHW_TAIL_DEV_H

#include "nhw_ws.h"

#define DEV __device__ static
#define NHW_OK 0
#define NHW_E_CODEBOOK (-2)
#define NHW_E_SPACE (-3)
#define S_CAP 131072   /* capacity of the sign-bit scratch lists */
#ifdef NHW_PROFILE
#define PROF_BEGIN() unsigned long long t0_ = wall_clock64()
#define PROF(c, slot) do { unsigned long long t1_ = wall_clock64(); ((unsigned long long *)((c)->prof))[slot] += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define PROF_BEGIN() do {} while (0)
#define PROF(c, slot) do {} while (0)
#endif

namespace nhw {

DEV inline int iabs(int v) { return v < 0 ? -v : v; }

struct PosList { uint8_t *list, *bits, *word; NhwPosLens *len; };

/* per-image view of the workspace; the scalar encoder state (reference encode_state, codec.h:125-181) lives in
 * NhwMeta in global memory so that every thread of the workgroup sees one copy */
struct Ctx {
	int q;
	int compat;                    /* NhwWs::compat */
	int defer_verbatim;            /* NhwWs::defer_verbatim */
	const int16_t *stale;          /* compat mode: kernel-map cells (k_front_stale) */
	int16_t *jpeg, *proc, *cjpeg, *cproc, *ll1, *l2save, *cll1, *cl2save, *keep, *first_order, *band, *hs, *tmp16;
	uint8_t *ubytes;               /* U's symbols, parked until V's quantiser merges them (quantise_chroma_par) */
	uint8_t *pu, *pv, *scan, *ll_bytes, *ll_full, *exw, *res4, *ll_comp, *ll_word, *ch_res, *res_u64, *res_v64;
	uint8_t *sel_word1, *sel_word2, *book1, *book2, *raw, *pay, *cc, *half, *s1, *s2;
	uint16_t *ll_mem, *char_res1;
	uint32_t *qsetting3, *packet;
	int *hist;
	uint64_t *nzq, *nzs;           /* luma symbol list: non-zero map [flush][strip] (quantiser), [strip][flush] = stream order (Y31) */
	uint32_t *fbase, *voff;        /* first value of every flush (33 entries: the last is the total); first value of every 64-symbol slice, stream order */
	uint8_t *vals;                 /* the non-zero symbols, flush after flush, strip after strip, stream order inside a slice */
	uint64_t *cnzq; uint32_t *cfbase; uint8_t *cvals;   /* the same of the chroma part of the stream (U and V byte-interleaved): [flush = 4 wavefront + turn][lane][2 slices]; a wavefront's values from 32768 x wavefront on */
	void *prof;
	NhwMeta *m;
	PosList res1, res3, res5, res6;
};

DEV void poslist_finish(Ctx *c, PosList *pl, uint8_t *raw, int raw_len, const uint8_t *payload, int payload_len, int word_mode);

DEV void ctx_load(Ctx *c, const NhwWs &ws, int img, int comp = 0)
{
	const bool vp = comp && ws.split_chroma;                       /* the V sequence's own planes */
	NhwMeta *m = ws.buf<NhwMeta>(B_META, img);
	c->m = m;
	c->q = ws.q;
	c->compat = ws.compat; c->defer_verbatim = ws.defer_verbatim; c->stale = ws.buf<int16_t>(B_STALE, img);
	c->jpeg = ws.buf<int16_t>(B_JPEG, img); c->proc = ws.buf<int16_t>(B_PROC, img);
	c->cjpeg = ws.buf<int16_t>(vp ? B_CJPEG_V : B_CJPEG, img); c->cproc = ws.buf<int16_t>(vp ? B_CPROC_V : B_CPROC, img);
	c->ll1 = ws.buf<int16_t>(B_LL1, img); c->l2save = ws.buf<int16_t>(B_L2SAVE, img);
	c->cll1 = ws.buf<int16_t>(vp ? B_CLL1_V : B_CLL1, img); c->cl2save = ws.buf<int16_t>(vp ? B_CL2SAVE_V : B_CL2SAVE, img);
	c->keep = ws.buf<int16_t>(B_KEEP, img); c->first_order = ws.buf<int16_t>(B_FIRST, img);
	c->band = ws.buf<int16_t>(B_BAND, img); c->hs = ws.buf<int16_t>(B_HS, img); c->tmp16 = ws.buf<int16_t>(B_TMP16, img);
	c->pu = ws.buf<uint8_t>(B_PU, img); c->pv = ws.buf<uint8_t>(B_PV, img); c->scan = ws.buf<uint8_t>(B_SCAN, img);
	c->ll_bytes = ws.buf<uint8_t>(B_LLBYTES, img); c->ll_full = ws.buf<uint8_t>(B_LLFULL, img);
	c->exw = ws.buf<uint8_t>(B_EXW, img); c->res4 = ws.buf<uint8_t>(B_RES4, img);
	c->ll_comp = ws.buf<uint8_t>(B_LLCOMP, img); c->ll_word = ws.buf<uint8_t>(B_LLWORD, img);
	c->ch_res = c->ll_comp;
	c->res_u64 = ws.buf<uint8_t>(B_RESU64, img); c->res_v64 = ws.buf<uint8_t>(B_RESV64, img);
	c->sel_word1 = ws.buf<uint8_t>(B_SEL1, img); c->sel_word2 = ws.buf<uint8_t>(B_SEL2, img);
	c->book1 = ws.buf<uint8_t>(B_BOOK1, img); c->book2 = ws.buf<uint8_t>(B_BOOK2, img);
	c->raw = ws.buf<uint8_t>(B_RAW, img); c->pay = ws.buf<uint8_t>(B_PAY, img);
	c->cc = ws.buf<uint8_t>(B_CC, img); c->half = ws.buf<uint8_t>(B_HALF, img);
	c->s1 = ws.buf<uint8_t>(B_S1, img); c->s2 = ws.buf<uint8_t>(B_S2, img);
	c->ll_mem = ws.buf<uint16_t>(B_LLMEM, img); c->char_res1 = ws.buf<uint16_t>(B_CHARRES, img);
	c->qsetting3 = ws.buf<uint32_t>(B_QSET3, img); c->packet = ws.buf<uint32_t>(B_PACKET, img);
	c->hist = ws.buf<int>(B_HIST, img);
	c->nzq = ws.buf<uint64_t>(B_NZQ, img); c->fbase = reinterpret_cast<uint32_t *>(c->nzq + 4096); c->nzs = ws.buf<uint64_t>(B_NZS, img);
	c->voff = ws.buf<uint32_t>(B_VOFF, img); c->vals = ws.buf<uint8_t>(B_VALS, img);
	c->ubytes = ws.buf<uint8_t>(B_UBYTES, img);
	c->cnzq = ws.buf<uint64_t>(B_CNZQ, img); c->cfbase = reinterpret_cast<uint32_t *>(c->cnzq + 2048); c->cvals = ws.buf<uint8_t>(B_CVALS, img);
	c->prof = ws.buf<uint8_t>(B_PROF, img);
	c->res1.list = ws.buf<uint8_t>(B_R1LIST, img); c->res1.bits = ws.buf<uint8_t>(B_R1BITS, img); c->res1.word = ws.buf<uint8_t>(B_R1WORD, img); c->res1.len = &m->r1;
	c->res3.list = ws.buf<uint8_t>(B_R3LIST, img); c->res3.bits = ws.buf<uint8_t>(B_R3BITS, img); c->res3.word = ws.buf<uint8_t>(B_R3WORD, img); c->res3.len = &m->r3;
	c->res5.list = ws.buf<uint8_t>(B_R5LIST, img); c->res5.bits = ws.buf<uint8_t>(B_R5BITS, img); c->res5.word = ws.buf<uint8_t>(B_R5WORD, img); c->res5.len = &m->r5;
	c->res6.list = ws.buf<uint8_t>(B_R6LIST, img); c->res6.bits = ws.buf<uint8_t>(B_R6BITS, img); c->res6.word = ws.buf<uint8_t>(B_R6WORD, img); c->res6.len = &m->r6;
}

__device__ static const uint8_t k_big_pos[19] = { 10, 12, 14, 18, 20, 22, 26, 28, 30, 34, 36, 38, 42, 44, 46, 50, 52, 54, 58 };
__device__ static const uint8_t k_big_neg[19] = { 60, 62, 66, 68, 70, 74, 76, 78, 82, 84, 86, 90, 92, 94, 98, 100, 102, 106, 108 };

DEV int odd(int v) { return (v & 1) == 1; }
DEV int in_4_7(int v) { return v > 3 && v <= 7; }
DEV int in_m7_m4(int v) { return v < -3 && v >= -7; }
DEV int is_567(int v) { return v == 5 || v == 6 || v == 7; }
DEV int is_m567(int v) { return v == -5 || v == -6 || v == -7; }
DEV int16_t clear_bit0(int v) { return (int16_t)(v & 0xFFFE); }

/* shared tail of the dequantiser: dead zone, bias by 128, floor the magnitude to a multiple of 8,
 * then the decoder's reconstruction offsets (image_processing.c:3003-3015) */
DEV int dequant_value(int a)
{
	if (a < DEADZONE && a > -DEADZONE) return 0;
	a += 128;
	if (a < 0) a = -((-a) & 0xFFF8); else a &= 0xFFF8;
	return a > 128 ? a - 125 : a - 131;
}

/* Byte tests on whole words.  nzb: 0x80 in every byte of y that is not zero (exact per byte: no carry crosses a byte).  nz8x128: the
 * flags of two words gathered into 8 bits (bit k: byte k of the first word, bit 4 + k: byte k of the second) -- times 128, because the
 * gather is two v_dot4_u32_u8 with the bit weights as the second operand (a flag byte is 0 or 128); shifts and ORs did that in a dozen
 * instructions a word, and the byte-mask builders were a third of the packetiser's and of Y31's vector work. */
DEV uint32_t nzb(uint32_t y) { return (((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u; }
DEV uint32_t nz8x128(uint32_t fa, uint32_t fb) { return __builtin_amdgcn_udot4(fb, 0x80402010u, __builtin_amdgcn_udot4(fa, 0x08040201u, 0u, false), false); }
/* bit 4 i + k: byte k of w[i] is NOT `pat`'s byte, i < 8 */
DEV uint32_t ne_mask32(const uint32_t *w, uint32_t pat)
{
	const uint32_t r0 = nz8x128(nzb(w[0] ^ pat), nzb(w[1] ^ pat)), r1 = nz8x128(nzb(w[2] ^ pat), nzb(w[3] ^ pat));
	const uint32_t r2 = nz8x128(nzb(w[4] ^ pat), nzb(w[5] ^ pat)), r3 = nz8x128(nzb(w[6] ^ pat), nzb(w[7] ^ pat));
	return (r0 >> 7) | (r1 << 1) | (r2 << 9) | (r3 << 17);
}

/* the codes of values beyond +-127 (the tables above: steps of 2 with a gap after every third entry).  Worked out, not looked up: the tables
 * live in global memory and the look-up sat on the quantisers' row chains as a dependent load. */
DEV int big_code(int a, const uint8_t *tab)
{
	int k = ((a & 0xFFF8) - 128) >> 3;
	k = k > 18 ? 18 : k;
	return tab == k_big_pos ? 10 + 2 * k + 2 * ((k * 11) >> 5) : 60 + 2 * k + 2 * (((k + 1) * 11) >> 5);   /* (k * 11) >> 5 = k / 3 for k < 20 */
}

DEV int mult8_or_7(int m) { return !(m & 7) || (m & 7) == 7; } /* on a magnitude */

DEV int big_step(int d) /* correction for a large closed-loop error (:225-232) */
{
	if (d > 11) return -7; if (d > 7) return -4; if (d > 5) return -2; if (d > 4) return -1;
	if (d < -11) return 7; if (d < -7) return 4; if (d < -5) return 2; if (d < -4) return 1;
	return 0;
}


/* ------------------------------------------------------------------------------------------
 * position-list side streams
 *   raw: column indices per row with a 254 marker closing every row; payload: one symbol per entry
 *   word_mode 1: one bit per payload symbol, res1 sizing (Y+1 bytes)
 *   word_mode 2: two bits per payload symbol
 * ------------------------------------------------------------------------------------------ */
DEV void poslist_finish(Ctx *c, PosList *pl, uint8_t *raw, int raw_len, const uint8_t *payload,
                         int payload_len, int word_mode)
{
	uint8_t *cc = c->cc;
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
	half = c->half;
	for (i = 0; i < n; i++) half[i] = cc[i] >> 1;
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
	pl->len->list_len = packed;

	/* plane of the dropped low bits, markers excluded (nhw_encoder.c:1594-1615) */
	for (i = 0, nb = 0; i < n; i++) if (cc[i] != H - 2) half[nb++] = cc[i];
	for (i = nb; i < nb + 8; i++) half[i] = 0;
	groups = (nb >> 3) + 1;
	for (i = 0; i < groups; i++) {
		int b, v = 0;
		for (b = 0; b < 8; b++) v = (v << 1) | (half[8 * i + b] & 1);
		pl->bits[i] = (uint8_t)v;
	}
	pl->len->bits_len = groups;

	/* payload symbols (nhw_encoder.c:1620-1631, 1751-1763); symbols behind payload_len read as 0 */
	groups = (payload_len >> 3) + 1;
	pl->len->word_len = 0;
	for (i = 0; i < groups; i++) {
		int b, sym[8];
		for (b = 0; b < 8; b++) sym[b] = (8 * i + b < payload_len) ? payload[8 * i + b] : 0;
		if (word_mode == 2) {
			pl->word[pl->len->word_len++] = (uint8_t)(((sym[0] & 3) << 6) | ((sym[1] & 3) << 4) | ((sym[2] & 3) << 2) | (sym[3] & 3));
			pl->word[pl->len->word_len++] = (uint8_t)(((sym[4] & 3) << 6) | ((sym[5] & 3) << 4) | ((sym[6] & 3) << 2) | (sym[7] & 3));
		} else {
			int v = 0;
			for (b = 0; b < 8; b++) v = (v << 1) | (sym[b] & 1);
			pl->word[pl->len->word_len++] = (uint8_t)v;
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * RLE + VLC packetiser (wavlts2packet): the code table; the packetiser itself is workgroup-parallel (nhw_tail_par.h)
 * ------------------------------------------------------------------------------------------ */

/* rank -> (length << 24) | code word of the NHW prefix code (format constant; reference encoder/tree.h:58-140) */
__device__ static const uint32_t k_vlc[290] = {
	0x02000000, 0x03000002, 0x03000004, 0x0400000a, 0x0400000b, 0x04000006, 0x04000007, 0x05000018,
	0x05000019, 0x0500001a, 0x06000036, 0x06000037, 0x07000070, 0x07000071, 0x080000e8, 0x080000e9,
	0x080000ea, 0x080000eb, 0x080000ec, 0x080000ed, 0x080000ee, 0x080000ef, 0x080000f0, 0x080000f1,
	0x080000f2, 0x080000f3, 0x090001c8, 0x090001c9, 0x090001ca, 0x090001cb, 0x090001cc, 0x090001cd,
	0x090001ce, 0x090001cf, 0x090001e8, 0x090001e9, 0x090001ea, 0x090001eb, 0x090001ec, 0x090001ed,
	0x090001ee, 0x090001ef, 0x0a0003e8, 0x0a0003e9, 0x0a0003ea, 0x0a0003eb, 0x0a0003ec, 0x0a0003ed,
	0x0a0003ee, 0x0a0003ef, 0x0a0003e4, 0x0a0003e5, 0x0a0003e6, 0x0a0003e7, 0x0b0007c0, 0x0b0007c1,
	0x0b0007e0, 0x0b0007e1, 0x0b0007f0, 0x0b0007f1, 0x0b0007f2, 0x0b0007f3, 0x0b0007f4, 0x0b0007f5,
	0x0b0007f6, 0x0b0007f7, 0x0b0007f8, 0x0b0007f9, 0x0b0007fa, 0x0b0007fb, 0x0b0007fc, 0x0b0007fd,
	0x0b0007fe, 0x0b0007ff, 0x0b0007e8, 0x0b0007e9, 0x0b0007ea, 0x0b0007eb, 0x0b0007ec, 0x0b0007ed,
	0x0b0007ee, 0x0b0007ef, 0x0c000f88, 0x0c000f89, 0x0c000f8a, 0x0c000f8b, 0x0c000f8c, 0x0c000f8d,
	0x0c000f8e, 0x0c000f8f, 0x0c000fc8, 0x0c000fc9, 0x0c000fca, 0x0c000fcb, 0x0c000fcc, 0x0c000fcd,
	0x0c000fce, 0x0c000fcf, 0x0d001f08, 0x0d001f09, 0x0d001f0a, 0x0d001f0b, 0x0e003f10, 0x0e003f11,
	0x0e003f12, 0x0e003f13, 0x0e003f14, 0x0e003f15, 0x0e003f16, 0x0e003f17, 0x1101f0c0, 0x1101f0c1,
	0x1101f0c2, 0x1101f0c3, 0x1101f0c4, 0x1101f0c5, 0x1101f0c6, 0x1101f0c7, 0x1101f0c8, 0x1101f0c9,
	0x1101f0ca, 0x1101f0cb, 0x1101f0cc, 0x1101f0cd, 0x1101f0ce, 0x1101f0cf, 0x1101f0d0, 0x1101f0d1,
	0x1101f0d2, 0x1101f0d3, 0x1101f0d4, 0x1101f0d5, 0x1101f0d6, 0x1101f0d7, 0x1101f0d8, 0x1101f0d9,
	0x1101f0da, 0x1101f0db, 0x1101f0dc, 0x1101f0dd, 0x1101f0de, 0x1101f0df, 0x1101f0e0, 0x1101f0e1,
	0x1101f0e2, 0x1101f0e3, 0x1101f0e4, 0x1101f0e5, 0x1101f0e6, 0x1101f0e7, 0x1101f0e8, 0x1101f0e9,
	0x1101f0ea, 0x1101f0eb, 0x1101f0ec, 0x1101f0ed, 0x1101f0ee, 0x1101f0ef, 0x1101f0f0, 0x1101f0f1,
	0x1101f0f2, 0x1101f0f3, 0x1101f0f4, 0x1101f0f5, 0x1101f0f6, 0x1101f0f7, 0x1101f0f8, 0x1101f0f9,
	0x1101f0fa, 0x1101f0fb, 0x1101f0fc, 0x1101f0fd, 0x1101f0fe, 0x1101f0ff, 0x1101f8c0, 0x1101f8c1,
	0x1101f8c2, 0x1101f8c3, 0x1101f8c4, 0x1101f8c5, 0x1101f8c6, 0x1101f8c7, 0x1101f8c8, 0x1101f8c9,
	0x1101f8ca, 0x1101f8cb, 0x1101f8cc, 0x1101f8cd, 0x1101f8ce, 0x1101f8cf, 0x1101f8d0, 0x1101f8d1,
	0x1101f8d2, 0x1101f8d3, 0x1101f8d4, 0x1101f8d5, 0x1101f8d6, 0x1101f8d7, 0x1101f8d8, 0x1101f8d9,
	0x1101f8da, 0x1101f8db, 0x1101f8dc, 0x1101f8dd, 0x1101f8de, 0x1101f8df, 0x1101f8e0, 0x1101f8e1,
	0x1101f8e2, 0x1101f8e3, 0x1101f8e4, 0x1101f8e5, 0x1101f8e6, 0x1101f8e7, 0x1101f8e8, 0x1101f8e9,
	0x1101f8ea, 0x1101f8eb, 0x1101f8ec, 0x1101f8ed, 0x1203f1dc, 0x1203f1dd, 0x1203f1de, 0x1203f1df,
	0x1203f1e0, 0x1203f1e1, 0x1203f1e2, 0x1203f1e3, 0x1203f1e4, 0x1203f1e5, 0x1203f1e6, 0x1203f1e7,
	0x1307e3d0, 0x1307e3d1, 0x1307e3d2, 0x1307e3d3, 0x1307e3d4, 0x1307e3d5, 0x1307e3d6, 0x1307e3d7,
	0x1307e3d8, 0x1307e3d9, 0x1307e3da, 0x1307e3db, 0x1307e3dc, 0x1307e3dd, 0x1307e3de, 0x1307e3df,
	0x1307e3e0, 0x1307e3e1, 0x1307e3e2, 0x1307e3e3, 0x1307e3e4, 0x1307e3e5, 0x1307e3e6, 0x1307e3e7,
	0x1307e3e8, 0x1307e3e9, 0x1307e3ea, 0x1307e3eb, 0x1307e3ec, 0x1307e3ed, 0x1307e3ee, 0x1307e3ef,
	0x1307e3f0, 0x1307e3f1, 0x1307e3f2, 0x1307e3f3, 0x1307e3f4, 0x1307e3f5, 0x140fc7ec, 0x140fc7ed,
	0x140fc7ee, 0x140fc7ef, 0x140fc7f0, 0x140fc7f1, 0x140fc7f2, 0x140fc7f3, 0x140fc7f4, 0x140fc7f5,
	0x140fc7f6, 0x140fc7f7, 0x140fc7f8, 0x140fc7f9, 0x140fc7fa, 0x140fc7fb, 0x140fc7fc, 0x140fc7fd,
	0x140fc7fe, 0x140fc7ff,
};

/* ---------------------------------------------------------------- q>=22 extras */

/* decoder's view of the quantised LH1 band (rows<256, cols 256..511): image_processing.c:523-556.
 * code -> signed index of the |coef|>127 escape codes (tree.h:142-147): 10,12,14,18.. -> 1..19, 60.. -> -1..-19 */
DEV int big_index(int code)
{
	__device__ static const uint8_t pos[19] = { 10, 12, 14, 18, 20, 22, 26, 28, 30, 34, 36, 38, 42, 44, 46, 50, 52, 54, 58 };
	__device__ static const uint8_t neg[19] = { 60, 62, 66, 68, 70, 74, 76, 78, 82, 84, 86, 90, 92, 94, 98, 100, 102, 106, 108 };
	int k;
	for (k = 0; k < 19; k++) { if (code == pos[k]) return k + 1; if (code == neg[k]) return -(k + 1); }
	return 0;
}

} // namespace nhw
#endif
