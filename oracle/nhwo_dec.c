/*
 * nhwo_dec.c -- oracle: CPU restatement of the NHW *decoder* (BASELINE config 5).
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and nothing else (see nhwo.h).
 *
 * Follows   decoder/nhw_decoder.c:54-1476 (decode_image), :1478-2032 (parse_file),
 *           decoder/compress_pixel.c:49-444 / :446-640 (retrieve_pixel_Y_comp / _UV_comp),
 *           decoder/wavelet_filterbank.c:52-357, decoder/filters.c:143-194,
 *           decoder/nhw_decoder_cli.c:108-291 (write_image_bmp), :61-65 + :293-312 (BMP header).
 * Pinned against oracle/_ref/libnhwref_dec.so (the unmodified reference decoder + zero-guard allocator)
 * by tests/test_oracle_decode.py on reference-encoded files of every quality 1..23.
 *
 * Out-of-bounds / never-written reads follow the canonical model (they read 0): every work buffer here
 * is zero-filled and sits inside a zero margin.
 */
#include <stdlib.h>
#include <string.h>
#include "nhwo.h"

#define W 512
#define H 256
#define Q 65536
#define MARGIN 4096

static inline int iabs(int v) { return v < 0 ? -v : v; }

/* checkpoint probe (tools/dev/gpu_dec_debug.py): copy one intermediate buffer out of the next decode */
static struct { int id; void *dst; size_t cap, got; } g_probe;
void nhwo_dec_probe(int id, void *dst, size_t cap) { g_probe.id = id; g_probe.dst = dst; g_probe.cap = cap; g_probe.got = 0; }
size_t nhwo_dec_probe_len(void) { return g_probe.got; }
static void probe(int id, const void *p, size_t n)
{
	if (id != g_probe.id || !g_probe.dst) return;
	if (n > g_probe.cap) n = g_probe.cap;
	memcpy(g_probe.dst, p, n); g_probe.got = n;
}

/* ------------------------------------------------------------------------------------------ container */
typedef struct {
	const uint8_t *p; size_t n, at; int bad;
} src;
static unsigned get8(src *s) { if (s->at + 1 > s->n) { s->bad = 1; return 0; } return s->p[s->at++]; }
static unsigned get16(src *s) { unsigned a = get8(s); return a | (get8(s) << 8); }
static uint32_t get32(src *s) { uint32_t a = get16(s); return a | ((uint32_t)get16(s) << 16); }
static const uint8_t *getn(src *s, size_t n) { const uint8_t *r = s->p + s->at; if (s->at + n > s->n) { s->bad = 1; return s->p; } s->at += n; return r; }

typedef struct {
	int res_high, q;
	int book1_len, book2_len, data1, data2, tree_end, exw_len;
	int res1_len, res1_bits, res3_len, res3_bits, res4_len, res5_len, res5_bits, res6_len, res6_bits, char_res1_len, qs3_len;
	int select1, select2, ll_word_len, ch_res_len;
	const uint8_t *book1, *book2, *exw, *res1, *res1_bit, *res1_word, *res4, *res3, *res3_bit, *res3_word;
	const uint8_t *res5, *res5_bit, *res5_word, *res6, *res6_bit, *res6_word, *char_res1, *qs3;
	const uint8_t *sel1, *sel2, *res_u64, *res_v64, *ll_word, *ch_res, *packet1, *packet2;
} nhw_file;

/* parse_file, nhw_decoder.c:1497-1659: header fields, then the sections in file order */
static int parse_container(const uint8_t *d, size_t len, nhw_file *f)
{
	src s = { d, len, 0, 0 };
	int q;
	memset(f, 0, sizeof *f);
	f->res_high = (int)get8(&s);
	f->q = q = (int)get8(&s);
	if (s.bad || f->res_high > 6 || q < 1 || q > 23) return -1;             /* :1500 "Not an .nhw file" */
	f->book1_len = (int)get16(&s); f->book2_len = (int)get16(&s);
	f->data1 = (int)get32(&s); f->data2 = (int)get32(&s);
	f->tree_end = (int)get16(&s); f->exw_len = (int)get16(&s);
	if (q > 12) f->res1_len = (int)get16(&s);
	if (q >= 19) { f->res3_len = (int)get16(&s); f->res3_bits = (int)get16(&s); }
	if (q > 17) f->res4_len = (int)get16(&s);
	if (q > 12) f->res1_bits = (int)get16(&s);
	if (q >= 21) { f->res5_len = (int)get16(&s); f->res5_bits = (int)get16(&s); }
	if (q > 21) { f->res6_len = (int)get32(&s); f->res6_bits = (int)get16(&s); f->char_res1_len = (int)get16(&s); }
	if (q > 22) f->qs3_len = (int)get16(&s);
	f->select1 = (int)get16(&s); f->select2 = (int)get16(&s);
	if (q > 15) f->ll_word_len = (int)get16(&s);
	f->ch_res_len = (int)get16(&s);
	if (s.bad || f->data2 < f->data1 || f->data1 < 0) return -1;

	f->book1 = getn(&s, (size_t)f->book1_len); f->book2 = getn(&s, (size_t)f->book2_len);
	f->exw = getn(&s, (size_t)f->exw_len);
	if (q > 12) { f->res1 = getn(&s, (size_t)f->res1_len); f->res1_bit = getn(&s, (size_t)f->res1_bits); f->res1_word = getn(&s, (size_t)f->res1_bits); }
	if (q > 17) f->res4 = getn(&s, (size_t)f->res4_len);
	if (q >= 19) { f->res3 = getn(&s, (size_t)f->res3_len); f->res3_bit = getn(&s, (size_t)f->res3_bits); f->res3_word = getn(&s, (size_t)f->res3_bits * 2); }
	if (q >= 21) { f->res5 = getn(&s, (size_t)f->res5_len); f->res5_bit = getn(&s, (size_t)f->res5_bits); f->res5_word = getn(&s, (size_t)f->res5_bits); }
	if (q > 21) {
		f->res6 = getn(&s, (size_t)f->res6_len); f->res6_bit = getn(&s, (size_t)f->res6_bits); f->res6_word = getn(&s, (size_t)f->res6_bits);
		f->char_res1 = getn(&s, (size_t)f->char_res1_len * 2);
	}
	if (q > 22) f->qs3 = getn(&s, (size_t)f->qs3_len * 4);
	f->sel1 = getn(&s, (size_t)f->select1); f->sel2 = getn(&s, (size_t)f->select2);
	if (q > 15) { f->res_u64 = getn(&s, 2 * H); f->res_v64 = getn(&s, 2 * H); f->ll_word = getn(&s, (size_t)f->ll_word_len); }
	f->ch_res = getn(&s, (size_t)f->ch_res_len);
	f->packet1 = getn(&s, (size_t)f->data1 * 4);
	f->packet2 = getn(&s, (size_t)(f->data2 - f->data1) * 4);
	return s.bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------ work memory */
typedef struct { uint8_t *base; size_t cap, used; } pool;
static void *grab(pool *p, size_t bytes)
{
	const size_t need = (bytes + 63) & ~(size_t)63;
	uint8_t *r;
	if (p->used + need + 2 * MARGIN > p->cap) return NULL;
	r = p->base + p->used + MARGIN;
	p->used += need + MARGIN;
	return r;
}

/* ------------------------------------------------------------------------------------------ d1: LL2 samples
 * nhw_decoder.c:1661-2026.  ll[] is the reference's res_comp (unsigned char arithmetic, wraps mod 256):
 * 16384 luma LL2 samples, then 4096 U and 4096 V. `code` = res_ch (a private copy: the walk edits it). */
static int ll_expand(const nhw_file *f, uint8_t *code, uint8_t *ll)
{
	const int mode = f->res_high & 3, q = f->q;
	int i = 1, j = 1, a = 0, e;
	static const int8_t dc_pair[8][2] = { {0,4},{0,-4},{4,0},{-4,0},{4,4},{4,-4},{-4,4},{-4,-4} };   /* :1482 */
#define PUSH(v) do { ll[j] = (uint8_t)(v); j++; } while (0)
#define PREV ((int)ll[j - 1])
	ll[0] = code[0];
	while (j < Q / 4) {
		const int b = code[i];
		if (b >= 128) {                                        /* verbatim sample, preceded by its fine byte when q>15 */
			if (q > 15) PUSH(f->ll_word[a++]);
			PUSH((b - 128) << 1);
		}
		else if (mode == 0 || mode == 3) {                     /* :1665-1789 (3 never occurs; the reference would land here) */
			if (b < 16) {
				const int run = ((b >> 3) & 1) + 2, v = PREV;
				for (e = 0; e < run; e++) PUSH(v);
				switch (b & 7) {
				case 1: PUSH(PREV + 2); break;
				case 2: PUSH(PREV + 2); PUSH(PREV - 2); break;
				case 3: PUSH(PREV + 2); PUSH(PREV); break;
				case 4: PUSH(PREV - 2); PUSH(PREV + 2); break;
				case 5: PUSH(PREV - 2); PUSH(PREV); break;
				case 6: PUSH(PREV - 2); break;
				case 7: PUSH(PREV + 4); break;
				default: break;
				}
			}
			else if (b < 32) { PUSH(PREV + (b >= 24 ? 4 : 2)); PUSH(((b & 7) << 1) - 8 + PREV); }
			else if (b < 64) { const int c = b - 32; PUSH(((c >> 3) << 1) - 6 + PREV); PUSH(((c & 7) << 1) - 8 + PREV); }
			else goto triple;
		}
		else if (mode == 1) {                                  /* :1791-1842 */
			if (b < 32) {
				const int run = ((b >> 2) & 7) + 2, v = PREV;
				for (e = 0; e < run; e++) PUSH(v);
				switch (b & 3) { case 1: PUSH(PREV + 2); break; case 2: PUSH(PREV - 2); break; case 3: PUSH(PREV); break; default: break; }
			}
			else if (b < 64) { const int c = b - 32; PUSH(((c >> 3) << 1) - 4 + PREV); PUSH(((c & 7) << 1) - 8 + PREV); }
			else goto triple;
		}
		else {                                                 /* mode 2, :1844-1876 */
			if (b < 64) { const int run = (b & 63) + 2, v = PREV; for (e = 0; e < run; e++) PUSH(v); }
			else {
				int c, d;
triple:				c = code[i] - 64; i++; d = code[i];            /* three differences in two bytes: 5 + 4 + 5 bits */
				PUSH((((c >> 1) & 31) << 1) - 32 + PREV);
				PUSH(((((c & 1) << 3) | (d >> 5)) << 1) - 16 + PREV);
				PUSH(((d & 31) << 1) - 32 + PREV);
			}
		}
		i++;
	}
	ll[Q / 4] = code[i++];                                         /* :1878: first chroma sample, verbatim */
	j = Q / 4 + 1;
	while (j < Q / 4 + Q / 8) {                                    /* :1882-1979 */
		const int b = code[i];
		if (b >= 192) {
			const int c = b - 192;
			PUSH(dc_pair[c >> 2][0] + PREV); PUSH(dc_pair[c >> 2][1] + PREV);
			switch (c & 3) { case 0: PUSH(PREV); break; case 1: PUSH(PREV + 4); break; case 2: PUSH(PREV - 4); break; default: PUSH(PREV + 8); break; }
		}
		else if (b >= 128) PUSH((b - 128) << 2);
		else if (b >= 64) {
			int run = (b >> 3) & 7;
			const int v = PREV;
			if (run == 7) { run = (b & 7) + 7; for (e = 0; e < run + 2; e++) PUSH(v); }
			else {
				for (e = 0; e < run + 2; e++) PUSH(v);
				switch (b & 7) {
				case 1: PUSH(PREV + 4); break;
				case 2: PUSH(PREV + 4); PUSH(PREV - 4); break;
				case 3: PUSH(PREV + 4); PUSH(PREV - 4); PUSH(PREV); break;
				case 4: PUSH(PREV - 4); PUSH(PREV + 4); PUSH(PREV); break;
				case 5: PUSH(PREV - 4); PUSH(PREV + 4); break;
				case 6: PUSH(PREV - 4); break;
				case 7: PUSH(PREV + 8); break;
				default: break;
				}
			}
		}
		else { PUSH(((b >> 3) << 2) - 16 + PREV); PUSH(((b & 7) << 2) - 16 + PREV); }
		i++;
	}
#undef PUSH
#undef PREV
	if (q > 15) {                                                  /* bit-1 planes of the chroma samples, :1983-2026 */
		for (i = 0; i < 2 * H; i++) for (e = 0; e < 8; e++) {
			ll[Q / 4 + 8 * i + e] = (uint8_t)(ll[Q / 4 + 8 * i + e] + (((f->res_u64[i] >> (7 - e)) & 1) << 1));
			ll[Q / 4 + Q / 16 + 8 * i + e] = (uint8_t)(ll[Q / 4 + Q / 16 + 8 * i + e] + (((f->res_v64[i] >> (7 - e)) & 1) << 1));
		}
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------ d2: VLC
 * The fixed prefix code (same 290 words as the encoder, encoder/tree.h:58-140) as {first, length, count}
 * runs; decoder/tables.h:59,125 are this code laid out as two lookup tables. */
static const struct { uint32_t first; uint8_t len; uint16_t count; } k_runs[] = {
	{0x0000,2,1},{0x0002,3,1},{0x0004,3,1},{0x000a,4,2},{0x0006,4,2},{0x0018,5,3},{0x0036,6,2},{0x0070,7,2},
	{0x00e8,8,12},{0x01c8,9,8},{0x01e8,9,8},{0x03e8,10,8},{0x03e4,10,4},{0x07c0,11,2},{0x07e0,11,2},
	{0x07f0,11,16},{0x07e8,11,8},{0x0f88,12,8},{0x0fc8,12,8},{0x1f08,13,4},{0x3f10,14,8},
	{0x1f0c0,17,64},{0x1f8c0,17,46},{0x3f1dc,18,12},{0x7e3d0,19,38},{0xfc7ec,20,20}
};

typedef struct { const uint8_t *p; size_t words; size_t bit; } bitsrc;          /* MSB-first within little-endian 32-bit words */
static inline unsigned peek_bit(const bitsrc *b, size_t at)
{
	const size_t w = at >> 5;
	uint32_t v;
	if (w >= b->words) return 0;
	v = (uint32_t)b->p[4 * w] | ((uint32_t)b->p[4 * w + 1] << 8) | ((uint32_t)b->p[4 * w + 2] << 16) | ((uint32_t)b->p[4 * w + 3] << 24);
	return (v >> (31 - (at & 31))) & 1;
}
static inline uint32_t peek_bits(const bitsrc *b, int n)
{
	uint32_t v = 0; int k;
	for (k = 0; k < n; k++) v = (v << 1) | peek_bit(b, b->bit + (size_t)k);
	return v;
}
/* next code word -> rank (0..289), or -1 */
static int next_rank(bitsrc *b)
{
	const uint32_t look = peek_bits(b, 20);
	size_t r; int rank = 0;
	for (r = 0; r < sizeof k_runs / sizeof k_runs[0]; r++) {
		const uint32_t v = look >> (20 - k_runs[r].len);
		if (v >= k_runs[r].first && v < k_runs[r].first + k_runs[r].count) { b->bit += k_runs[r].len; return rank + (int)(v - k_runs[r].first); }
		rank += k_runs[r].count;
	}
	return -1;
}

/* the |a|>127 escape symbols (decoder/tables.h:51): even codes 10..108 with residue 2,4,6 -> +-(123 + 8n) */
static int extra_level(int word)
{
	const int off = word & 7;
	int n;
	if (word < 10 || word > 108 || (off != 2 && off != 4 && off != 6)) return 0;
	n = ((word >> 3) - 1) * 3 + (off >> 1);
	return n <= 19 ? n : -(n - 19);
}
static int plain_level(int word)                                                  /* L_INVQ, compress_pixel.c:384 */
{
	const int x = word < 110 ? extra_level(word) : 0;
	if (x > 0) return 123 + (x << 3);
	if (x < 0) return (x << 3) - 123;
	return word > 128 ? word - 125 : word - 131;
}

/* books: compress_pixel.c:86-117 (luma) / :456-478 (chroma).  entry = (run length << 8) | symbol */
static int build_book(const uint8_t *raw, int raw_len, int chroma, int tree_end, uint16_t *book)
{
	uint8_t flat[1024], inter[1024];
	const int rep = chroma ? 128 : 3;
	int i, j, e = 0, n = 0;
	memset(flat, 0, sizeof flat); memset(inter, 0, sizeof inter);
	for (i = 0; i < raw_len; i++) {
		if (raw[i] == rep) { const int cnt = i + 1 < raw_len ? raw[i + 1] : 0; for (j = 0; j < cnt && e < 1000; j++) flat[e++] = (uint8_t)rep; i++; }
		else if (e < 1000) flat[e++] = raw[i];
	}
	if (chroma) e = tree_end;                                     /* :472 */
	if (e > 708) e = 708;
	for (i = 0, j = 0; i < e; i += 2) inter[i] = flat[j++];       /* undo the even/odd split */
	for (i = 1; i < e; i += 2) inter[i] = flat[j++];
	for (i = 0; i < e; i++) {
		if (!chroma) {
			if (inter[i] == 3) { book[n++] = (uint16_t)((inter[i + 1] << 8) | 128); i++; }
			else book[n++] = (uint16_t)(256 | inter[i]);
		} else {
			if (!(inter[i] & 1)) { book[n++] = (uint16_t)((inter[i + 1] << 8) | inter[i]); i++; }
			else book[n++] = (uint16_t)(256 | (inter[i] & 0xfe));
		}
	}
	return n;
}

static inline int bit_of(const uint8_t *bytes, int nbytes, int k) { return (k >> 3) < nbytes ? (bytes[k >> 3] >> (7 - (k & 7))) & 1 : 0; }

/* retrieve_pixel_Y_comp: out[] = 262144 shorts in scan order (zero-filled by the caller, zero margin before it) */
static int vlc_luma(const nhw_file *f, int16_t *out)
{
	uint16_t book[720];
	bitsrc b = { f->packet1, (size_t)f->data1, 0 };
	const int zoned = f->res_high < 4;                            /* zone_number==1, :84 */
	const int limit = 4 * Q - 1;
	int e = 0, mem = 0, mem2 = 0, ac1 = 0, run_over = -257, t = 0, t2 = 0, nbook;
	memset(book, 0, sizeof book);
	nbook = build_book(f->book1, f->book1_len, 0, 0, book);
	(void)nbook;
	int nsym = 0, nrun = 0, nput = 0;
	while (e < limit) {
		int rank, word, rle;
		nsym++;
		if (b.bit >= ((size_t)f->data1 + 2) * 32) return -1;
		if (zoned && peek_bits(&b, 9) == 1) { b.bit += 9; rank = 110 + (int)peek_bits(&b, 6); b.bit += 6; }     /* :127-142 */
		else {
			rank = next_rank(&b);
			if (rank < 0) return -1;
			if (zoned && rank >= 110) rank += 64;                 /* :277 */
		}
		if (rank >= 720) return -1;
		word = book[rank] & 255; rle = book[rank] >> 8;
		if (word == 128) {                                        /* zero run; +-8 values the encoder folded into runs come back here (:282-318) */
			int put = 0, neg = 0;
			mem++;
			if (mem2 == 1) {
				if ((e >= 5 && !out[e - 2] && !out[e - 3] && !out[e - 4] && !out[e - 5]) || (rle >= 4 && !out[e - 2])) { put = 1; neg = !bit_of(f->sel2, f->select2, t2++); }
				mem2 = 0;
			}
			else {
				const int room = rle >= 4 && e > 0 && !out[e - 1] && !ac1 && (e + rle - 257) >= run_over;
				if (mem == 2 && !ac1) {
					if ((e >= 4 && !out[e - 1] && !out[e - 2] && !out[e - 3] && !out[e - 4] && (e + rle - 257) >= run_over) || room) { put = 1; neg = bit_of(f->sel1, f->select1, t++); mem = 1; }
				}
				else if (room) { put = 1; neg = bit_of(f->sel1, f->select1, t++); mem = 1; }
			}
			nrun++; nput += put;
			if (put) out[e++] = (int16_t)(neg ? -11 : 11);
			if (rle == 254) { ac1 = 1; mem = 0; run_over = e; } else ac1 = 0;
			e += rle;
		}
		else {
			mem = 0; mem2 = 0; ac1 = 0;
			switch (word) {                                       /* :324-386 */
			case 136: out[e++] = 11; mem2 = 1; break;
			case 120: out[e++] = -11; mem2 = 1; break;
			case 132: out[e] = 11; e += 4; out[e++] = 11; break;
			case 133: out[e] = 11; e += 4; out[e++] = -11; break;
			case 134: out[e] = -11; e += 4; out[e++] = 11; break;
			case 135: out[e] = -11; e += 4; out[e++] = -11; break;
			case 127: out[e++] = 1008; break;
			case 129: out[e++] = 1009; break;
			case 125: out[e++] = 1006; break;
			case 126: out[e++] = 1007; break;
			case 121: out[e++] = 1010; break;
			case 122: out[e++] = 1011; break;
			case 124: out[e++] = 11; break;
			case 123: out[e++] = -11; break;
			default: out[e++] = (int16_t)plain_level(word); break;
			}
		}
	}
	{ int stats[4] = { nsym, nrun, nput, (int)b.bit }; probe(90, stats, sizeof stats); }
	return 0;
}

/* retrieve_pixel_UV_comp: out[] = 131072 shorts (U on even, V on odd positions) */
static int vlc_chroma(const nhw_file *f, int16_t *out)
{
	uint16_t book[720];
	bitsrc b = { f->packet2, (size_t)(f->data2 - f->data1), 0 };
	const int limit = 2 * Q - 2;                                  /* p1-1 with p1 = 2*IM_SIZE-1, nhw_decoder.c:897 */
	int e = 0;
	memset(book, 0, sizeof book);
	build_book(f->book2, f->book2_len, 1, f->tree_end, book);
	while (e < limit) {
		int rank, word;
		if (b.bit >= ((size_t)(f->data2 - f->data1) + 2) * 32) return -1;
		rank = next_rank(&b);
		if (rank < 0) return -1;
		word = book[rank] & 255;
		if (word == 128) e += book[rank] >> 8;
		else if (word >= 110 && (word == 124 || word == 126 || word == 122 || word == 130))
			out[e++] = (int16_t)(word == 124 ? 5005 : word == 126 ? 5006 : word == 122 ? 5003 : 5004);   /* :596-599 */
		else out[e++] = (int16_t)plain_level(word);
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------ d3: position lists
 * nhw_decoder.c:93-196 (res1), :201-301 (res5), :306-420 (res6), :425-487 (res3): list bytes -> positions
 * (row << 8 | column), low bit from the bit plane.  The walk edits the list, so it runs on a copy.
 * pos[] has bits_len*8 zero-filled entries (entries the list does not reach stay 0 + their bit). */
static void poslist_decode(const uint8_t *list_in, int len, const uint8_t *bits, int bits_len, uint32_t *pos, int row_step, int mask16, uint8_t *tmp)
{
	uint8_t *b = tmp;
	const int cap = bits_len * 8;
	int n = 0, i, row;
	int dummy_prev = 0;
	if (len > 0) memcpy(b, list_in, (size_t)len);
#define LASTCOL ((int)((n > 0 ? pos[n - 1] : (uint32_t)dummy_prev) & 255))
#define EMIT(v) do { if (n < cap) pos[n] = mask16 ? ((uint32_t)(v) & 0xFFFFu) : (uint32_t)(v); n++; } while (0)
	if (len <= 0) goto planes;
	if (b[0] == 127) row = row_step;
	else { EMIT(b[0] << 1); row = 0; }
	for (i = 1; i < len; i++) {
		if (b[i] >= 128) {                                        /* two positions as column steps from the previous one */
			const int step1 = (b[i] - 128) >> 4, step2 = b[i] & 15;
			int col;
			if (b[i - 1] == 127) { b[i] = 127; row += 2 * row_step; continue; }
			col = LASTCOL + (step1 << 1);
			if (col >= 254) { row += row_step; b[i] = 127; } else EMIT(col + row);
			col += step2 << 1;
			if (col >= 254) { row += row_step; b[i] = 127; } else EMIT(col + row);
		}
		else if (b[i] == 127) row += row_step;
		else {
			if ((b[i] << 1) < LASTCOL && b[i - 1] != 127) row += row_step;
			EMIT((b[i] << 1) + row);
		}
	}
planes:
	for (i = 0; i < cap; i++) pos[i] = mask16 ? ((pos[i] + (uint32_t)bit_of(bits, bits_len, i)) & 0xFFFFu) : pos[i] + (uint32_t)bit_of(bits, bits_len, i);
#undef LASTCOL
#undef EMIT
}

/* ------------------------------------------------------------------------------------------ d4: synthesis filters
 * decoder/filters.c:143-194; lo/hi: M samples each, out: 2M samples */
static void synth_low(const int16_t *lo, int M, int16_t *out)            /* upfilter53I */
{
	int k;
	for (k = 0; k < M - 1; k++) { out[2 * k] = (int16_t)(lo[k] << 3); out[2 * k + 1] = (int16_t)((lo[k + 1] + lo[k]) << 2); }
	out[2 * M - 2] = (int16_t)(lo[M - 1] << 3); out[2 * M - 1] = (int16_t)(lo[M - 1] << 3);
}
static void synth_high(const int16_t *hi, int M, int16_t *out, int normalise)   /* upfilter53III / upfilter53VI */
{
	int k;
	for (k = 0; k < M; k++) {
		int ev, od;
		if (k == 0) { ev = out[0] - (hi[0] << 2); od = out[1] + (5 * hi[0] - hi[1]); }
		else if (k < M - 1) { ev = out[2 * k] - ((hi[k] + hi[k - 1]) << 1); od = out[2 * k + 1] + (6 * hi[k] - hi[k + 1] - hi[k - 1]); }
		else { ev = out[2 * k] - ((hi[M - 1] + hi[M - 2]) << 1); od = out[2 * k + 1] + (5 * hi[M - 1] - hi[M - 2]); }
		ev = (int16_t)ev; od = (int16_t)od;
		if (normalise) { if (ev > 0) ev = (int16_t)(ev + 32); ev >>= 6; if (od > 0) od = (int16_t)(od + 32); od >>= 6; }
		out[2 * k] = (int16_t)ev; out[2 * k + 1] = (int16_t)od;
	}
}
/* one pass over `rows` rows of a plane with stride `st`: row = [low half | high half] of n samples */
static void synth_rows(const int16_t *src, int16_t *dst, int st, int rows, int n, int normalise)
{
	int r;
	for (r = 0; r < rows; r++) { synth_low(src + (size_t)r * st, n / 2, dst + (size_t)r * st); synth_high(src + (size_t)r * st + n / 2, n / 2, dst + (size_t)r * st, normalise); }
}
static void transpose(const int16_t *src, int16_t *dst, int st, int n)
{
	int i, j;
	for (i = 0; i < n; i++) for (j = 0; j < n; j++) dst[(size_t)i * st + j] = src[(size_t)j * st + i];
}

static inline int lap8(const int16_t *p, int st)        /* 8*centre - the 8 neighbours (nhw_decoder.c:789-795, :1068) */
{
	return (p[0] << 3) - p[-1] - p[1] - p[-st] - p[st] - p[-st - 1] - p[st - 1] - p[-st + 1] - p[st + 1];
}
static inline uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* ------------------------------------------------------------------------------------------ luma */
typedef struct {
	pool *mem;
	const nhw_file *f;
	uint8_t *ll;            /* res_comp */
	uint8_t *exw;           /* private copy of exw_Y (edited while read) */
	int exw_at;             /* the reference's exw1 cursor */
	int16_t *chroma_syms;   /* im_nhw3 */
} dctx;

static int decode_luma(dctx *d, uint8_t *ybytes)
{
	const nhw_file *f = d->f;
	const int q = f->q;
	int16_t *stream = (int16_t *)grab(d->mem, 4 * Q * 2 + 64);    /* im_process in its first life */
	int16_t *a = (int16_t *)grab(d->mem, 4 * Q * 2);              /* im_jpeg */
	int16_t *b = stream;                                          /* im_process */
	uint32_t *p1 = NULL, *p3 = NULL, *p5 = NULL, *p6 = NULL;
	uint8_t *tmp = (uint8_t *)grab(d->mem, 70000 + (size_t)f->res6_len);
	uint16_t *marks;
	int i, j, k, n1 = 0, n3 = 0, n5 = 0, n6 = 0, carry, nmarks = 0;
	if (!stream || !a || !tmp) return NHWO_E_ALLOC;
	if (vlc_luma(f, stream)) return NHWO_E_SPACE;

	/* strips of 4 columns, serpentine down the rows (:71-91) */
	for (k = 0, carry = 0; k < W / 4; k++)
		for (i = 0; i < W; i += 2, carry += 8) {
			int16_t *r0 = a + (size_t)i * W + 4 * k, *r1 = r0 + W;
			r0[0] = stream[carry]; r0[1] = stream[carry + 1]; r0[2] = stream[carry + 2]; r0[3] = stream[carry + 3];
			r1[3] = stream[carry + 4]; r1[2] = stream[carry + 5]; r1[1] = stream[carry + 6]; r1[0] = stream[carry + 7];
		}
	probe(2, a, 4 * Q * 2);
	/* `carry` now stands for the reference's `count`, whose left-over value reaches the loop at :571 */

	if (q > 12) { n1 = f->res1_bits * 8; p1 = (uint32_t *)grab(d->mem, (size_t)n1 * 4 + 64); poslist_decode(f->res1, f->res1_len, f->res1_bit, f->res1_bits, p1, 256, 1, tmp); carry = (f->res1_bits - 1) * 8; if (carry < 0) carry = 0; }
	if (q >= 21) { n5 = f->res5_bits * 8; p5 = (uint32_t *)grab(d->mem, (size_t)n5 * 4 + 64); poslist_decode(f->res5, f->res5_len, f->res5_bit, f->res5_bits, p5, 256, 1, tmp); carry = (f->res5_bits - 1) * 8; if (carry < 0) carry = 0; }
	if (q > 21) { n6 = f->res6_bits * 8; p6 = (uint32_t *)grab(d->mem, (size_t)n6 * 4 + 64); poslist_decode(f->res6, f->res6_len, f->res6_bit, f->res6_bits, p6, 256, 0, tmp); carry = (f->res6_bits - 1) * 8; if (carry < 0) carry = 0; }
	if (q >= 19) { n3 = f->res3_bits * 8; p3 = (uint32_t *)grab(d->mem, (size_t)n3 * 4 + 64); poslist_decode(f->res3, f->res3_len, f->res3_bit, f->res3_bits, p3, 256, 1, tmp); carry = (f->res3_bits * 2 - 2) * 4; if (carry < 0) carry = 0; }
	(void)n1; (void)n3; (void)n5; (void)n6;

	/* pattern symbols back to coefficients, raster order, in place (:493-560) */
	for (i = 0; i < H; i++) for (j = 0; j < W; j++) {
		int16_t *p = a + (size_t)i * W + j;
		switch (*p) {
		case 1008: p[-1] = 5; p[1] = 5; p[0] = (int16_t)(j < H ? 5 : 6); break;
		case 1009: p[-1] = -5; p[1] = -5; p[0] = (int16_t)(j < H ? -6 : -7); break;
		case 1010: p[0] = 5; p[1] = 5; p[W] = 5; p[W + 1] = 5; break;
		case 1011: p[0] = -5; p[1] = -5; p[W] = -5; p[W + 1] = -5; break;
		case 1006: p[0] = -6; p[1] = -6; break;
		case 1007: p[0] = 6; p[1] = 6; break;
		default: break;
		}
	}
	for (i = H; i < W; i++) for (j = 0; j < H; j++) {
		int16_t *p = a + (size_t)i * W + j;
		switch (*p) {
		case 1008: p[-1] = 5; p[0] = 6; p[1] = 5; break;
		case 1009: p[-1] = -5; p[0] = -7; p[1] = -5; break;
		case 1006: p[0] = -7; p[1] = -7; break;
		case 1007: p[0] = 7; p[1] = 7; break;
		default: break;
		}
	}
	for (i = H; i < W; i++) for (j = H; j < W; j++) {             /* :562-616 */
		int16_t *p = a + (size_t)i * W + j;
		if (*p > 1000) {
			switch (*p) {
			case 1008: p[-1] = 5; p[0] = 6; p[1] = 5; break;
			case 1009: p[-1] = -5; p[0] = -7; p[1] = -5; break;
			case 1006: p[-H] = -7; p[-3 * H] = -7; p[0] = 0; break;
			case 1007: p[-H] = 7; p[-3 * H] = 7; p[0] = 0; break;
			default: break;
			}
		}
		else if (iabs(*p) > 8 && iabs(*p) < 16 && q < 23 && j > H && j < W - 1) {
			if (iabs(p[-1]) < 8) carry++;
			if (iabs(p[1]) < 8) carry++;
			if (iabs(p[-W]) < 8) carry++;
			if (iabs(p[W]) < 8) carry++;
			if (carry >= 2) *p = (int16_t)(*p > 0 ? *p + 1 : *p - 1);
			carry = 0;
		}
	}

	for (i = 0, k = 0; i < H / 2; i++) for (j = 0; j < H / 2; j++) a[(size_t)i * W + j] = d->ll[k++];      /* LL2, :618-625 */

	if (q > 17) {                                                 /* odd-LL tags, :627-654 */
		int row = 0;
		for (i = 0; i < f->res4_len; i++) {
			const int v = f->res4[i];
			int at;
			if (v == 128) { row++; continue; }
			at = (row << 9) + (v > 128 ? v - 129 : v - 1);
			for (k = 0; k < 4; k++) if (!(a[at + k] & 1)) a[at + k]++;
			if (v > 128) row++;
		}
	}
	for (i = 0, d->exw_at = 0; i < f->exw_len; i += 3, d->exw_at += 3) {     /* out-of-range LL2 samples, :656-668 */
		int v;
		if (!d->exw[i] && !d->exw[i + 1]) break;
		if (d->exw[i + 1] >= 128) { v = d->exw[i + 2] + 255; d->exw[i + 1] = (uint8_t)(d->exw[i + 1] - 128); } else v = -d->exw[i + 2];
		a[(d->exw[i] << 9) + d->exw[i + 1]] = (int16_t)v;
	}

	probe(3, a, 4 * Q * 2);
	{                                                             /* isolated level-2 coefficients shrink by one (:670-721) */
		const int diag = q <= 16 ? 16 : 8;
		for (i = 1; i < H - 1; i++) for (j = 1; j < H - 1; j++) {
			int16_t *p = a + (size_t)i * W + j;
			if (iabs(*p) <= 8) continue;
			if (iabs(p[-W - 1]) > diag || iabs(p[-W]) > 8 || iabs(p[-W + 1]) > diag || iabs(p[-1]) > 8 || iabs(p[1]) > 8 ||
			    iabs(p[W - 1]) > diag || iabs(p[W]) > 8 || iabs(p[W + 1]) > diag) continue;
			if (i >= H / 2 || j >= H / 2) *p = (int16_t)(*p > 0 ? *p - 1 : *p + 1);
		}
	}

	probe(4, a, 4 * Q * 2);
	/* level 2 synthesis: rows, transpose, rows with normalisation (wavelet_filterbank.c:52-141) */
	synth_rows(a, b, W, H, H, 0);
	transpose(b, a, W, H);
	synth_rows(a, b, W, H, H, 1);

	probe(5, b, 4 * Q * 2);
#define AT(p) (((int)((p) & 65280u) << 1) + (int)((p) & 255u))
	/* residual lists onto the level-1 LL (:731-787).  Selector bits/pairs: nhw_res*_word */
	if (q >= 21) {
		const int cnt = (f->res5_bits - 1) * 8;
		for (k = 0; k < cnt; k++) if (bit_of(f->res5_word, f->res5_bits, k)) b[AT(p5[k])] -= 3;
		for (k = 0; k < cnt; k++) if (!bit_of(f->res5_word, f->res5_bits, k)) b[AT(p5[k])] += 3;
	}
	if (q > 12) {
		const int amp = q >= 18 ? 5 : q >= 15 ? 7 : 9, cnt = (f->res1_bits - 1) * 8;
		for (k = 0; k < cnt; k++) if (bit_of(f->res1_word, f->res1_bits, k)) b[AT(p1[k])] = (int16_t)(b[AT(p1[k])] - amp);
		for (k = 0; k < cnt; k++) if (!bit_of(f->res1_word, f->res1_bits, k)) b[AT(p1[k])] = (int16_t)(b[AT(p1[k])] + amp);
	}
	if (q >= 19) {
		const int cnt = (f->res3_bits * 2 - 2) * 4;
		int pass;
		for (pass = 0; pass < 4; pass++) {                        /* reference order: selector 1, 0, 2, 3 */
			const int want = pass == 0 ? 1 : pass == 1 ? 0 : pass;
			for (k = 0; k < cnt; k++) {
				const int sel = (f->res3_word[k >> 2] >> (6 - 2 * (k & 3))) & 3;
				int16_t *t;
				if (sel != want) continue;
				t = b + AT(p3[k]);
				if (sel == 1) { t[0] -= 4; t[W] -= 3; }
				else if (sel == 0) { t[0] += 4; t[W] += 3; }
				else if (sel == 2) { t[0] += 2; t[W] += 2; t[2 * W] += 2; }
				else { t[0] -= 2; t[W] -= 2; t[2 * W] -= 2; }
			}
		}
	}
#undef AT

	probe(6, b, 4 * Q * 2);
	/* mark smooth-edge samples of the level-1 LL, in place, raster order (:789-836) */
	for (i = 1; i < H - 1; i++) for (j = 1; j < H - 2; j += 2) {
		int16_t *p = b + (size_t)i * W + j;
		const int r0 = lap8(p, W), r1 = lap8(p + 1, W);
		if (r0 > 41 && r0 < 108 && r1 < 16) p[0] = (int16_t)(p[0] + 16000);
		else if (r0 < -41 && r0 > -108 && r1 > -16) p[0] = (int16_t)(p[0] + 16000);
		else if (r1 > 41 && r1 < 108 && r0 < 16) p[1] = (int16_t)(p[1] + 16000);
		else if (r1 < -41 && r1 > -108 && r0 > -16) p[1] = (int16_t)(p[1] + 16000);
	}
	marks = (uint16_t *)grab(d->mem, Q * 2);
	if (!marks) return NHWO_E_ALLOC;
	for (i = 1; i < H - 1; i++) for (j = 0; j < H; j++) {
		int16_t *p = b + (size_t)i * W + j;
		if (*p > 10000) { marks[nmarks++] = (uint16_t)(i * H + j); *p = (int16_t)(*p - 16000); }
	}

	probe(7, marks, (size_t)nmarks * 2);
	for (i = 0; i < H; i++) for (j = 0; j < H; j++) a[(size_t)i * W + j] = b[(size_t)j * W + i];            /* :850-853 */

	/* level 1, first direction (wavelet_synthesis2, wavelet_filterbank.c:237-357) */
	synth_rows(a, b, W, W, W, 0);
	if (q > 21) {
		const int cnt = (f->res6_bits - 1) * 8;
		for (k = 0; k < cnt; k++) if (bit_of(f->res6_word, f->res6_bits, k)) b[p6[k]] -= 32;
		for (k = 0; k < cnt; k++) if (!bit_of(f->res6_word, f->res6_bits, k)) b[p6[k]] += 32;
		for (k = 0; k < f->char_res1_len; k++) {
			const int v = f->char_res1[2 * k] | (f->char_res1[2 * k + 1] << 8);
			switch (v & 3) {
			case 0: b[(v << 1) + H - 2] += 32; break;
			case 1: b[((v - 1) << 1) + H - 2] -= 32; break;
			case 2: b[((v - 2) << 1) + H - 1] += 32; break;
			default: b[((v - 3) << 1) + H - 1] -= 32; break;
			}
		}
	}
	if (q > 22) {
		for (k = 0; k < f->qs3_len; k++) {
			const uint32_t v = (uint32_t)f->qs3[4 * k] | ((uint32_t)f->qs3[4 * k + 1] << 8) | ((uint32_t)f->qs3[4 * k + 2] << 16) | ((uint32_t)f->qs3[4 * k + 3] << 24);
			if (!(v & 1)) b[v >> 1] += 56; else b[v >> 1] -= 56;
		}
	}
	probe(8, b, 4 * Q * 2);
	transpose(b, a, W, W);

	for (k = 0; k < nmarks; k++) {                                /* 5-tap smoothing at the marked samples, list order (:859-876) */
		int16_t *p = a + ((size_t)(marks[k] >> 8) << 10) + (marks[k] & 255);
		if (iabs(lap8(p, W)) < 116) *p = (int16_t)(((p[0] << 2) + p[-1] + p[1] + p[-W] + p[W] + 4) >> 3);
	}

	probe(9, a, 4 * Q * 2);
	synth_rows(a, b, W, W, W, 1);                                 /* second direction (Y==3) */
	for (i = 0; i < 4 * Q; i++) ybytes[i] = clip8(b[i]);
	return NHWO_OK;
}

/* ------------------------------------------------------------------------------------------ chroma */
static int decode_chroma(dctx *d, int comp, uint8_t *full)
{
	const nhw_file *f = d->f;
	const int q = f->q, thr = q <= 14 ? 35 : 60;
	int16_t *a = (int16_t *)grab(d->mem, Q * 2), *b = (int16_t *)grab(d->mem, Q * 2);
	uint8_t *tall = (uint8_t *)grab(d->mem, 2 * Q);
	const int16_t *s = d->chroma_syms + comp;
	const uint8_t *ll = d->ll + Q / 4 + (comp ? Q / 16 : 0);
	int i, j, k, carry;
	if (!a || !b || !tall) return NHWO_E_ALLOC;

	for (k = 0, carry = 0; k < H / 8; k++)                       /* strips of 8 columns (:904-932 / :1192-1220) */
		for (i = 0; i < H; i += 2, carry += 32) {
			int16_t *r0 = a + (size_t)i * H + 8 * k, *r1 = r0 + H;
			for (j = 0; j < 8; j++) { r0[j] = s[carry + 2 * j]; r1[7 - j] = s[carry + 16 + 2 * j]; }
		}
	for (i = 0, k = 0; i < H / 4; i++) for (j = 0; j < H / 4; j++) a[(size_t)i * H + j] = (int16_t)(ll[k++] + (q > 15 ? 0 : 1));     /* :943-963 */

	d->exw_at += 2;                                               /* :965-981 / :1255-1267 */
	for (i = d->exw_at; i < f->exw_len; i += 3) {
		int v;
		if (!comp) { if (!d->exw[i] && !d->exw[i + 1]) break; d->exw_at += 3; }
		if (d->exw[i + 1] >= 128) { v = d->exw[i + 2] + 255; d->exw[i + 1] = (uint8_t)(d->exw[i + 1] - 128); } else v = -d->exw[i + 2];
		a[(d->exw[i] << 8) + d->exw[i + 1]] = (int16_t)v;
	}

	probe(30 + comp, a, Q * 2);
	/* level 2: rows, transpose, rows normalised (wavelet_filterbank.c:143-213) */
	synth_rows(a, b, H, H / 2, H / 2, 0);
	transpose(b, a, H, H / 2);
	synth_rows(a, b, H, H / 2, H / 2, 1);
	/* (a still holds the level-1 detail bands outside its top-left quarter: the transpose above only
	 * rewrote that quarter) */

	probe(40 + comp, b, Q * 2);
	/* pair / single corrections carried as symbols in the level-1 detail bands (:992-1083) */
	for (i = 0; i < H / 2; i++) for (j = H / 2; j < H; j++) {
		int16_t *p = a + (size_t)i * H + j, *t = b + (size_t)i * H + j - H / 2;
		switch (*p) {
		case 5005: t[0] -= 4; t[1] -= 4; *p = 0; break;
		case 5006: t[0] += 4; t[1] += 4; *p = 0; break;
		case 5003: t[0] -= 6; *p = 0; break;
		case 5004: t[0] += 6; *p = 0; break;
		default: break;
		}
	}
	for (i = H / 2; i < H; i++) for (j = 0; j < H; j++) {
		int16_t *p = a + (size_t)i * H + j, *t = b + (size_t)(i - H / 2) * H + (j < H / 2 ? j : j - H / 2);
		switch (*p) {
		case 5005: t[0] -= 4; t[1] -= 4; *p = 0; break;
		case 5006: t[0] += 4; t[1] += 4; *p = 0; break;
		case 5003: t[0] -= 6; *p = 0; break;
		case 5004: t[0] += 6; *p = 0; break;
		default: break;
		}
	}
	probe(42 + comp, b, Q * 2);
	for (i = 0; i < H / 2; i++) for (j = 0; j < H / 2; j++) a[(size_t)i * H + j] = b[(size_t)j * H + i];  /* :1085-1088 */

	/* level 1 */
	synth_rows(a, b, H, H, H, 0);
	transpose(b, a, H, H);
	synth_rows(a, b, H, H, H, 1);

	probe(44 + comp, b, Q * 2);
	for (i = 1; i < H - 1; i++) for (j = 1; j < H - 1; j++) {     /* sharpen, in place, raster order (:1097-1121) */
		int16_t *p = b + (size_t)i * H + j;
		const int r = lap8(p, H);
		if (r > thr) *p = (int16_t)(*p + (r > 160 ? 3 : 2));
		else if (r < -thr) *p = (int16_t)(*p - (r < -160 ? 3 : 2));
	}
	for (i = 0; i < Q; i++) b[i] = clip8(b[i]);
	probe(46 + comp, b, Q * 2);

	/* x2 bilinear: rows first (:1150-1163), then columns (:1181-1196) */
	for (j = 0; j < H; j++) {
		for (i = 0; i < H - 1; i++) {
			tall[(size_t)(2 * i) * H + j] = (uint8_t)b[(size_t)i * H + j];
			tall[(size_t)(2 * i + 1) * H + j] = (uint8_t)((b[(size_t)i * H + j] + b[(size_t)(i + 1) * H + j] + 1) >> 1);
		}
		tall[(size_t)(2 * H - 2) * H + j] = (uint8_t)b[(size_t)(H - 1) * H + j];
		tall[(size_t)(2 * H - 1) * H + j] = (uint8_t)b[(size_t)(H - 1) * H + j];
	}
	for (i = 0; i < W; i++) {
		const uint8_t *r = tall + (size_t)i * H;
		uint8_t *o = full + (size_t)i * W;
		for (j = 0; j < H - 1; j++) { o[2 * j] = r[j]; o[2 * j + 1] = (uint8_t)((r[j] + r[j + 1] + 1) >> 1); }
		o[W - 2] = r[H - 1]; o[W - 1] = r[H - 1];
	}
	return NHWO_OK;
}

/* ------------------------------------------------------------------------------------------ d6: colour
 * write_image_bmp, nhw_decoder_cli.c:135-286.  Output byte order as the reference writes it ("R" first). */
void nhwo_dec_color(const uint8_t *y, const uint8_t *u, const uint8_t *v, int q, uint8_t *out)
{
	static const float inv_low[17] = { 0.0f, 2.060881f, 1.985939f, 1.916257f, 1.820444f, 1.741126f, 1.665887f, 1.587597f, 1.521263f,
		1.392014f, 1.281502f, 1.190611f, 1.177434f, 1.186945f, 1.138331f, 1.048174f, 1.012139f };       /* index = quality 1..16 (:229-244) */
	int i;
	for (i = 0; i < 4 * Q; i++) {
		int R, G, B;
		if (q >= 20) {
			const int Y = y[i], U = u[i] - 128, V = v[i] - 128;
			R = (int)(Y + 1.402 * V + 0.5f); G = (int)(Y - 0.34414 * U - 0.71414 * V + 0.5f); B = (int)(Y + 1.772 * U + 0.5f);
		}
		else if (q >= 18) {
			const float yinv = q == 19 ? 1.025641f : 1.075269f;
			const float Yq = (float)(y[i] * yinv);
			const int U = u[i] - 128, V = v[i] - 128;
			R = (int)(Yq + 1.402 * V + 0.5f); G = (int)(Yq - 0.34414 * U - 0.71414 * V + 0.5f); B = (int)(Yq + 1.772 * U + 0.5f);
		}
		else if (q == 17) {
			const float yinv = 1.063830f;
			const int Y = y[i], U = u[i] - 128, V = v[i] - 128;
			R = (int)((Y + 1.402 * V) * yinv + 0.5f); G = (int)((Y - 0.34414 * U - 0.71414 * V) * yinv + 0.5f); B = (int)((Y + 1.772 * U) * yinv + 0.5f);
		}
		else {
			const float yinv = inv_low[q];
			const int Y = y[i] * 298, U = u[i], V = v[i];
			R = ((int)((Y + 409 * V + (-56992 - 128)) * yinv + 128.5f)) >> 8;
			G = ((int)((Y - 100 * U - 208 * V + (34784 - 128)) * yinv + 128.5f)) >> 8;
			B = ((int)((Y + 516 * U + (-70688 - 128)) * yinv + 128.5f)) >> 8;
		}
		out[3 * i] = clip8(R); out[3 * i + 1] = clip8(G); out[3 * i + 2] = clip8(B);
	}
}

/* the 54-byte header the CLI writes (nhw_decoder_cli.c:61-65 patched by setup_bmp_header :293-312) */
void nhwo_dec_bmp_header(uint8_t h[54])
{
	static const uint8_t base[54] = { 66,77,54,0,12,0,0,0,0,0, 54,0,0,0,40,0,0,0,0,2, 0,0,0,2,0,0,1,0,24,0, 0,0,0,0,0,0,12,0,0,0 };
	const uint32_t bytes = 512u * 512u * 3u, total = bytes + 54u;
	memcpy(h, base, 54);
	h[2] = (uint8_t)total; h[3] = (uint8_t)(total >> 8); h[4] = (uint8_t)(total >> 16); h[5] = (uint8_t)(total >> 24);
	h[18] = 0; h[19] = 2; h[20] = 0; h[21] = 0;
	h[22] = 0; h[23] = 2; h[24] = 0; h[25] = 0;
	h[28] = 24; h[29] = 0; h[30] = 0; h[31] = 0;
	h[34] = (uint8_t)bytes; h[35] = (uint8_t)(bytes >> 8); h[36] = (uint8_t)(bytes >> 16); h[37] = (uint8_t)(bytes >> 24);
}

/* ------------------------------------------------------------------------------------------ entry points */
int nhwo_decode_planes(const uint8_t *nhw, size_t len, uint8_t *planes, int *quality)
{
	nhw_file f;
	pool mem;
	dctx d;
	uint8_t *code;
	int rc;
	if (parse_container(nhw, len, &f)) return NHWO_E_SPACE;
	if (quality) *quality = f.q;
	mem.cap = (size_t)24 << 20; mem.used = 0;
	mem.base = (uint8_t *)calloc(1, mem.cap);
	if (!mem.base) return NHWO_E_ALLOC;
	memset(&d, 0, sizeof d);
	d.mem = &mem; d.f = &f;
	d.ll = (uint8_t *)grab(&mem, 96 * H + 64);
	code = (uint8_t *)grab(&mem, (size_t)f.ch_res_len + 64);
	d.exw = (uint8_t *)grab(&mem, (size_t)f.exw_len + 64);
	memcpy(code, f.ch_res, (size_t)f.ch_res_len);
	memcpy(d.exw, f.exw, (size_t)f.exw_len);
	ll_expand(&f, code, d.ll);
	rc = decode_luma(&d, planes);
	if (!rc) {
		d.chroma_syms = (int16_t *)grab(&mem, 2 * Q * 2 + 256);
		if (vlc_chroma(&f, d.chroma_syms)) rc = NHWO_E_SPACE;
	}
	if (!rc) rc = decode_chroma(&d, 0, planes + 4 * Q);
	if (!rc) rc = decode_chroma(&d, 1, planes + 8 * Q);
	free(mem.base);
	return rc;
}

int nhwo_decode(const uint8_t *nhw, size_t len, uint8_t *bgr, int *quality)
{
	uint8_t *planes = (uint8_t *)malloc(12 * Q);
	int q = 0, rc;
	if (!planes) return NHWO_E_ALLOC;
	rc = nhwo_decode_planes(nhw, len, planes, &q);
	if (!rc) nhwo_dec_color(planes, planes + 4 * Q, planes + 8 * Q, q, bgr);
	if (quality) *quality = q;
	free(planes);
	return rc;
}
