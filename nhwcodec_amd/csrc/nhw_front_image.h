/*
 * nhw_front_image.h -- THE fused front kernel of the encode path (included by nhw_front.hip):
 * colour conversion + 4:2:0 + luma pre-filter (quality 17..21) + both directions of the level-1 analysis,
 * one workgroup per image, walking the image top to bottom with a rolling window of rows in LDS.
 *
 * Algorithm sources (behaviour only):
 *   colour        rcanut/nhwcodec encoder/colorspace.c:55-260
 *   pre-filter    encoder/image_processing.c:558-837, 1927-1990 (quality 17..21 branch)
 *   filterbank    encoder/wavelet_filterbank.c:52-184, encoder/filters.c:55-114, 203-287, 346-386
 *
 * Why a walk: a band of 16 output rows needs 37 rows of the horizontal pass, 39 rows of luma; independent bands converted and
 * contrasted 22 % of their rows twice, and the pre-filter's carry needed two pre-pass kernels to hand every row its entry state.
 * Here band b keeps what band b+1 shares with it (2 luma rows, 5 horizontal-pass rows, 1 row of filtered chroma, the carry at the
 * end of its last row): no row is converted, contrasted or filtered twice, the carry simply runs on, and the next band's BGR rows
 * are in flight (in registers) while this band's vertical pass runs.
 *
 * Everything is integer / byte streaming work: no MFMA.  The arithmetic is done two pixels to a dword (packed 16-bit) wherever the
 * values fit, bytes four to a dword in the chroma filters (v_lerp_u8: the [1 2 1]/4 filter is two byte-wise averages), and the colour
 * conversion in single precision on v_cvt_f32_ubyteN / v_fma_f32 / v_cvt_pk_u8_f32 -- exact by construction, see convert16().
 */
#ifndef NHW_FRONT_IMAGE_H
#define NHW_FRONT_IMAGE_H

namespace nhw {

#define FI_NT    512                 /* threads of a workgroup: two workgroups share a CU (78 KB of LDS each) */
#define FI_RS    514                 /* LDS row stride in shorts (257 dwords: a lane per row walks over consecutive banks) */
#define FI_RD    (FI_RS / 2)         /* the same in dwords */
/* The LUMA rows have a layout of their own: one unused dword behind every 32 (dword d of a row sits at d + (d >> 5)), 265 dwords a row.
 * Phase 0 writes a row 32 lanes side by side, eight dwords a lane: unskewed, lane g's dwords 8g + e fall on banks (8g + e) mod 32 -- four
 * banks for the 32 lanes of a store instruction, an 8-way conflict on every one of the band's luma stores (SQ_LDS_BANK_CONFLICT: half of the
 * kernel's 267 M conflict cycles a launch in round 4).  With the skew lane g starts at bank 8g + (g >> 2): 32 different banks.  265 = 9 mod 32
 * is odd, so the phases that put a lane on every row still walk over 32 different banks. */
#define FI_YRD   265
#define FI_YRS   (2 * FI_YRD)
#define FI_YP(d) ((d) + ((d) >> 5))  /* place of dword d (>= 0) of a luma row */
#define FI_BR    32                  /* new image rows per band = 16 output rows of every sub-band */
#define FI_YROWS 35                  /* luma rows 32b .. 32b+33 (index = row - 32b) + one spare row (index 34: row 32b+32 as it was before the pair rules) */
#define FI_KROWS 37                  /* horizontal-pass rows 32b-4 .. 32b+32 (index = row - 32b + 4); rows 5.. hold the contrast / kernel map before that */
#define FI_SEG   32                  /* pixels per carry segment */
#define FI_NSEG  16
#ifndef FI_LOOK
#define FI_LOOK  12                  /* pixels of look-back for a segment's entry state ... */
#define FI_LOOK2 32                  /* ... and of the second look of the lanes whose candidates have not merged by then (round 6): what is still open after it goes
                                      * to the serial replay, which cost 0.17 ms a batch with the first look alone (tools/dev/look_ab.sh, q20 front kernel: one look
                                      * of 8 / 10 / 12 / 16 / 20 pixels 5.7 / 3.5 / 3.01 / 2.88 / 2.89 ms; 12 + 32: 2.84, 16 + 32: 2.86, 10 + 32: 2.85, 12 + 24: 2.85) */
#endif
/* byte offsets into the dynamic LDS block */
#define FI_YB_OFF   16
#define FI_KB_OFF   (FI_YB_OFF + FI_YROWS * FI_YRS * 2)
#define FI_STG_OFF  (FI_KB_OFF + 5 * FI_RS * 2)             /* chroma staging: 32 rows x (256 U + 256 V) bytes, over rows 5..20: the rows the SECOND half of the contrast pass fills */
#define FI_TAB_OFF  (FI_KB_OFF + FI_KROWS * FI_RS * 2)
#define FI_PT_OFF   FI_TAB_OFF                              /* 225 x 12 B: pair-rule entries (three dwords: a stride of 3 spreads the entries over all 32 banks; at 16 B they shared eight) */
#define FI_CA_OFF   (FI_PT_OFF + 2704)                      /* 405 B (+3): class of a kernel value, clamped to -202 .. 202 */
#define FI_CB_OFF   (FI_CA_OFF + 408)                       /* the same x 12 */
#define FI_EN_OFF   (FI_CB_OFF + 408)                       /* 512 B: entry state of every carry segment of the band */
#define FI_CR_OFF   (FI_EN_OFF + 512)                       /* 2 x 512 B: the last row of filtered chroma, kept for the next band (written by one band while the row of the band before is read) */
#define FI_MISC_OFF (FI_CR_OFF + 1024)
#define FI_LDS_BYTES (FI_MISC_OFF + 64)
static_assert(FI_STG_OFF % 16 == 0 && FI_CR_OFF % 16 == 0 && FI_PT_OFF % 16 == 0, "16-byte pieces");
static_assert(32 * 512 <= 16 * FI_RS * 2, "the chroma staging fits into rows 5..20 of the contrast map");
static_assert(FI_LDS_BYTES <= 81920, "two workgroups to a CU");

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s(uint32_t x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ u16x2 as_us(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ uint32_t as_w(s16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t as_w(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
/* (lo half of a) | (lo half of b) << 16, and the same of the high halves */
__device__ __forceinline__ uint32_t pack_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
/* (hi half of a) | (lo half of b) << 16 */
__device__ __forceinline__ uint32_t pack_hl(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040302u); }
/* packed 16-bit minimum / maximum as ONE instruction each (left to itself the compiler turns them into compares and selects on the halves) */
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_mul_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
/* byte-wise (a + b) >> 1 and (a + b + 1) >> 1 on four bytes */
__device__ __forceinline__ uint32_t avg_dn(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0u); }
__device__ __forceinline__ uint32_t avg_up(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
/* byte-wise (a + 2 b + c + 2) >> 2 = avg_up(avg_dn(a, c), b): with a + c even the inner average is exact, with a + c odd
 * a + 2b + c + 2 is odd and losing the half changes nothing under the floor (checked over all byte triples on the device) */
__device__ __forceinline__ uint32_t tri121(uint32_t a, uint32_t b, uint32_t c) { return avg_up(avg_dn(a, c), b); }

/* ------------------------------------------------------------------------------------------------
 * 16 pixels (48 BGR bytes in 12 dwords) -> luma, two pixels to a dword, and (uv) the U and V bytes, four pixels to a dword.
 * colorspace.c:55-214.  FAMILY 0: quality >= 20, 1: 18 / 19 (luma scaled by yq), 2: 17 (everything scaled by 0.94).
 *
 * Single precision is exact here because every sum of products is an integer below 2^24:
 *   luma, q >= 20: the reference's (int)(0.299 b0 + 0.587 b1 + 0.114 b2 + 0.5f) is floor(s / 1000), s = 299 b0 + 587 b1 + 114 b2 + 500,
 *     except where the division is exact (one triple in a thousand: there the double rounding of the three products decides and the
 *     lane takes the reference's arithmetic).  floor(s / 1000) comes out of ONE fma: (s + 0.11) * 0.001f + (2^23 - 0.5) is rounded to
 *     the integer 2^23 + floor(s / 1000) (the fraction of (s + 0.11) / 1000 lies in [1.1e-4, 0.9992], the constant's error is 1.2e-5),
 *     and the low 16 bits of that float ARE the result; s - 1000 y < 0.5 finds the exact divisions.
 *   chroma, q >= 18: the reference's (int)(cb + 128.5f) (cb >= 0) / (int)(cb + 128.4f) (cb < 0), cb = su / 10000 through a float, is
 *     floor((su + 1285000 or 1284000) / 10000) (checked on all 2^24 triples: nhw_front.hip, convert_uv); v_cvt_pk_u8_f32 rounds to
 *     nearest even and saturates to 0 .. 255 (probed on the device), so the floor is rne(su * 1e-4f + bias - 0.5 + 4.6e-5): the true
 *     fractions are multiples of 1e-4, the float error is below 1.2e-5.  One instruction converts, clips and packs the byte.
 * The exhaustive test (all 2^24 triples, tests/test_gpu_parity.py) runs this function through k_color.
 * ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ float ubf(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }   /* v_cvt_f32_ubyte<k> */
/* four pixels = three dwords at a time (few values alive at once: the kernel has 128 registers a lane).  The sums run as packed single
 * precision (v_pk_fma_f32: two lanes of arithmetic an instruction): two pixels side by side for the luma, U beside V for the chroma. */
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk2(float a, float b) { return (f32x2){ a, b }; }
template <int FAMILY>
__device__ __forceinline__ void convert4(uint32_t w0, uint32_t w1, uint32_t w2, float yq, uint32_t yw[2], uint32_t &uw, uint32_t &vw, bool uv)
{
	const uint32_t wv[3] = { w0, w1, w2 };
	float f[12];
#pragma unroll
	for (int b = 0; b < 12; b++) f[b] = ubf(wv[b >> 2], b & 3);
	uint32_t ym[4];
#pragma unroll
	for (int p = 0; p < 2; p++) {                                   /* pixels 2p, 2p+1 */
		const f32x2 B0 = pk2(f[6 * p], f[6 * p + 3]), B1 = pk2(f[6 * p + 1], f[6 * p + 4]), B2 = pk2(f[6 * p + 2], f[6 * p + 5]);
		if (FAMILY == 0) {
			const f32x2 s = pk_fma(B2, pk2(114.f, 114.f), pk_fma(B1, pk2(587.f, 587.f), pk_fma(B0, pk2(299.f, 299.f), pk2(500.109375f, 500.109375f))));
			const f32x2 m = pk_fma(s, pk2(0.001f, 0.001f), pk2(8388607.5f, 8388607.5f));
			const f32x2 r = pk_fma(m - pk2(8388608.f, 8388608.f), pk2(-1000.f, -1000.f), s);
#pragma unroll
			for (int h = 0; h < 2; h++) {
				uint32_t bits = __float_as_uint(h ? m.y : m.x);
				if ((h ? r.y : r.x) < 0.5f) {                           /* s is a multiple of 1000: the reference's double arithmetic decides */
					const double ly = 0.299 * (double)(h ? B0.y : B0.x) + 0.587 * (double)(h ? B1.y : B1.x) + 0.114 * (double)(h ? B2.y : B2.x);
					bits = (uint32_t)(int)(ly + 0.5f);
				}
				ym[2 * p + h] = bits;
			}
		} else {
			/* q 17..19: (int)(ly x scale + 0.5) in double.  The same product in single precision is off by less than 1e-4, so its floor is the
			 * answer unless it lands within 2.5e-4 of an integer (5e-4 of the triples): those lanes take the double path. */
			const f32x2 s = pk_fma(B2, pk2(114.f, 114.f), pk_fma(B1, pk2(587.f, 587.f), B0 * pk2(299.f, 299.f)));
			const float c = (FAMILY == 1 ? yq : 0.94f) * 0.001f;
			const f32x2 v = s * pk2(c, c) + pk2(0.5f, 0.5f);
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const float vv = h ? v.y : v.x, fl = floorf(vv), fr = vv - fl;
				int y = (int)fl;
				if (!(fr > 2.5e-4f && fr < 1.f - 2.5e-4f)) {
					const double ly = 0.299 * (double)(h ? B0.y : B0.x) + 0.587 * (double)(h ? B1.y : B1.x) + 0.114 * (double)(h ? B2.y : B2.x);
					y = FAMILY == 1 ? (int)(ly * yq + 0.5f) : (int)(ly * 0.94 + 0.5f);
				}
				ym[2 * p + h] = (uint32_t)y;
			}
		}
	}
	yw[0] = pack_lo(ym[0], ym[1]); yw[1] = pack_lo(ym[2], ym[3]);
	if (!uv) return;
	uw = 0; vw = 0;
#pragma unroll
	for (int e = 0; e < 4; e++) {
		const float b0 = f[3 * e], b1 = f[3 * e + 1], b2 = f[3 * e + 2];
		if (FAMILY != 2) {
			const f32x2 suv = pk_fma(pk2(b2, b2), pk2(5000.f, -813.f), pk_fma(pk2(b1, b1), pk2(-3313.f, -4187.f), pk2(b0, b0) * pk2(-1687.f, 5000.f)));
			/* the bias: 127.90005 below zero, + 0.1 from zero on.  The sums are integers, so clamp(s + 1) to 0 .. 1 is the step (one packed add with the
			 * clamp modifier for U and V together; a compare and a select each otherwise) */
			f32x2 step;
			asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(step) : "v"(suv), "v"(pk2(1.f, 1.f)));
			const f32x2 t = pk_fma(suv, pk2(1e-4f, 1e-4f), pk_fma(step, pk2(0.1f, 0.1f), pk2(127.90005f, 127.90005f)));
			uw = __builtin_amdgcn_cvt_pk_u8_f32(t.x, e, uw);
			vw = __builtin_amdgcn_cvt_pk_u8_f32(t.y, e, vw);
		} else {
			/* q17: 0.94 x the same sums through a float (colorspace.c:196-214).  Exactly, the value is N / 10^6, N = 94 s + 128.5e6 (128.4e6 below zero), and
			 * the reference's float form can only differ from floor(N / 10^6) within 1e-4 of an integer (convert_uv<2>).  x = s x 9.4e-5f + bias is off by
			 * less than 2e-5: further than 1.3e-4 from an integer its floor is the answer; the 2.6e-4 of the values that are closer take the reference's
			 * arithmetic.  (All 2^24 triples: the colour test at q17.) */
			const f32x2 suv = pk_fma(pk2(b2, b2), pk2(5000.f, -813.f), pk_fma(pk2(b1, b1), pk2(-3313.f, -4187.f), pk2(b0, b0) * pk2(-1687.f, 5000.f)));
			f32x2 step;
			asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(step) : "v"(suv), "v"(pk2(1.f, 1.f)));
			const f32x2 x = pk_fma(suv, pk2(9.4e-5f, 9.4e-5f), pk_fma(step, pk2(0.1f, 0.1f), pk2(128.4f, 128.4f)));
			const f32x2 fl = x - pk2(0.5f, 0.5f);
			if (__builtin_fabsf(x.x - __builtin_rintf(x.x)) < 1.3e-4f || __builtin_fabsf(x.y - __builtin_rintf(x.y)) < 1.3e-4f) {
				const uint8_t px[3] = { (uint8_t)(wv[(3 * e) >> 2] >> (8 * ((3 * e) & 3))), (uint8_t)(wv[(3 * e + 1) >> 2] >> (8 * ((3 * e + 1) & 3))), (uint8_t)(wv[(3 * e + 2) >> 2] >> (8 * ((3 * e + 2) & 3))) };
				int U, V;
				convert_uv<2>(px, yq, U, V);
				uw = (uw & ~(0xFFu << (8 * e))) | ((uint32_t)U << (8 * e)); vw = (vw & ~(0xFFu << (8 * e))) | ((uint32_t)V << (8 * e));
			} else {
				uw = __builtin_amdgcn_cvt_pk_u8_f32(fl.x, e, uw);
				vw = __builtin_amdgcn_cvt_pk_u8_f32(fl.y, e, vw);
			}
		}
	}
}
template <int FAMILY>
__device__ __forceinline__ void convert16(const uint32_t wv[12], float yq, uint32_t yw[8], uint32_t uw[4], uint32_t vw[4], bool uv)
{
#pragma unroll
	for (int c = 0; c < 4; c++) {
		convert4<FAMILY>(wv[3 * c], wv[3 * c + 1], wv[3 * c + 2], yq, yw + 2 * c, uw[c], vw[c], uv);
		__builtin_amdgcn_sched_barrier(0);                             /* one chunk after the other: interleaved, the four of them need more registers than there are */
	}
}

/* horizontal [1 2 1]/4 at the even pixels of 16 (colorspace.c:220-234): c[0..3] = the bytes of pixels 0..15, left = a dword whose top byte is
 * pixel -1 (for the first group of a row: pixel 1, which turns the filter into the reference's (c0 + c1 + 1) >> 1) -> eight bytes */
__device__ __forceinline__ uint2 chroma_h8(const uint32_t c[4], uint32_t left)
{
	const uint32_t e01 = __builtin_amdgcn_perm(c[1], c[0], 0x06040200u), o01 = __builtin_amdgcn_perm(c[1], c[0], 0x07050301u);
	const uint32_t e23 = __builtin_amdgcn_perm(c[3], c[2], 0x06040200u), o23 = __builtin_amdgcn_perm(c[3], c[2], 0x07050301u);
	const uint32_t p01 = __builtin_amdgcn_alignbyte(o01, left, 3), p23 = __builtin_amdgcn_alignbyte(o23, o01, 3);   /* the odd pixels one to the left */
	return make_uint2(tri121(p01, e01, o01), tri121(p23, e23, o23));
}

/* packed 16-bit helpers of the two filter passes (filters.c:88-287, 346-386), two columns to a dword.
 *
 * INPUT DOMAIN of the level-1 analysis in this file, and why packed 16-bit arithmetic equals the reference's on it.  Let every input
 * sample lie in [-L, U] (L, U >= 0).  Additions, subtractions and multiplications by constants are exact modulo 2^16 whatever the
 * intermediate sums do, and the reference stores the first pass in `short` cells too (filters.c:346-386 writes its `int` sums into the
 * short plane, i.e. modulo 2^16 as well) -- so only the operations that are NOT modular need their operand to be the true integer:
 * the sign tests and right shifts of the second pass (round-half-away, the error diffusion of :246-276, the predict's halving).
 *   * second pass, low-pass rows (downfilter53VI, :203-287): r = 6 x0 + 2 (x-1 + x1) - (x-2 + x2) on first-pass low-pass values, which are
 *     themselves that tap on the input: the 2-D kernel is the outer product of [-1 2 6 2 -1], positive weights (6+2+2)^2 + (1+1)^2 = 104,
 *     negative ones 2 * 10 * 2 = 40, so  -(104 L + 40 U) <= r <= 104 U + 40 L.  pk_diffuse() reads r's sign and r mod 64: r must be
 *     the true value.  Then acc = r + carry, |carry| <= 15 (the reference holds acc in a short: modular, fine), and pk_rnd_half_away(acc, 6)
 *     adds 32 (31 below zero) before the shift: |r| + 15 + 32 <= 32767.  Hence the bound  104 U + 40 L <= 32720,  104 L + 40 U <= 32720
 *     (NHW_ANA512_BOUND in include/nhw_hip.h; nhw_stage_analysis(size 512) checks it and refuses planes outside).
 *   * the other sums are smaller for the same L, U and fit a fortiori: the predict's pair sum x0 + x2 (+1) on first-pass low values is at
 *     most 2 (10 U + 2 L) + 1; the high-pass rows' taps see first-pass differences, |2 x1 - x0 - x2| <= 2 (U + L), so their tap is at most
 *     24 (U + L) + 8 with the rounding offset.
 * The encoder's own luma is 0 .. 255 plus what the pre-filters add (at most +-4 a pixel): 104 * 259 + 40 * 4 = 27 096. */
__device__ __forceinline__ s16x2 pk_rnd_half_away(s16x2 v, int shift) { return (v + (s16x2)(short)(1 << (shift - 1)) + (v >> 15)) >> shift; }
__device__ __forceinline__ s16x2 pk_diffuse(s16x2 r)
{
	const s16x2 s = r >> 15, a = (r ^ s) - s;
	const s16x2 t = (s16x2)(a << 10) >> 10;                           /* |r| mod 64 read as a signed 6-bit number */
	const s16x2 d = (t + ((t >> 15) & (s16x2)(short)3)) >> 2;
	return (d ^ s) - s;
}

/* The kernel is one long loop over the bands; left alone, the compiler computes every phase's per-thread offsets once in front of the loop and
 * then has 40 more values alive than there are registers (they went to scratch memory).  A thread index taken through this at the start of
 * a phase is opaque to it: the few adds and shifts are redone every band, and nothing of one phase is alive in another. */
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

#ifdef NHW_DEV   /* developer builds: 32 rows of a plane in LDS (from row index 1 / 5 on) into a plane in memory, for tools/dev/gpu_front_debug.py */
#define FI_DUMP(kind, base, yl) do { if ((flags & 2) && ((flags >> 4) & 15) == (kind) && keepb) { \
	for (int k_ = t0; k_ < 32 * 256; k_ += FI_NT) { const int rr_ = 1 + (k_ >> 8), o_ = k_ & 255; \
		if (r0 + rr_ < W - ((kind) == 2 || (kind) == 3)) reinterpret_cast<uint32_t *>(keepb + (size_t)img * keep_stride + (size_t)(r0 + rr_) * W)[o_] = reinterpret_cast<const uint32_t *>((base) + (rr_ - 1) * ((yl) ? FI_YRS : FI_RS))[(yl) ? FI_YP(o_) : o_]; } \
	__syncthreads(); } } while (0)
#else
#define FI_DUMP(kind, base, yl) do { } while (0)
#endif
#ifdef NHW_DEV   /* developer builds: clock ticks of thread 0 between the phase boundaries, summed over all workgroups (NHW_FRONT_PROF=1 prints them) */
__device__ unsigned long long g_fi_prof[16];
#define FI_TICK(i) do { if ((flags & 0x10000) && t0 == 0) { const long long now_ = clock64(); acc_[i] += (unsigned)(now_ - tick_); tick_ = now_; } } while (0)
#else
#define FI_TICK(i) do { } while (0)
#endif
#ifdef NHW_DEV   /* developer builds: a switch that ends every band after phase i (tools/dev/gpu_band_ablate.py: the cost of the phases under real contention) */
#define FI_STAMP(i) do { if (((flags >> 8) & 255) == (i)) { __syncthreads(); continue; } } while (0)
#else
#define FI_STAMP(i) do { } while (0)
#endif

/* PRE: with the pre-filter (quality 17..21).  SRC 1: from the BGR bytes (quality 17..23).  SRC 0: from a luma plane (quality 1..16
 * behind their own pre-filter, nhw_low.hip; the analysis stage entry point): no colour, no chroma, PRE = 0.
 * flags bit 0: test switch, every carry segment takes its exact replay instead of the look-back. */
template <int PRE, int SRC, int FAMILY>
__global__ __launch_bounds__(FI_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_front_image(const void *__restrict__ srcb, size_t src_stride, float yq, uint8_t *__restrict__ pub, uint8_t *__restrict__ pvb, size_t c_stride,
                                                     uint8_t *__restrict__ stb, size_t s_stride,
                                                     int16_t *__restrict__ procb, int16_t *__restrict__ jpegb, size_t plane_stride,
                                                     int16_t *__restrict__ ll1b, size_t ll1_stride, int16_t *__restrict__ keepb, size_t keep_stride, int flags)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	int16_t *const ybuf = reinterpret_cast<int16_t *>(lds + FI_YB_OFF);
	int16_t *const kbuf = reinterpret_cast<int16_t *>(lds + FI_KB_OFF);
	uint8_t *const stg = lds + FI_STG_OFF;                          /* row i of it: filtered chroma of image row 32b + 2 + i */
	uint32_t *const ptab = reinterpret_cast<uint32_t *>(lds + FI_PT_OFF);
	uint8_t *const tca = lds + FI_CA_OFF, *const tcb = lds + FI_CB_OFF;
	uint8_t *const entry = lds + FI_EN_OFF;
	uint8_t *const crow = lds + FI_CR_OFF;
	uint8_t *const misc = lds + FI_MISC_OFF;                        /* [0]: carry behind the band's last row, [2 + (b & 1)]: hand-over flag of its last pixel pair */
	const int t0 = threadIdx.x, img = blockIdx.x;

	const uint8_t *const src = (const uint8_t *)srcb + (size_t)img * (SRC ? (size_t)(W * W * 3) : src_stride);
	int16_t *const proc = procb + (size_t)img * plane_stride, *const jpeg = jpegb + (size_t)img * plane_stride;
	int16_t *const ll1 = ll1b + (size_t)img * ll1_stride;

	/* the next band's rows, on their way while this band is worked on: image rows 32b+2 .. 32b+33 */
	constexpr int NPF = SRC ? 6 : 4;
	uint4 pf[NPF];
	auto issue = [&](int b) {
		const int t = opaque(t0);
		/* Rows outside the image are asked for as the nearest row inside and not used: no branches here, so that every address is computed before
		 * the first load goes out (with a branch per item the compiler put a wait for the first item's loads in front of the second item's --
		 * a memory round trip in the open, every band). */
		if (SRC) {
			const uint4 *rp[2];
#pragma unroll
			for (int it = 0; it < 2; it++) {                         /* 16 pixels = 48 bytes an item, two items a thread */
				int row = FI_BR * b + 2 + (t >> 5) + 16 * it;
				row = row < 0 ? 0 : row > W - 1 ? W - 1 : row;
				rp[it] = reinterpret_cast<const uint4 *>(src + (size_t)row * (W * 3) + 48 * (t & 31));
			}
#pragma unroll
			for (int it = 0; it < 2; it++) { pf[3 * it] = rp[it][0]; pf[3 * it + 1] = rp[it][1]; pf[3 * it + 2] = rp[it][2]; }
		} else {
			const uint4 *rp[4];
#pragma unroll
			for (int it = 0; it < 4; it++) {                         /* 8 pixels = 16 bytes an item, four items a thread */
				const int k = t + FI_NT * it;
				int row = FI_BR * b + 2 + (k >> 6);
				row = row < 0 ? 0 : row > W - 1 ? W - 1 : row;
				rp[it] = reinterpret_cast<const uint4 *>(reinterpret_cast<const int16_t *>(src) + (size_t)row * W) + (k & 63);
			}
#pragma unroll
			for (int it = 0; it < 4; it++) pf[it] = *rp[it];
		}
	};
	issue(-1);

	if (PRE) {
		const int t = t0;
		/* The pair rules only ask which of eight magnitude classes the two kernel values are in (the constants of image_processing.c:810-837,
		 * :1927-1990) and their signs: 15 signed classes.  Entry (c0, c1) = 16 bytes: the two luma deltas packed as two 16-bit halves for
		 * hand-over flag 0 and 1, and the flag the pair hands on; filled by evaluating the rules themselves on one representative per class. */
		if (t < PCLS * PCLS) {
			const int c0 = t / PCLS, c1 = t % PCLS;
			const int r0 = pair_class_rep(c0 - 7), r1 = pair_class_rep(c1 - 7);
			ptab[3 * t] = prefilter_pair_delta(r0, r1, 0); ptab[3 * t + 1] = prefilter_pair_delta(r0, r1, 1);
			ptab[3 * t + 2] = (uint32_t)pair_big_flag_fwd(r0, r1);
		}
		if (t < 405) { const int k = t - 202, m = pair_mag_class(iabs(k)), c = k < 0 ? 7 - m : 7 + m; tca[t] = (uint8_t)c; tcb[t] = (uint8_t)(12 * c); }
		if (t == 0) { misc[0] = 0; misc[2] = 0; misc[3] = 0; misc[8] = 0; }
	}

#ifdef NHW_DEV
	long long tick_ = clock64();
	unsigned acc_[14] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
#pragma unroll 1
	for (int b = -1; b < W / FI_BR; b++) {
		const int r0 = FI_BR * b;
		FI_TICK(0);                                     /* image row of luma index 0 */
		/* ---------------------------------------------------------------- phase 0: the prefetched rows -> luma (and filtered chroma) in LDS */
		if (SRC) {
			const int t = opaque(t0);
#pragma unroll
			for (int it = 0; it < 2; it++) {
				const int i = (t >> 5) + 16 * it, g = t & 31, row = r0 + 2 + i;
				uint32_t yw[8], uw[4], vw[4];
				const bool live = row >= 0 && row < W;
				if (live) {
					const uint32_t wv[12] = { pf[3 * it].x, pf[3 * it].y, pf[3 * it].z, pf[3 * it].w, pf[3 * it + 1].x, pf[3 * it + 1].y, pf[3 * it + 1].z, pf[3 * it + 1].w,
					                          pf[3 * it + 2].x, pf[3 * it + 2].y, pf[3 * it + 2].z, pf[3 * it + 2].w };
					convert16<FAMILY>(wv, yq, yw, uw, vw, true);
				} else {
#pragma unroll
					for (int e = 0; e < 8; e++) yw[e] = 0;
#pragma unroll
					for (int e = 0; e < 4; e++) { uw[e] = 0; vw[e] = 0; }
				}
				uint32_t *d = reinterpret_cast<uint32_t *>(ybuf) + (i + 2) * FI_YRD + 8 * g + (g >> 2);   /* FI_YP(8g): eight dwords never straddle a pad */
#pragma unroll
				for (int e = 0; e < 8; e++) d[e] = yw[e];
				/* the pixel on the left of the group comes from the lane on the left (every lane takes part in the shuffle) */
				uint32_t lu = (uint32_t)__shfl_up((int)uw[3], 1), lv = (uint32_t)__shfl_up((int)vw[3], 1);
				if (g == 0) { lu = uw[0] << 16; lv = vw[0] << 16; }
				*reinterpret_cast<uint2 *>(stg + i * 512 + 8 * g) = chroma_h8(uw, lu);
				*reinterpret_cast<uint2 *>(stg + i * 512 + 256 + 8 * g) = chroma_h8(vw, lv);
			}
		} else {
			const int t = opaque(t0);
#pragma unroll
			for (int it = 0; it < 4; it++) {
				const int k = t + FI_NT * it, i = k >> 6, row = r0 + 2 + i;
				const uint4 v = (row >= 0 && row < W) ? pf[it] : make_uint4(0, 0, 0, 0);
				uint32_t *d = reinterpret_cast<uint32_t *>(ybuf) + (i + 2) * FI_YRD + FI_YP(4 * (k & 63));
				d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
			}
		}
		FI_TICK(9);
		__syncthreads();
		if (b >= 0 && b + 1 < W / FI_BR) issue(b + 1);                /* the next band's rows are on their way while this band is worked on */
		FI_TICK(1); FI_STAMP(1);
		/* ---------------------------------------------------------------- 4:2:0: vertical [1 2 1]/4 over image rows 2r-1, 2r, 2r+1 (colorspace.c:241-256) */
		if (SRC) {
			const int t = opaque(t0);
			const int pl = t >> 8, rr = (t >> 4) & 15, c16 = t & 15, r = 16 * b + 1 + rr;   /* chroma rows 16b+1 .. 16b+16; the band before the first: row 0 */
			if (b < 0 ? rr == 15 : r < H) {
				/* image rows 32b+1+2rr, +1, +2: for rr = 0 the first of them is the last row of the band before */
				const uint8_t *sp = stg + 2 * rr * 512 + pl * 256 + 16 * c16;
				const uint4 x0 = *reinterpret_cast<const uint4 *>(rr ? sp - 512 : crow + ((b + 1) & 1) * 512 + pl * 256 + 16 * c16), x1 = *reinterpret_cast<const uint4 *>(sp), x2 = *reinterpret_cast<const uint4 *>(sp + 512);
				uint4 o;
				if (b < 0) o = make_uint4(avg_up(x1.x, x2.x), avg_up(x1.y, x2.y), avg_up(x1.z, x2.z), avg_up(x1.w, x2.w));   /* row 0: (r0 + r1 + 1) >> 1 */
				else o = make_uint4(tri121(x0.x, x1.x, x2.x), tri121(x0.y, x1.y, x2.y), tri121(x0.z, x1.z, x2.z), tri121(x0.w, x1.w, x2.w));
				*reinterpret_cast<uint4 *>((pl ? pvb : pub) + (size_t)img * c_stride + (b < 0 ? 0 : r) * H + 16 * c16) = o;
				if (rr == 15) *reinterpret_cast<uint4 *>(crow + (b & 1) * 512 + pl * 256 + 16 * c16) = x2;
			}
		}
		if (b < 0) {
			/* rows 0 and 1 sit at luma index 32 and 33: move them to 0 and 1, ask for the first band's rows */
			const int t = opaque(t0);
			for (int k = t; k < 2 * FI_YRD; k += FI_NT) { uint32_t *y = reinterpret_cast<uint32_t *>(ybuf); y[k] = y[32 * FI_YRD + k]; }
			issue(0);
			__syncthreads();
			continue;
		}
		const int last_row = r0 + FI_BR < W - 2 ? r0 + FI_BR : W - 2; /* last image row of this band with a kernel value */
		if (PRE) {
			FI_TICK(2);
			/* ------------------------------------------------------------ contrast of image rows 32b+1 .. 32b+32, 8 pixels an item (image_processing.c:568-640).
			 * The rows stay packed, two pixels to a dword (luma is never negative here).  The signed sum 9 x centre - block sum is packed
			 * arithmetic on both pixels of a dword; a pixel's eight absolute differences are four v_sad_u16 against dwords that hold two
			 * neighbours each.  Items are (row, group) with the row fastest: consecutive lanes sit a padded row apart, on consecutive banks.
			 * Two halves: rows 17..32 first -- their map rows (21..36) are free, so this half runs beside the 4:2:0 step above, which still reads
			 * the staging rows --, a barrier, then rows 1..16 over the staging rows. */
			{
			const int t = opaque(t0);
#pragma unroll 1
			for (int it = 0; it < 4; it++) {
				if (it == 2) __syncthreads();                            /* the staging rows become the contrast map */
				const int k = t + FI_NT * (it & 1), gh = k >> 4;
				const int rr = (it < 2 ? 17 : 1) + (k & 15), g = (gh & ~7) | ((gh & 1) << 2) | ((gh >> 1) & 3);   /* 16 lanes a group; the two groups of a half-wavefront 16 banks apart */
				if (r0 + rr > W - 2) continue;
				/* dwords 4g-1 .. 4g+4 of three rows: the first / last of them lies behind a pad where the group starts / ends a block of 32 dwords */
				const uint32_t *ru = reinterpret_cast<const uint32_t *>(ybuf) + (rr - 1) * FI_YRD + 4 * g + (g >> 3), *rm = ru + FI_YRD, *rd = rm + FI_YRD;
				const int om = (g & 7) == 0 ? -2 : -1, op = (g & 7) == 7 ? 5 : 4;
				uint32_t U[6], M[6], D[6], S[6];
				U[0] = ru[om]; M[0] = rm[om]; D[0] = rd[om]; U[5] = ru[op]; M[5] = rm[op]; D[5] = rd[op];
#pragma unroll
				for (int j = 1; j < 5; j++) { U[j] = ru[j - 1]; M[j] = rm[j - 1]; D[j] = rd[j - 1]; }
#pragma unroll
				for (int j = 0; j < 6; j++) S[j] = pk_add16(pk_add16(U[j], M[j]), D[j]);
				uint32_t out[4];
#pragma unroll
				for (int K = 1; K < 5; K++) {
					const uint32_t T = pk_add16(S[K], __builtin_amdgcn_alignbit(S[K], S[K], 16));   /* both halves: the dword's two column sums */
					const uint32_t wsum = pk_add16(T, pack_hl(S[K - 1], S[K + 1]));                  /* + the column on the left (low pixel) / right (high pixel) */
					const s16x2 sum = as_s(as_w(as_us(M[K]) * (u16x2)(unsigned short)9)) - as_s(wsum);
					const uint32_t sgw = pk_max_i16(pk_min_i16(as_w(sum), 0x00010001u), 0xFFFFFFFFu);   /* -1, 0, 1 */
					const s16x2 sg = as_s(sgw);
					const u16x2 ab = as_us(pk_mul_u16(as_w(sum), sgw));
					uint32_t mag[2];
#pragma unroll
					for (int h = 0; h < 2; h++) {
						const uint32_t cc = __builtin_amdgcn_perm(M[K], M[K], h ? 0x03020302u : 0x01000100u);
						const uint32_t side = h ? __builtin_amdgcn_perm(D[K + 1], U[K + 1], 0x05040100u) : __builtin_amdgcn_perm(D[K - 1], U[K - 1], 0x07060302u);
						const uint32_t mids = h ? __builtin_amdgcn_perm(M[K + 1], M[K], 0x05040100u) : __builtin_amdgcn_perm(M[K], M[K - 1], 0x07060302u);
						mag[h] = sad_u16(cc, U[K], sad_u16(cc, D[K], sad_u16(cc, side, sad_u16(cc, mids, 0u))));
					}
					if (K == 1 && g == 0) mag[0] = 0;                     /* column 0 has no kernel value, and what lies in front of the row is not luma: its sum must not spill into column 1's half */
					const u16x2 base = ab * (u16x2)(unsigned short)15 + as_us(mag[0] | (mag[1] << 16));
					out[K - 1] = pk_mul_u16(as_w(base), sgw);               /* the sign of the sum; a zero sum gives no kernel value */
					(void)sg;
				}
				if (g == 0) out[0] &= 0xFFFF0000u;                       /* columns 0 and 511 have none */
				if (g == 63) out[3] &= 0x0000FFFFu;
				uint32_t *d = reinterpret_cast<uint32_t *>(kbuf + (rr + 4) * FI_RS + 8 * g);
				d[0] = out[0]; d[1] = out[1]; d[2] = out[2]; d[3] = out[3];
			}
			if (t < FI_YRD) reinterpret_cast<uint32_t *>(ybuf + 34 * FI_YRS)[t] = reinterpret_cast<const uint32_t *>(ybuf + 32 * FI_YRS)[t];   /* the next band's contrast wants row 32b+32 as it is now */
			}
			__syncthreads();
			FI_TICK(3); FI_STAMP(2); FI_STAMP(3);
			FI_DUMP(3, kbuf + 5 * FI_RS, 0);
			FI_DUMP(4, ybuf + FI_YRS, 1);
			FI_DUMP(5, ybuf, 1);
			FI_DUMP(6, ybuf + 3 * FI_YRS, 1);
			/* ------------------------------------------------------------ carry state at the start of every 32-pixel segment (image_processing.c:641-700).
			 * The carry is a 16-state machine that runs in raster order over the interior of the whole image.  A step maps the 16 states onto at
			 * most five neighbouring ones and a zero sum resets it: run five candidates (5-bit fields of one dword) through the 12 pixels (where they have not merged by then: the 32 pixels) in
			 * front of the segment -- for a row's first segment the end of the row above --; if they end in one state, that is the entry state
			 * whatever came before.  Where they do not (rare), the segments are replayed in order from the one before.  The band's first
			 * segment continues from where the band before stopped. */
			{
				const int t = opaque(t0);
				const int rr = 1 + (t & 31), sg = t >> 5;
				int e = 0;
				if (r0 + rr <= W - 2) {
					if (sg == 0 && rr == 1) e = misc[0];
					else {
						const int16_t *kend = sg ? kbuf + (rr + 4) * FI_RS + 1 + FI_SEG * sg : kbuf + (rr + 3) * FI_RS + (W - 1);   /* the first cell behind the look-back */
						const uint32_t R = 0x108421u;                    /* 1 in each field */
						auto look = [&](const int len) -> int {
							const int16_t *km = kend - len;
							uint32_t x = km[0] == 0 ? 0u : ((((uint32_t)iabs(km[0]) & 15u) * R + 0x418820u) & (15u * R));
							for (int i = 1; i < len; i++) {
								const int vb = km[i];
								const uint32_t nx = (((uint32_t)iabs(vb) & 15u) * R + (((x + 2u * R) >> 2) & (7u * R))) & (15u * R);
								x = vb == 0 ? 0u : nx;
							}
							return (x == (x & 31u) * R && !(flags & 1)) ? (int)(x & 15u) : 0xFF;
						};
						e = look(FI_LOOK);
#ifdef FI_LOOK2
						if (e == 0xFF && !(flags & 1)) e = look(FI_LOOK2);   /* a second, longer look where the first has not merged: only the wavefronts with such a lane pay for it */
#endif
					}
					entry[(rr - 1) * FI_NSEG + sg] = (uint8_t)e;
				}
#ifndef FI_TIMING_NO_SERIAL   /* (developer timing experiment: what the serial replay still costs -- results are wrong without it) */
				if (e == 0xFF) misc[8] = 1;                              /* (cleared again in the pair-rule phase) */
#endif
				__syncthreads();
				if (misc[8]) {
					if (t == 0) {
						for (int s = 1; s < (last_row - r0) * FI_NSEG; s++) {   /* raster order; segment 0 of the band is never open */
							if (entry[s] != 0xFF) continue;
							const int pr = (s - 1) >> 4, ps = (s - 1) & 15, n = ps == FI_NSEG - 1 ? FI_SEG - 2 : FI_SEG;
							const int16_t *km = kbuf + (pr + 5) * FI_RS + 1 + FI_SEG * ps;
							int carry = entry[s - 1];
							for (int i = 0; i < n; i++) { const int vb = km[i]; carry = vb == 0 ? 0 : ((iabs(vb) + ((carry + 2) >> 2)) & 15); }
							entry[s] = (uint8_t)carry;
						}
					}
					__syncthreads();
				}
				if (stb && sg == 0 && r0 + rr <= W - 2) (stb + (size_t)img * s_stride)[r0 + rr] = entry[(rr - 1) * FI_NSEG];   /* compatibility mode replays a few rows from these */
			}
			FI_TICK(4); FI_STAMP(4);
			/* ------------------------------------------------------------ replay the carry: a lane per row and PAIR of segments (sp, sp + 8), side by side in the halves
			 * of a dword, every step packed 16-bit arithmetic.  Eight pixels at a time through registers. */
			if (t0 < 32 * (FI_NSEG / 2)) {
				/* half the workgroup's wavefronts walk the segments, the others wait at the barrier: the walkers are the band's critical path and
				 * share their SIMDs with the other workgroup's wavefronts -- they go first */
				__builtin_amdgcn_s_setprio(3);
				const int t = opaque(t0);
				const int rr = 1 + (t & 31), sp = t >> 5;
				if (r0 + rr <= W - 2) {
					/* a segment's 32 cells are the high half of dword 16 sg of the row, fifteen whole dwords and the low half of dword 16 sg + 16: the
					 * cells come in as dwords (a tenth of the LDS instructions of 16-bit accesses), a step's two cells -- one of either segment -- are
					 * put side by side with one v_perm, and the results go back as dwords, the two half dwords at the ends as 16-bit stores (their other
					 * halves belong to the neighbouring segments' lanes).  The last segment of a row runs two cells over its end: cell 511 is 0 and
					 * stays 0, the cell behind it is padding. */
					uint32_t *da = reinterpret_cast<uint32_t *>(kbuf + (rr + 4) * FI_RS) + (FI_SEG / 2) * sp, *db = da + 128;
					u16x2 carry = { entry[(rr - 1) * FI_NSEG + sp], entry[(rr - 1) * FI_NSEG + sp + FI_NSEG / 2] }, c29 = carry;
					uint32_t la = da[0], lb = db[0];                      /* the dword the chunk starts in */
					uint32_t pend = 0;                                   /* the last step's pair of results: the low halves of the dwords the next chunk starts in */
#pragma unroll 1
					for (int ch = 0; ch < FI_SEG / 8; ch++) {
						uint32_t A[5], B[5];
						A[0] = la; B[0] = lb;
#pragma unroll
						for (int e = 1; e < 5; e++) { A[e] = da[4 * ch + e]; B[e] = db[4 * ch + e]; }
						la = A[4]; lb = B[4];
						uint32_t o[8];
#pragma unroll
						for (int e = 0; e < 8; e++) {                      /* v == 0: |v| + f(carry) <= 4 gives output 0 by itself; only the carry needs the reset */
							const s16x2 v = as_s((e & 1) ? pack_lo(A[(e + 1) >> 1], B[(e + 1) >> 1]) : pack_hi(A[e >> 1], B[e >> 1]));
							const s16x2 sgn = v >> 15;
							const u16x2 a = __builtin_bit_cast(u16x2, (s16x2)((v ^ sgn) - sgn));
							const u16x2 acc = a + ((carry + (u16x2)(2)) >> 2);
							const s16x2 ov = __builtin_bit_cast(s16x2, (u16x2)(acc >> 4));
							o[e] = as_w((s16x2)((ov ^ sgn) - sgn));
							carry = as_us(pk_mul_u16(as_w((u16x2)(acc & (u16x2)(15))), pk_min_u16(as_w(a), 0x00010001u)));
							if (e == 5) c29 = carry;                        /* in the last chunk: the state behind column 510 */
						}
						if (ch == 0) { reinterpret_cast<int16_t *>(da)[1] = (int16_t)(o[0] & 0xFFFF); reinterpret_cast<int16_t *>(db)[1] = (int16_t)(o[0] >> 16); }
						else { da[4 * ch] = pack_lo(pend, o[0]); db[4 * ch] = pack_hi(pend, o[0]); }
#pragma unroll
						for (int e = 1; e < 4; e++) { da[4 * ch + e] = pack_lo(o[2 * e - 1], o[2 * e]); db[4 * ch + e] = pack_hi(o[2 * e - 1], o[2 * e]); }
						pend = o[7];
					}
					reinterpret_cast<int16_t *>(da + 16)[0] = (int16_t)(pend & 0xFFFF); reinterpret_cast<int16_t *>(db + 16)[0] = (int16_t)(pend >> 16);
					if (sp == FI_NSEG / 2 - 1 && r0 + rr == last_row) misc[0] = (uint8_t)c29.y;
				}
				__builtin_amdgcn_s_setprio(0);
			}
			__syncthreads();
			FI_TICK(5); FI_STAMP(5);
			FI_DUMP(2, kbuf + 5 * FI_RS, 0);
			/* ------------------------------------------------------------ pair rules (image_processing.c:810-837, 1927-1990): pixel pairs (1,2), (3,4) .. (509,510) of a row,
			 * four pairs an item; a pair's deltas depend on its two kernel values and on a flag the pair before hands over -- which depends on
			 * that pair's values only, so nothing is serial.  Class of a value, entry of a pair: table look-ups (see the fill above). */
			{
			const int t = opaque(t0);
#pragma unroll 2
			for (int it = 0; it < 4; it++) {
				const int k = t + FI_NT * it, rr = 1 + (k & 31), g = k >> 5;
				if (r0 + rr > W - 2) continue;
				const uint32_t *kr = reinterpret_cast<const uint32_t *>(kbuf + (rr + 4) * FI_RS);
				/* values of columns 8g-1 .. 8g+8: v0 = high half of P[0], v1 = low half of P[1], v2 = high half of KQ[0], v3 / v4 = KQ[1] .. v9 = low half of KQ[4].
				 * The first item of a row looks at the last pair (509, 510) of the row above instead of columns -1, 0. */
				const uint32_t *pp = g ? kr + 4 * g - 1 : kr - FI_RD + 254;
				uint32_t P[2] = { pp[0], pp[1] }, KQ[5];
#pragma unroll
				for (int j = 0; j < 5; j++) KQ[j] = kr[4 * g + j];
				const s16x2 lim = (s16x2)(short)202;
				auto clampw = [&](uint32_t w) { return as_w((s16x2)__builtin_elementwise_min(__builtin_elementwise_max(as_s(w), -lim), lim)); };
				P[0] = clampw(P[0]); P[1] = clampw(P[1]);
#pragma unroll
				for (int j = 0; j < 5; j++) KQ[j] = clampw(KQ[j]);
				auto lo = [](uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); };
				auto hi = [](uint32_t w) { return (int)w >> 16; };
				/* byte offset of a pair's entry: 180 x class of the first + 12 x class of the second */
				int prev = tcb[202 + lo(P[1])] + 180 * tca[202 + hi(P[0])];
				int flag = (int)*reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(ptab) + prev + 8);
				if (g == 0 && rr == 1) flag = misc[2 + ((b + 1) & 1)];     /* the band before left it (0 at the top of the image) */
				uint32_t dl[4];
				int flag2 = 0;
#pragma unroll
				for (int e = 0; e < 4; e++) {
					const int off = tcb[202 + lo(KQ[e + 1])] + 180 * tca[202 + hi(KQ[e])];
					const uint8_t *en = reinterpret_cast<const uint8_t *>(ptab) + off;
					dl[e] = *reinterpret_cast<const uint32_t *>(en + 4 * flag);
					flag = (int)*reinterpret_cast<const uint32_t *>(en + 8);
					if (e == 2) flag2 = flag;
				}
				if (k == 0) misc[8] = 0;
				if (g == 63) {                                           /* the last item's fourth pair would be (511, 512): no such pair; its third, (509, 510), hands the flag to the next row */
					dl[3] = 0;
					if (r0 + rr == last_row) misc[2 + (b & 1)] = (uint8_t)flag2;
				}
				/* luma columns 8g+1 .. 8g+8: the high half of dword 4g, dwords 4g+1 .. 4g+3, the low half of dword 4g+4 */
				uint32_t *yo = reinterpret_cast<uint32_t *>(ybuf) + rr * FI_YRD + 4 * g + (g >> 3);
				int16_t *ys = reinterpret_cast<int16_t *>(yo), *ye = reinterpret_cast<int16_t *>(yo + ((g & 7) == 7 ? 5 : 4));   /* dword 4g+4: behind a pad for the last group of a block */
				ys[1] = (int16_t)(ys[1] + (int16_t)(dl[0] & 0xFFFFu));
#pragma unroll
				for (int j = 1; j < 4; j++) yo[j] = pk_add16(yo[j], pack_hl(dl[j - 1], dl[j]));
				ye[0] = (int16_t)(ye[0] + (int16_t)(dl[3] >> 16));
			}
			}
			__syncthreads();
			FI_TICK(6); FI_STAMP(6);
			FI_DUMP(1, ybuf + FI_YRS, 1);
		}
		if (!PRE && SRC) __syncthreads();                              /* the staging rows become rows of the horizontal pass */
		/* ---------------------------------------------------------------- horizontal pass (filters.c:346-386) of image rows 32b+1 .. 32b+32 (and row 0), four kx an item,
		 * two outputs to a dword: L(k, k+1) = 6 E(k) + 2 (O(k-1) + O(k)) - (E(k-1) + E(k+1)), H(k, k+1) = 2 O(k) - (E(k) + E(k+1)) with
		 * E(j) = (x[2j], x[2j+2]), O(j) = (x[2j+1], x[2j+3]) put together from the row's dwords (x[2j], x[2j+1]). */
		{
			const int t = opaque(t0);
			const int top = r0 + FI_BR < W ? FI_BR : W - 1 - r0;       /* luma index of the band's last image row */
			for (int k = t; k < (b == 0 ? 33 : 32) * 64; k += FI_NT) {
				int rr, g;
				if (b == 0) { rr = k % 33; g = k / 33; } else { rr = 1 + (k & 31); g = k >> 5; if (rr > top) continue; }
				const uint32_t *yr = reinterpret_cast<const uint32_t *>(ybuf) + rr * FI_YRD + 4 * g + (g >> 3);
				uint32_t X[6];                                           /* X[j] = (x[8g - 2 + 2j], x[8g - 1 + 2j]): dwords 4g-1 .. 4g+4 of the row */
				X[0] = yr[(g & 7) == 0 ? -2 : -1]; X[5] = yr[(g & 7) == 7 ? 5 : 4];
#pragma unroll
				for (int j = 1; j < 5; j++) X[j] = yr[j - 1];
				if (g == 0) X[0] = __builtin_amdgcn_perm(X[1], X[2], 0x07060100u);   /* x[-2] = x[2], x[-1] = x[1] */
				if (g == 63) X[5] = X[4];                               /* x[512] = x[510] */
				uint32_t E[5], O[4];
#pragma unroll
				for (int j = 0; j < 5; j++) E[j] = pack_lo(X[j], X[j + 1]);
#pragma unroll
				for (int j = 0; j < 4; j++) O[j] = pack_hi(X[j], X[j + 1]);
				uint32_t L[2], Hh[2];
#pragma unroll
				for (int p = 0; p < 2; p++) {
					const s16x2 e0 = as_s(E[2 * p]), e1 = as_s(E[2 * p + 1]), e2 = as_s(E[2 * p + 2]), o0 = as_s(O[2 * p]), o1 = as_s(O[2 * p + 1]);
					L[p] = as_w((s16x2)(e1 * (s16x2)(short)6 + ((o0 + o1) << 1) - (e0 + e2)));
					Hh[p] = as_w((s16x2)((o1 << 1) - (e1 + e2)));
				}
				uint32_t *dlo = reinterpret_cast<uint32_t *>(kbuf + (rr + 4) * FI_RS + 4 * g), *dhi = reinterpret_cast<uint32_t *>(kbuf + (rr + 4) * FI_RS + H + 4 * g);
				dlo[0] = L[0]; dlo[1] = L[1]; dhi[0] = Hh[0]; dhi[1] = Hh[1];
			}
		}
		__syncthreads();
		FI_TICK(7); FI_STAMP(7);
		/* ---------------------------------------------------------------- vertical pass + stores */
		const int t = opaque(t0);
		if (keepb && !(flags & 2)) {                                   /* q >= 22: transposed horizontal-pass plane, rows kx < 256 (wavelet_filterbank.c:107-112): image rows 32b .. 32b+31 */
			int16_t *keep = keepb + (size_t)img * keep_stride;
			for (int k = t; k < H * 4; k += FI_NT) {
				const int kx = k >> 2, part = k & 3;
				uint32_t v[4];
#pragma unroll
				for (int e = 0; e < 4; e++) {
					const int ri = 4 + 8 * part + 2 * e;
					v[e] = (uint16_t)kbuf[ri * FI_RS + kx] | ((uint32_t)(uint16_t)kbuf[(ri + 1) * FI_RS + kx] << 16);
				}
				*reinterpret_cast<uint4 *>(keep + (size_t)kx * W + r0 + 8 * part) = make_uint4(v[0], v[1], v[2], v[3]);
			}
		}
		FI_TICK(10);
		uint32_t hold[3];                                              /* horizontal-pass rows 32b+28 .. 32b+32: the next band's first five */
#pragma unroll
		for (int e = 0; e < 3; e++) { const int k = t + FI_NT * e; if (k < 5 * FI_RD) hold[e] = reinterpret_cast<const uint32_t *>(kbuf + 32 * FI_RS)[k]; }
		/* luma rows 32b+32 (before the pair rules) and 32b+33 become the next band's rows 0 and 1 */
		for (int k = t; k < 2 * FI_YRD; k += FI_NT) {
			uint32_t *y = reinterpret_cast<uint32_t *>(ybuf);
			y[k] = k < FI_YRD ? y[(PRE ? 34 : 32) * FI_YRD + k] : y[32 * FI_YRD + k];   /* k >= FI_YRD: row 33, dword k - FI_YRD */
		}
		{
			/* a thread takes a PAIR of columns (2cp, 2cp+1) -- the halves of one dword -- and eight of the band's output rows.  Columns below 256
			 * (the low band of the horizontal pass) and the others take different rounding rules; a wavefront lies wholly on one side. */
			const int cp = t >> 1, kb = 8 * (t & 1);                    /* neighbouring lanes: the two halves of a column's 32 bytes */
			auto vertical = [&](auto side) {
				constexpr bool LEFT = decltype(side)::value;
				uint32_t col[21];                                       /* col[i] = horizontal-pass row 32b - 4 + 2kb + i, symmetric extension x[-j] = x[j], x[511+j] = x[511-j] */
#pragma unroll
				for (int i = 0; i < 21; i++) {
					int ri = 2 * kb + i;
					if (b == 0 && ri < 4) ri = 8 - ri;
					if (b == W / FI_BR - 1 && ri == 36) ri = 34;
					col[i] = reinterpret_cast<const uint32_t *>(kbuf + ri * FI_RS)[cp];
				}
				FI_TICK(11);
				uint32_t lo[8], hi[8];
				s16x2 rprev = (s16x2)(short)0;
				if (LEFT) rprev = as_s(col[2]) * (s16x2)(short)6 + ((as_s(col[1]) + as_s(col[3])) << 1) - (as_s(col[0]) + as_s(col[4]));
#pragma unroll
				for (int kk = 0; kk < 8; kk++) {
#define XS(d) as_s(col[2 * kk + 4 + (d)])                            /* x[2ky + d], -4 <= d <= 2 */
					const s16x2 r = XS(0) * (s16x2)(short)6 + ((XS(-1) + XS(1)) << 1) - (XS(-2) + XS(2));
					s16x2 l, h;
					if (LEFT) {                                        /* filters.c:203-287 */
						s16x2 carry = pk_diffuse(rprev);
						if (kk == 0 && b == 0 && kb == 0) carry = (s16x2)(short)0;
						l = pk_rnd_half_away(r + carry, 6);
						rprev = r;
					} else l = pk_rnd_half_away(r, 4);                 /* filters.c:88-113 */
					s16x2 a = XS(0) + XS(2);
					if (kk & 1) a = a + (a & (XS(-2) + XS(0)) & (s16x2)(short)1);   /* odd ky: both neighbouring sums odd -> the mean is taken one up */
					const s16x2 pr = XS(1) - (a >> 1);
					h = pk_rnd_half_away(pr, LEFT ? 3 : 1);            /* right half: pr > 0 ? (pr + 1) >> 1 : pr >> 1, the same thing */
					if (kk == 7 && b == W / FI_BR - 1 && kb == 8) { const s16x2 dd = XS(1) - XS(0); h = LEFT ? dd >> 3 : (dd + (s16x2)(short)1) >> 1; }   /* ky = 255 */
#undef XS
					lo[kk] = as_w(l); hi[kk] = as_w(h);
				}
				FI_TICK(12);
				/* the level-1 plane is kept transposed: column c is row c of it, eight output rows = 16 bytes */
				int16_t *orow = proc + (size_t)(2 * cp) * W + 16 * b + kb;
#ifdef NHW_DEV
				if (!(flags & 4))
#endif
				{
				auto st16 = [&](int16_t *p, uint32_t a, uint32_t b_, uint32_t c, uint32_t d) { *reinterpret_cast<uint4 *>(p) = make_uint4(a, b_, c, d); };
				if (!LEFT || (flags & 0x100000)) {                       /* the LL quadrant of the transposed plane: the level-2 analysis writes all of it, nothing reads it before (stored for the stage checks only) */
					st16(orow, pack_lo(lo[0], lo[1]), pack_lo(lo[2], lo[3]), pack_lo(lo[4], lo[5]), pack_lo(lo[6], lo[7]));
					st16(orow + W, pack_hi(lo[0], lo[1]), pack_hi(lo[2], lo[3]), pack_hi(lo[4], lo[5]), pack_hi(lo[6], lo[7]));
				}
				st16(orow + H, pack_lo(hi[0], hi[1]), pack_lo(hi[2], hi[3]), pack_lo(hi[4], hi[5]), pack_lo(hi[6], hi[7]));
				st16(orow + W + H, pack_hi(hi[0], hi[1]), pack_hi(hi[2], hi[3]), pack_hi(hi[4], hi[5]), pack_hi(hi[6], hi[7]));
				}
#ifdef NHW_DEV
				if (!(flags & 8))
#endif
				if (LEFT) {                                            /* LL in natural orientation: jpeg[ky][kx] and ll1[ky][kx], kx < 256 (wavelet_filterbank.c:172-184, nhw_encoder.c:127-135) */
					uint32_t *jp = reinterpret_cast<uint32_t *>(jpeg + (size_t)(16 * b + kb) * W) + cp, *lp = reinterpret_cast<uint32_t *>(ll1 + (size_t)(16 * b + kb) * H) + cp;
#pragma unroll
					for (int kk = 0; kk < 8; kk++) { if (flags & 0x100000) jp[kk * (W / 2)] = lo[kk]; lp[kk * (H / 2)] = lo[kk]; }   /* (the copy in the work plane: stage checks only -- the level-2 analysis reads ll1) */
				}
			};
			if (__builtin_amdgcn_readfirstlane(cp) < H / 2) vertical(std::true_type{}); else vertical(std::false_type{});
		}
		FI_TICK(13);
		__syncthreads();
		FI_TICK(8);
#pragma unroll
		for (int e = 0; e < 3; e++) { const int k = t + FI_NT * e; if (k < 5 * FI_RD) reinterpret_cast<uint32_t *>(kbuf)[k] = hold[e]; }
	}
#ifdef NHW_DEV
	if ((flags & 0x10000) && t0 == 0) for (int i = 0; i < 14; i++) atomicAdd(&g_fi_prof[i], (unsigned long long)acc_[i]);
#endif
}


/* ------------------------------------------------------------------------------------------------
 * The front WITHOUT the pre-filter of quality 17..21: quality 22 / 23 from the BGR bytes (SRC 1), quality 1..16 and the analysis stage from a
 * luma plane (SRC 0).  Nothing needs a row of luma twice then: the horizontal pass runs in phase 0 straight from the registers the colour
 * conversion left the row in (the two neighbours' edge pixels come over the lanes), so there is no luma buffer and no horizontal phase, and
 * the LDS that frees holds 64 image rows of the horizontal pass instead of 32: the vertical pass makes 32 output rows a time, a column's
 * run in the transposed level-1 plane is 64 bytes (two neighbouring lanes) instead of 32, the rows kept for quality >= 22 go out as whole
 * 128-byte lines, and an image costs 34 barriers instead of 150.
 * A step takes 32 image rows, 32s+1 .. 32s+32 (row 0 comes with a step of its own before the first: s = -1), a thread two neighbouring rows
 * of a group of 16 pixels.  4:2:0: chroma row m needs image rows 2m-1, 2m, 2m+1 = a thread's two rows and the first row of the thread below;
 * the first rows go through LDS, the last thread's pair waits there for the next step.
 * ------------------------------------------------------------------------------------------------ */
#define FP_RS     512                 /* LDS row stride in shorts: rows are written 16 bytes a lane, columns read a dword a lane -- no padding needed */
#define FP_HROWS  69                  /* horizontal-pass rows 64B-4 .. 64B+64 (index = row - 64B + 4) */
#define FP_STG_OFF (FP_HROWS * FP_RS * 2)
#define FP_LDS_BYTES (FP_STG_OFF + 19 * 512)   /* filtered chroma of the threads' first rows (slots 0..14: threads 1..15), two pairs of rows of the last thread (slots 15..18) */
static_assert(FP_LDS_BYTES <= 81920, "two workgroups to a CU");

/* the horizontal pass (filters.c:346-386) of 16 pixels: X[0] = (x[-2], x[-1]), X[1..8] = the row's dwords, X[9] = (x[16], .) -> four dwords of the low band, four of the high */
__device__ __forceinline__ void hp16(const uint32_t X[10], uint32_t L[4], uint32_t Hh[4])
{
	uint32_t E[9], O[8];
#pragma unroll
	for (int j = 0; j < 9; j++) E[j] = pack_lo(X[j], X[j + 1]);     /* (x[2j-2], x[2j]) */
#pragma unroll
	for (int j = 0; j < 8; j++) O[j] = pack_hi(X[j], X[j + 1]);     /* (x[2j-1], x[2j+1]) */
#pragma unroll
	for (int p = 0; p < 4; p++) {                                    /* outputs 2p, 2p+1 */
		const s16x2 e0 = as_s(E[2 * p]), e1 = as_s(E[2 * p + 1]), e2 = as_s(E[2 * p + 2]), o0 = as_s(O[2 * p]), o1 = as_s(O[2 * p + 1]);
		L[p] = as_w((s16x2)(e1 * (s16x2)(short)6 + ((o0 + o1) << 1) - (e0 + e2)));
		Hh[p] = as_w((s16x2)((o1 << 1) - (e1 + e2)));
	}
}

template <int SRC, int FAMILY>
__global__ __launch_bounds__(FI_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_front_plain(const void *__restrict__ srcb, size_t src_stride, float yq, uint8_t *__restrict__ pub, uint8_t *__restrict__ pvb, size_t c_stride,
                                                     int16_t *__restrict__ procb, int16_t *__restrict__ jpegb, size_t plane_stride,
                                                     int16_t *__restrict__ ll1b, size_t ll1_stride, int16_t *__restrict__ keepb, size_t keep_stride, int flags)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	int16_t *const hbuf = reinterpret_cast<int16_t *>(lds);
	uint8_t *const stg = lds + FP_STG_OFF;
	const int t0 = threadIdx.x, img = blockIdx.x;
	const uint8_t *const src = (const uint8_t *)srcb + (size_t)img * (SRC ? (size_t)(W * W * 3) : src_stride);
	int16_t *const proc = procb + (size_t)img * plane_stride, *const jpeg = jpegb + (size_t)img * plane_stride;
	int16_t *const ll1 = ll1b + (size_t)img * ll1_stride;

	constexpr int NPF = SRC ? 6 : 4;
	uint4 pf[NPF];
	auto issue = [&](int s) {                                       /* image rows 32s+1+2j, 32s+2+2j of the thread's group of 16 pixels */
		const int t = opaque(t0), g = t & 31, j = t >> 5;
		const uint4 *rp[2];                                            /* (rows outside the image: the nearest row inside, not used -- see k_front_image) */
#pragma unroll
		for (int h = 0; h < 2; h++) {
			int row = 32 * s + 1 + 2 * j + h;
			row = row < 0 ? 0 : row > W - 1 ? W - 1 : row;
			rp[h] = SRC ? reinterpret_cast<const uint4 *>(src + (size_t)row * (W * 3) + 48 * g) : reinterpret_cast<const uint4 *>(reinterpret_cast<const int16_t *>(src) + (size_t)row * W + 16 * g);
		}
#pragma unroll
		for (int h = 0; h < 2; h++) {
			if (SRC) { pf[3 * h] = rp[h][0]; pf[3 * h + 1] = rp[h][1]; pf[3 * h + 2] = rp[h][2]; }
			else { pf[2 * h] = rp[h][0]; pf[2 * h + 1] = rp[h][1]; }
		}
	};
	issue(-1);
#pragma unroll 1
	for (int s = -1; s < W / 32; s++) {
		const int B = s < 0 ? 0 : s >> 1;                            /* the 64-row band the step belongs to */
		uint2 cu[2], cv[2];                                          /* the thread's two rows of horizontally filtered chroma */
		{
			const int t = opaque(t0), g = t & 31, j = t >> 5;
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int row = 32 * s + 1 + 2 * j + h;
				const bool live = row >= 0 && row < W;
				uint32_t X[10], uw[4] = { 0, 0, 0, 0 }, vw[4] = { 0, 0, 0, 0 };
				if (live) {
					if (SRC) {
						const uint32_t wv[12] = { pf[3 * h].x, pf[3 * h].y, pf[3 * h].z, pf[3 * h].w, pf[3 * h + 1].x, pf[3 * h + 1].y, pf[3 * h + 1].z, pf[3 * h + 1].w,
						                          pf[3 * h + 2].x, pf[3 * h + 2].y, pf[3 * h + 2].z, pf[3 * h + 2].w };
						convert16<FAMILY>(wv, yq, X + 1, uw, vw, true);
					} else {
						X[1] = pf[2 * h].x; X[2] = pf[2 * h].y; X[3] = pf[2 * h].z; X[4] = pf[2 * h].w; X[5] = pf[2 * h + 1].x; X[6] = pf[2 * h + 1].y; X[7] = pf[2 * h + 1].z; X[8] = pf[2 * h + 1].w;
					}
				} else {
#pragma unroll
					for (int e = 1; e < 9; e++) X[e] = 0;
				}
				/* the neighbours' edge pixels come over the lanes (every lane takes part); the ends of the row are mirrored: x[-2] = x[2], x[-1] = x[1], x[512] = x[510] */
				X[0] = (uint32_t)__shfl_up((int)X[8], 1);
				X[9] = (uint32_t)__shfl_down((int)X[1], 1);
				if (g == 0) X[0] = __builtin_amdgcn_perm(X[1], X[2], 0x07060100u);
				if (g == 31) X[9] = X[8];
				if (SRC) {
					uint32_t lu = (uint32_t)__shfl_up((int)uw[3], 1), lv = (uint32_t)__shfl_up((int)vw[3], 1);
					if (g == 0) { lu = uw[0] << 16; lv = vw[0] << 16; }
					cu[h] = chroma_h8(uw, lu); cv[h] = chroma_h8(vw, lv);
				}
				if (live) {
					uint32_t L[4], Hh[4];
					hp16(X, L, Hh);
					int16_t *hr = hbuf + (row - 64 * B + 4) * FP_RS + 8 * g;
					*reinterpret_cast<uint4 *>(hr) = make_uint4(L[0], L[1], L[2], L[3]);
					*reinterpret_cast<uint4 *>(hr + H) = make_uint4(Hh[0], Hh[1], Hh[2], Hh[3]);
				}
			}
			if (SRC) {
				/* slot of a first row: thread j - 1 finds it; the last thread leaves both its rows for the next step (two pairs, taken in turns) */
				if (j) { *reinterpret_cast<uint2 *>(stg + (j - 1) * 512 + 8 * g) = cu[0]; *reinterpret_cast<uint2 *>(stg + (j - 1) * 512 + 256 + 8 * g) = cv[0]; }
				if (j == 15) {
					uint8_t *kp = stg + (15 + 2 * (s & 1)) * 512;
					*reinterpret_cast<uint2 *>(kp + 8 * g) = cu[0]; *reinterpret_cast<uint2 *>(kp + 256 + 8 * g) = cv[0];
					*reinterpret_cast<uint2 *>(kp + 512 + 8 * g) = cu[1]; *reinterpret_cast<uint2 *>(kp + 768 + 8 * g) = cv[1];
				}
			}
		}
		if (SRC || (s & 1)) __syncthreads();
		if (s + 1 < W / 32) issue(s + 1);
		if (SRC && s >= 0) {
			/* 4:2:0 (colorspace.c:241-256): chroma row m = 16s+1+j from the thread's rows 2m-1, 2m and the first row of the thread below; the first
			 * thread also makes row 16s, which waited for its third row (row 0: (r0 + r1 + 1) >> 1 of image rows 0 and 1) */
			const int t = opaque(t0), g = t & 31, j = t >> 5, m = 16 * s + 1 + j;
			uint8_t *pu = pub + (size_t)img * c_stride, *pv = pvb + (size_t)img * c_stride;
			if (j < 15 && m < H) {
				const uint2 nu = *reinterpret_cast<const uint2 *>(stg + j * 512 + 8 * g), nv = *reinterpret_cast<const uint2 *>(stg + j * 512 + 256 + 8 * g);
				*reinterpret_cast<uint2 *>(pu + m * H + 8 * g) = make_uint2(tri121(cu[0].x, cu[1].x, nu.x), tri121(cu[0].y, cu[1].y, nu.y));
				*reinterpret_cast<uint2 *>(pv + m * H + 8 * g) = make_uint2(tri121(cv[0].x, cv[1].x, nv.x), tri121(cv[0].y, cv[1].y, nv.y));
			}
			if (j == 0) {
				const uint8_t *kp = stg + (15 + 2 * ((s + 1) & 1)) * 512;   /* what the step before left */
				const uint2 au = *reinterpret_cast<const uint2 *>(kp + 8 * g), av = *reinterpret_cast<const uint2 *>(kp + 256 + 8 * g);
				const uint2 bu = *reinterpret_cast<const uint2 *>(kp + 512 + 8 * g), bv = *reinterpret_cast<const uint2 *>(kp + 768 + 8 * g);
				uint2 ou, ov;
				if (s == 0) { ou = make_uint2(avg_up(bu.x, cu[0].x), avg_up(bu.y, cu[0].y)); ov = make_uint2(avg_up(bv.x, cv[0].x), avg_up(bv.y, cv[0].y)); }
				else { ou = make_uint2(tri121(au.x, bu.x, cu[0].x), tri121(au.y, bu.y, cu[0].y)); ov = make_uint2(tri121(av.x, bv.x, cv[0].x), tri121(av.y, bv.y, cv[0].y)); }
				*reinterpret_cast<uint2 *>(pu + 16 * s * H + 8 * g) = ou;
				*reinterpret_cast<uint2 *>(pv + 16 * s * H + 8 * g) = ov;
			}
		}
		if (s > 0 && (s & 1)) {
			/* ------------------------------------------------------------ vertical pass of the band's 32 output rows: a thread takes a pair of columns and 16 of them */
			const int t = opaque(t0);
			if (keepb) {                                               /* q >= 22: transposed horizontal-pass plane, rows kx < 256 (wavelet_filterbank.c:107-112): image rows 64B .. 64B+63, whole lines */
				int16_t *keep = keepb + (size_t)img * keep_stride;
				const int kx = t >> 1, part = t & 1;
				uint32_t v[16];
#pragma unroll
				for (int e = 0; e < 16; e++) {
					const int ri = 4 + 32 * part + 2 * e;
					v[e] = (uint16_t)hbuf[ri * FP_RS + kx] | ((uint32_t)(uint16_t)hbuf[(ri + 1) * FP_RS + kx] << 16);
				}
				uint4 *kd = reinterpret_cast<uint4 *>(keep + (size_t)kx * W + 64 * B + 32 * part);
#pragma unroll
				for (int e = 0; e < 4; e++) kd[e] = make_uint4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
			}
			uint32_t hold[3];                                          /* horizontal-pass rows 64B+60 .. 64B+64: the next band's first five */
#pragma unroll
			for (int e = 0; e < 3; e++) { const int k = t + FI_NT * e; if (k < 5 * (FP_RS / 2)) hold[e] = reinterpret_cast<const uint32_t *>(hbuf + 64 * FP_RS)[k]; }
			const int cp = t >> 1, kb = 16 * (t & 1);
			auto vertical = [&](auto side) {
				constexpr bool LEFT = decltype(side)::value;
				int16_t *orow = proc + (size_t)(2 * cp) * W + 32 * B + kb;
				uint32_t *jp = reinterpret_cast<uint32_t *>(jpeg + (size_t)(32 * B + kb) * W) + cp, *lp = reinterpret_cast<uint32_t *>(ll1 + (size_t)(32 * B + kb) * H) + cp;
				s16x2 rprev = (s16x2)(short)0;
#pragma unroll
				for (int part = 0; part < 2; part++) {                   /* eight output rows at a time */
					uint32_t col[21];                                   /* col[i] = horizontal-pass row 64B - 4 + 2 (kb + 8 part) + i, symmetric extension at both ends of the image */
#pragma unroll
					for (int i = 0; i < 21; i++) {
						int ri = 2 * (kb + 8 * part) + i;
						if (B == 0 && ri < 4) ri = 8 - ri;
						if (B == W / 64 - 1 && ri == 68) ri = 66;
						col[i] = reinterpret_cast<const uint32_t *>(hbuf + ri * FP_RS)[cp];
					}
					uint32_t lo[8], hi[8];
					if (LEFT && part == 0) rprev = as_s(col[2]) * (s16x2)(short)6 + ((as_s(col[1]) + as_s(col[3])) << 1) - (as_s(col[0]) + as_s(col[4]));
#pragma unroll
					for (int kk = 0; kk < 8; kk++) {
#define XS(d) as_s(col[2 * kk + 4 + (d)])
						const s16x2 r = XS(0) * (s16x2)(short)6 + ((XS(-1) + XS(1)) << 1) - (XS(-2) + XS(2));
						s16x2 l, h;
						if (LEFT) {                                    /* filters.c:203-287 */
							s16x2 carry = pk_diffuse(rprev);
							if (part == 0 && kk == 0 && B == 0 && kb == 0) carry = (s16x2)(short)0;
							l = pk_rnd_half_away(r + carry, 6);
							rprev = r;
						} else l = pk_rnd_half_away(r, 4);             /* filters.c:88-113 */
						s16x2 a = XS(0) + XS(2);
						if (kk & 1) a = a + (a & (XS(-2) + XS(0)) & (s16x2)(short)1);
						const s16x2 pr = XS(1) - (a >> 1);
						h = pk_rnd_half_away(pr, LEFT ? 3 : 1);
						if (part == 1 && kk == 7 && B == W / 64 - 1 && kb == 16) { const s16x2 dd = XS(1) - XS(0); h = LEFT ? dd >> 3 : (dd + (s16x2)(short)1) >> 1; }   /* ky = 255 */
#undef XS
						lo[kk] = as_w(l); hi[kk] = as_w(h);
					}
					auto st16 = [&](int16_t *p, uint32_t a, uint32_t b_, uint32_t c, uint32_t d) { *reinterpret_cast<uint4 *>(p) = make_uint4(a, b_, c, d); };
					int16_t *op = orow + 8 * part;
					if (!LEFT || (flags & 0x100000)) {                   /* the LL quadrant of the transposed plane: the level-2 analysis writes all of it, nothing reads it before (stored for the stage checks only) */
						st16(op, pack_lo(lo[0], lo[1]), pack_lo(lo[2], lo[3]), pack_lo(lo[4], lo[5]), pack_lo(lo[6], lo[7]));
						st16(op + W, pack_hi(lo[0], lo[1]), pack_hi(lo[2], lo[3]), pack_hi(lo[4], lo[5]), pack_hi(lo[6], lo[7]));
					}
					st16(op + H, pack_lo(hi[0], hi[1]), pack_lo(hi[2], hi[3]), pack_lo(hi[4], hi[5]), pack_lo(hi[6], hi[7]));
					st16(op + W + H, pack_hi(hi[0], hi[1]), pack_hi(hi[2], hi[3]), pack_hi(hi[4], hi[5]), pack_hi(hi[6], hi[7]));
					if (LEFT) {
#pragma unroll
						for (int kk = 0; kk < 8; kk++) { if (flags & 0x100000) jp[(8 * part + kk) * (W / 2)] = lo[kk]; lp[(8 * part + kk) * (H / 2)] = lo[kk]; }
					}
				}
			};
			if (__builtin_amdgcn_readfirstlane(cp) < H / 2) vertical(std::true_type{}); else vertical(std::false_type{});
			__syncthreads();
#pragma unroll
			for (int e = 0; e < 3; e++) { const int k = t + FI_NT * e; if (k < 5 * (FP_RS / 2)) reinterpret_cast<uint32_t *>(hbuf)[k] = hold[e]; }
		} else if (SRC) __syncthreads();                               /* the chroma slots are free for the next step */
	}
}

} // namespace nhw
#endif
