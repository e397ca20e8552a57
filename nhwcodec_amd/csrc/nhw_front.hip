/*
 * nhw_front.hip -- data-parallel front of the NHW encode path for gfx950 (MI355X):
 * colour conversion + 4:2:0, luma pre-filter, separable integer 5/3 filterbank (analysis + synthesis).
 *
 * Algorithm sources (behaviour only; nothing here is derived from the reference's code structure):
 *   colour        rcanut/nhwcodec encoder/colorspace.c:55-260
 *   pre-filter    encoder/image_processing.c:558-837, 1927-1990 (quality 17..21 branch)
 *   filterbank    encoder/wavelet_filterbank.c:52-496, encoder/filters.c:55-114, 203-287, 346-386, 521-572
 *
 * No MFMA: there is no dense contraction on this path; every kernel is integer/byte streaming work
 * bounded by HBM.  Built with -ffp-contract=off: the luma weights are double products summed left to
 * right and a fused multiply-add changes 300+ of the 2^24 colour triples.
 */
#include <type_traits>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include "nhw_ws.h"
#include "nhw_dwt.h"

#pragma clang fp contract(off)

namespace nhw {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
/* |a - b| + acc on unsigned operands in one instruction (the compiler does not form it from the C expression) */
__device__ __forceinline__ unsigned sad_u32(unsigned a, unsigned b, unsigned acc)
{
	unsigned d;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
	return d;
}
/* |a.lo - b.lo| + |a.hi - b.hi| + acc on the two unsigned 16-bit halves of a dword */
__device__ __forceinline__ unsigned sad_u16(unsigned a, unsigned b, unsigned acc)
{
	unsigned d;
	asm("v_sad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
	return d;
}
/* two 16-bit sums in a dword (no carry between the halves) */
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b)
{
	unsigned d;
	asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
	return d;
}

/* ------------------------------------------------------------------------------------------------
 * colour + 4:2:0.  One workgroup per pair of luma rows (2r, 2r+1) = one chroma row r.
 * The three BGR rows 2r-1..2r+1 are staged in LDS with coalesced 16-byte loads.
 * ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ uint8_t clip_u8(int v) { return (v >> 8) != 0 ? (v < 0 ? 0 : 255) : (uint8_t)v; }
__device__ __forceinline__ int chroma_round(float cb) { return cb >= 0 ? (int)(cb + 128.5f) : (int)(cb + 128.4f); }

/* For q >= 20 the conversion is exact integer arithmetic (checked against the double/float form on all 2^24 triples):
 *   chroma: cb = (-1687 b0 - 3313 b1 + 5000 b2) / 10000 lies at least 1e-4 away from every rounding boundary the float
 *           form can hit, so (int)(cb + 128.5f) (cb >= 0) / (int)(cb + 128.4f) (cb < 0) are floors of the exact value;
 *   luma:   (int)(0.299 b0 + 0.587 b1 + 0.114 b2 + 0.5f) is floor((299 b0 + 587 b1 + 114 b2 + 500) / 1000) except when that
 *           division is exact (one triple in a thousand): there the double rounding of the three products decides, and
 *           the lane takes the double path. */
/* the full-rate 24 x 24 bit multiplier: low 32 bits and bits 32..47 of the product (operands must fit 24 bits) */
__device__ __forceinline__ unsigned mulhi_u24(unsigned a, unsigned b) { unsigned r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned mul_u24(unsigned a, unsigned b) { unsigned r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int FAMILY> /* 0: q>=20, 1: q 18/19, 2: q17, 3: q<=16 (integer BT.601 scaled by the quality table; yq carries the table entry's bits) */
__device__ __forceinline__ void convert_uv(const uint8_t *px, float yq, int &U, int &V)
{
	const int b0 = px[0], b1 = px[1], b2 = px[2];
	if (FAMILY == 3) {                                             /* colorspace.c:172-214 */
		const int qz = __float_as_int(yq);
		U = clip_u8((((-38 * b0 - 74 * b1 + 112 * b2) * qz + 4194304) >> 23) + 128);
		V = clip_u8((((112 * b0 - 94 * b1 - 18 * b2) * qz + 4194304) >> 23) + 128);
		return;
	}
	const int su = -1687 * b0 - 3313 * b1 + 5000 * b2, sv = 5000 * b0 - 4187 * b1 - 813 * b2;
	if (FAMILY == 0 || FAMILY == 1) {                              /* q 18/19 scale the luma only: the chroma is that of q >= 20 */
		/* the biased sums are 9000 .. 2560000 (22 bits): x / 10000 = (x * 13743896) >> 37 exactly over that range, and both factors fit the
		 * full-rate 24-bit multiplier (a 32-bit v_mul_hi runs at a quarter of the rate); the quotient is 0 .. 256, so one min() clips it */
		U = (int)min(mulhi_u24((unsigned)(su + (su >= 0 ? 1285000 : 1284000)), 13743896u) >> 5, 255u);
		V = (int)min(mulhi_u24((unsigned)(sv + (sv >= 0 ? 1285000 : 1284000)), 13743896u) >> 5, 255u);
		return;
	}
	/* q17: 0.94 x the same sums, rounded through a float.  Exactly, the value is N / 10^6 with N = 94 s + 128.5e6 (128.4e6 below zero); the
	 * float form can only differ from floor(N / 10^6) when N sits within 1e-4 of a multiple of 10^6 (2.5e-4 of the triples): those lanes
	 * take the reference's own arithmetic.  N / 10^6 = ((N >> 6) * 8796094) >> 37 for N < 2^28 (floor(floor(N / 64) / 15625)). */
	const unsigned nu = (unsigned)(94 * su + (su >= 0 ? 128500000 : 128400000)), nv = (unsigned)(94 * sv + (sv >= 0 ? 128500000 : 128400000));
	const unsigned qu = mulhi_u24(nu >> 6, 8796094u) >> 5, qv = mulhi_u24(nv >> 6, 8796094u) >> 5;
	const unsigned ru = nu - mul_u24(qu, 1000000u), rv = nv - mul_u24(qv, 1000000u);
	U = (int)min(qu, 255u); V = (int)min(qv, 255u);
	if (ru - 100u > 999800u) U = clip_u8(chroma_round((float)((-0.1687 * b0 - 0.3313 * b1 + 0.5 * b2) * 0.94)));
	if (rv - 100u > 999800u) V = clip_u8(chroma_round((float)((0.5 * b0 - 0.4187 * b1 - 0.0813 * b2) * 0.94)));
}
template <int FAMILY>
__device__ __forceinline__ int convert_y(const uint8_t *px, float yq)
{
	const int b0 = px[0], b1 = px[1], b2 = px[2];
	if (FAMILY == 3) return (((66 * b0 + 129 * b1 + 25 * b2) * __float_as_int(yq) + 4194304) >> 23) + 16;
	if (FAMILY == 0) {
		/* s <= 255500 (18 bits): s / 1000 = (s * 8589935) >> 33 exactly, again on the 24-bit multiplier */
		const unsigned s = (unsigned)(299 * b0 + 587 * b1 + 114 * b2 + 500), y = mulhi_u24(s, 8589935u) >> 1;
		if (s != mul_u24(y, 1000u)) return (int)y;
	}
	if (FAMILY == 1 || FAMILY == 2) {
		/* q 17..19: (int)(ly x scale + 0.5) in double.  The same product in single precision is off by less than 1e-4, so its floor is the
		 * answer unless it lands within 2.5e-4 of an integer (5e-4 of the triples): those lanes take the double path below.  Checked, like
		 * everything here, on all 2^24 triples. */
		const float c = (FAMILY == 1 ? yq : 0.94f) * 0.001f;
		const float v = (float)(299 * b0 + 587 * b1 + 114 * b2) * c + 0.5f, fl = floorf(v), fr = v - fl;
		if (fr > 2.5e-4f && fr < 1.f - 2.5e-4f) return (int)fl;
	}
	const double ly = 0.299 * b0 + 0.587 * b1 + 0.114 * b2;
	if (FAMILY == 0) return (int)(ly + 0.5f);
	if (FAMILY == 1) return (int)(ly * yq + 0.5f);
	return (int)(ly * 0.94 + 0.5f);
}

template <int FAMILY> __device__ __forceinline__ void convert16(const uint32_t wv[12], float yq, uint32_t yw[8], uint32_t uw[4], uint32_t vw[4], bool uv);   /* nhw_front_image.h */
/* One workgroup per 8 luma rows = 4 chroma rows: the 9 BGR rows 8b-1 .. 8b+7 are staged in LDS with 16-byte loads
 * (the row above is the only one read twice), every pixel is converted once (Y straight to HBM, U and V as bytes to
 * LDS), then the [1 2 1] x [1 2 1] chroma filter runs on the LDS bytes. */
template <int FAMILY>
__global__ __launch_bounds__(256) void k_color(const uint8_t *__restrict__ bgr, int16_t *__restrict__ yb, size_t y_stride,
                                               uint8_t *__restrict__ ub, uint8_t *__restrict__ vb, size_t c_stride, float yq)
{
	__shared__ __attribute__((aligned(16))) uint8_t rows[9][W * 3];
	__shared__ __attribute__((aligned(16))) uint8_t uu[9][W], vv[9][W];
	const int b = blockIdx.x, img = blockIdx.y, t = threadIdx.x;
	const uint8_t *src = bgr + (size_t)img * (W * W * 3);
	for (int k = t; k < 9 * (W * 3 / 16); k += 256) {
		const int which = k / (W * 3 / 16), o = k % (W * 3 / 16), row = 8 * b - 1 + which;
		if (row >= 0) reinterpret_cast<uint4 *>(rows[which])[o] = reinterpret_cast<const uint4 *>(src + (size_t)row * (W * 3))[o];
	}
	__syncthreads();
	int16_t *yplane = (int16_t *)((uint8_t *)yb + (size_t)img * y_stride);
	if (FAMILY != 3) {                                             /* quality 17..23: the arithmetic of the fused front kernel (nhw_front_image.h), 16 pixels per item */
		for (int k = t; k < 9 * (W / 16); k += 256) {
			const int which = k / (W / 16), g = k % (W / 16), row = 8 * b - 1 + which;
			if (row < 0) continue;
			const uint4 *rp = reinterpret_cast<const uint4 *>(&rows[which][48 * g]);
			const uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
			const uint32_t wv[12] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w };
			uint32_t yw[8], uw[4], vw[4];
			convert16<FAMILY == 3 ? 0 : FAMILY>(wv, yq, yw, uw, vw, true);
			*reinterpret_cast<uint4 *>(&uu[which][16 * g]) = make_uint4(uw[0], uw[1], uw[2], uw[3]);
			*reinterpret_cast<uint4 *>(&vv[which][16 * g]) = make_uint4(vw[0], vw[1], vw[2], vw[3]);
			if (which) {
				uint4 *yo = reinterpret_cast<uint4 *>(yplane + (size_t)row * W + 16 * g);
				yo[0] = make_uint4(yw[0], yw[1], yw[2], yw[3]); yo[1] = make_uint4(yw[4], yw[5], yw[6], yw[7]);
			}
		}
	} else
	for (int k = t; k < 9 * (W / 2); k += 256) {                   /* two pixels per item */
		const int which = k / (W / 2), px = 2 * (k % (W / 2)), row = 8 * b - 1 + which;
		if (row < 0) continue;
		int u0, v0, u1, v1;
		convert_uv<FAMILY>(&rows[which][3 * px], yq, u0, v0);
		convert_uv<FAMILY>(&rows[which][3 * px + 3], yq, u1, v1);
		*reinterpret_cast<uint16_t *>(&uu[which][px]) = (uint16_t)(u0 | (u1 << 8));
		*reinterpret_cast<uint16_t *>(&vv[which][px]) = (uint16_t)(v0 | (v1 << 8));
		if (which) {                                               /* luma is not clipped (colorspace.c:80) */
			const int y0 = convert_y<FAMILY>(&rows[which][3 * px], yq), y1 = convert_y<FAMILY>(&rows[which][3 * px + 3], yq);
			*reinterpret_cast<uint32_t *>(yplane + (size_t)row * W + px) = (uint32_t)(uint16_t)y0 | ((uint32_t)(uint16_t)y1 << 16);
		}
	}
	__syncthreads();
	/* chroma: horizontal [1 2 1]/4 at even pixel 2c (first column: (c0+c1+1)>>1), then vertical [1 2 1]/4 over rows
	 * 2r-1, 2r, 2r+1 (first row: (r0+r1+1)>>1) */
	for (int k = t; k < 4 * H; k += 256) {
		const int rr = k / H, c = k % H, r = 4 * b + rr;
		int hu[3], hv[3];
		for (int j = 0; j < 3; j++) {
			const uint8_t *pu = uu[2 * rr + j] + 2 * c, *pv = vv[2 * rr + j] + 2 * c;
			if (c == 0) { hu[j] = (pu[0] + pu[1] + 1) >> 1; hv[j] = (pv[0] + pv[1] + 1) >> 1; }
			else { hu[j] = (pu[-1] + 2 * pu[0] + pu[1] + 2) >> 2; hv[j] = (pv[-1] + 2 * pv[0] + pv[1] + 2) >> 2; }
		}
		int U, V;
		if (r == 0) { U = (hu[1] + hu[2] + 1) >> 1; V = (hv[1] + hv[2] + 1) >> 1; }
		else { U = (hu[0] + 2 * hu[1] + hu[2] + 2) >> 2; V = (hv[0] + 2 * hv[1] + hv[2] + 2) >> 2; }
		(ub + (size_t)img * c_stride)[r * H + c] = (uint8_t)U;
		(vb + (size_t)img * c_stride)[r * H + c] = (uint8_t)V;
	}
}

/* ------------------------------------------------------------------------------------------------
 * luma pre-filter.  Pass A is a 3x3 contrast measure pushed through a 4-bit error-diffusion carry that
 * runs in raster order over the whole interior.  The carry is a 16-state machine; it is parallelised
 * by composing, per row, the state transfer map (16 states tracked at once, one byte each, SWAR on two
 * 64-bit words), chaining the 510 row maps, then replaying every row from its now-known start state.
 * ------------------------------------------------------------------------------------------------ */

/* the pair rules of the pre-filter (image_processing.c:810-837, 1927-1990) on one pixel pair, as sign-normalised
 * predicate arithmetic (no divergent branches): returns the two luma deltas packed as (d0 & 0xFFFF) | (d1 << 16) */
__device__ __forceinline__ uint32_t prefilter_pair_delta(int k0, int k1, int prev_big)
{
	const int s0 = k0 < 0 ? -1 : 1, s1 = k1 < 0 ? -1 : 1, a0 = iabs(k0), a1 = iabs(k1);
	const bool same = (k0 < 0) == (k1 < 0);
	/* :810-837 -- |k| above 176 / 201 pulls the pixel by 1 / 2 against the sign; the second pixel is damped when the first one moved */
	const int t0 = (a0 > 176) + (a0 > 201), t1 = (a1 > 176) + (a1 > 201);
	int d0 = -s0 * t0;
	const int m1 = t1 == 2 ? (t0 == 0 ? 2 : (t0 == 2 ? (same ? 0 : 2) : 1)) : (t1 == 1 ? !(t0 == 2 && same) : 0);
	int d1 = -s1 * m1;
	/* :1927-1990 -- a moderate kernel value next to a large one pushes the pixel the other way */
	const bool mod0 = a0 > 10 && a0 < 32, mod1 = a1 > 10 && a1 < 32;
	const bool first = mod0 && a1 >= 23;
	const bool second = !first && mod1 && a0 >= 23;
	const int k1n = s0 * k1, k0n = s1 * k0;                       /* neighbour value seen from the sign of the moderate one */
	const int f0 = a0 < 16 ? 1 : (prev_big ? 1 : 2), f1 = (a0 < 16) && k1n > 0 && k1n < 32 && a0 > 11;
	const int g1 = a1 < 16 ? 1 : 2, g0 = (a1 < 16) && k0n > 0 && k0n < 32 && a1 > 11;
	d0 += first ? s0 * f0 : (second ? s1 * g0 : 0);
	d1 += first ? s0 * f1 : (second ? s1 * g1 : 0);
	return (uint32_t)(uint16_t)d0 | ((uint32_t)(uint16_t)d1 << 16);
}
/* hand-over flag to the next pair: set when the second pixel took the "+-2" branch above */
__device__ __forceinline__ int pair_big_flag_fwd(int k0, int k1)
{
	const int a0 = iabs(k0), a1 = iabs(k1);
	const bool first = a0 > 10 && a0 < 32 && a1 >= 23;
	return !first && a1 >= 16 && a1 < 32 && a0 >= 23;
}
/* The pair rules only ask which of eight magnitude classes the two contrast values are in -- up to 10, 11, 12..15, 16..22, 23..31,
 * 32..176, 177..201, above (the constants of image_processing.c:810-837, :1927-1990) -- and their signs: 15 signed classes, so the ~60
 * predicate operations per pair become two class look-ups and one table entry.  The table is filled at kernel start by evaluating the
 * rules themselves on one representative per class (k_front_image, nhw_front_image.h). */
#define PCLS 15
__device__ __forceinline__ int pair_class_rep(int sc)           /* a value of signed class sc = -7 .. 7 (0: |k| <= 10) */
{
	const int m = iabs(sc);
	const int mag = m == 0 ? 5 : m == 1 ? 11 : m == 2 ? 13 : m == 3 ? 18 : m == 4 ? 27 : m == 5 ? 100 : m == 6 ? 190 : 300;
	return sc < 0 ? -mag : mag;
}
__device__ __forceinline__ int pair_mag_class(int a)             /* a = |k| */
{
	return (a > 10) + (a > 11) + (a > 15) + (a > 22) + (a > 31) + (a > 176) + (a > 201);
}

} // namespace nhw
#include "nhw_front_image.h"
namespace nhw {

/* SURVEY.md section 8d generator, one lane per image (setup only, never timed) */
__global__ void k_synth(uint8_t *__restrict__ bgr, int n, uint32_t seed_base)
{
	const int img = blockIdx.x * blockDim.x + threadIdx.x;
	if (img >= n) return;
	uint32_t x = 0x9E3779B9u * (seed_base + (uint32_t)img + 1u);
	if (!x) x = 1;
	uint8_t lat[17 * 17 * 3];
	for (int i = 0; i < 17 * 17 * 3; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; lat[i] = (uint8_t)(x >> 24); }
	uint32_t *out = reinterpret_cast<uint32_t *>(bgr + (size_t)img * (W * W * 3));
	uint32_t word = 0; int nb = 0;
	for (int yy = 0; yy < W; yy++) {
		const int cy = yy >> 5, wy = (yy & 31) << 3;
		for (int xx = 0; xx < W; xx++) {
			const int cx = xx >> 5, wx = (xx & 31) << 3;
			for (int c = 0; c < 3; c++) {
				const int l00 = lat[(cy * 17 + cx) * 3 + c], l01 = lat[(cy * 17 + cx + 1) * 3 + c];
				const int l10 = lat[((cy + 1) * 17 + cx) * 3 + c], l11 = lat[((cy + 1) * 17 + cx + 1) * 3 + c];
				const int v = ((l00 * (256 - wx) + l01 * wx) * (256 - wy) + (l10 * (256 - wx) + l11 * wx) * wy + 32768) >> 16;
				x ^= x << 13; x ^= x >> 17; x ^= x << 5;
				int b = v + (int)(x % 13u) - 6;
				b = b < 0 ? 0 : (b > 255 ? 255 : b);
				word |= (uint32_t)b << (8 * nb);
				if (++nb == 4) { *out++ = word; word = 0; nb = 0; }
			}
		}
	}
}

} // namespace nhw

/* ------------------------------------------------------------------------------------------------ launchers */
using namespace nhw;

static float color_yq(int q, int *family);
void nhw_launch_color(const uint8_t *bgr, int n, int q, int16_t *y, size_t y_stride, uint8_t *u, uint8_t *v, size_t c_stride, hipStream_t s)
{
	const dim3 grid(H / 4, n);
	int fam = 0;
	const float yq = color_yq(q, &fam);
	if (fam == 0) k_color<0><<<grid, 256, 0, s>>>(bgr, y, y_stride, u, v, c_stride, yq);
	else if (fam == 1) k_color<1><<<grid, 256, 0, s>>>(bgr, y, y_stride, u, v, c_stride, yq);
	else if (fam == 2) k_color<2><<<grid, 256, 0, s>>>(bgr, y, y_stride, u, v, c_stride, yq);
	else k_color<3><<<grid, 256, 0, s>>>(bgr, y, y_stride, u, v, c_stride, yq);         /* quality table of colorspace.c:174-189 (format constants) */
}

/* keep != nullptr: copy of the first 256 rows x 512 of the transposed pass-1 plane (q>=22, level 0) */
__device__ __forceinline__ uint32_t pk_max_u16x(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
/* ------------------------------------------------------------------------------------------------
 * Whole-block filterbank kernels for the 256- and 128-sized levels: one workgroup keeps the S x S block in
 * LDS (row stride S + 2 shorts: column walks hit 64 different banks), runs both directions there and writes
 * every plane the four-kernel sequence (rows, transpose, rows, transpose) leaves behind -- the coefficient
 * plane, the transposed first-direction plane in the source plane and, on a non-final level, the LL quadrant
 * copied back in natural orientation -- with one read and one write of each.  A wavefront owns a row (then a
 * column): it reads all its taps before it writes its outputs over them, so both directions run in place.
 * ------------------------------------------------------------------------------------------------ */
/* eight cells of a row of the block: from the int16 plane, or widened from a byte plane */
__device__ __forceinline__ uint4 ana_piece(const int16_t *src, const uint8_t *src8, int row, int o, int stride, int S)
{
	if (!src8) return *reinterpret_cast<const uint4 *>(src + (size_t)row * stride + 8 * o);
	const uint2 b = *reinterpret_cast<const uint2 *>(src8 + (size_t)row * S + 8 * o);
	return make_uint4((b.x & 0xFF) | ((b.x >> 8 & 0xFF) << 16), (b.x >> 16 & 0xFF) | ((b.x >> 24) << 16), (b.y & 0xFF) | ((b.y >> 8 & 0xFF) << 16), (b.y >> 16 & 0xFF) | ((b.y >> 24) << 16));
}
/* The second direction (filters.c:88-287) of two columns at once: Ew / Ow hold the even / odd rows' cells of the two columns, a lane its own row pair
 * k = lane + 64 u; lo / hi: what the pair leaves for its two columns.  left: the columns lie in the first direction's low-pass half. */
template <int PPL, int HLF, bool IN_RANGE = false /* the caller vouches for the 16-bit range (cells made from bytes): no test, no 32-bit form */>
__device__ __forceinline__ void ana_col_pair(const uint32_t (&Ew)[PPL], const uint32_t (&Ow)[PPL], bool left, int lane, int (&lo)[PPL][2], int (&hi)[PPL][2])
{
	/* Two columns side by side in packed 16-bit arithmetic wherever nothing can leave 16 bits: with every cell of the wavefront's two columns in
	 * -1300 .. 3000 the un-normalised sums stay inside (10 x 3000 + 2 x 1300 < 32768) -- which is every block of a real picture (the level-2
	 * input is LL1, the level-1 chroma input a byte plane).  A block outside that range takes the 32-bit form below, which follows the
	 * reference's int arithmetic where it wraps. */
	bool wide = false;
	if (!IN_RANGE)
#pragma unroll
	for (int u = 0; u < PPL; u++) {
		const uint32_t mx = pk_max_u16x(pk_add16(Ew[u], 0x05140514u), pk_add16(Ow[u], 0x05140514u));   /* + 1300: in range = at most 4300 as unsigned */
		wide |= (mx & 0xFFFFu) > 4300u || (mx >> 16) > 4300u;
	}
	if (IN_RANGE || !__any(wide)) {
		uint32_t rlast = 0;
#pragma unroll
		for (int u = 0; u < PPL; u++) {
			const int k = lane + 64 * u;
			uint32_t em = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Ew[u], 0x138, 0xF, 0xF, false), om = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Ow[u], 0x138, 0xF, 0xF, false);
			uint32_t en = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Ew[u], 0x130, 0xF, 0xF, false);
			if (u > 0) {
				const uint32_t se = (uint32_t)__builtin_amdgcn_readlane((int)Ew[u > 0 ? u - 1 : 0], 63), so_ = (uint32_t)__builtin_amdgcn_readlane((int)Ow[u > 0 ? u - 1 : 0], 63);
				if (lane == 0) { em = se; om = so_; }
			}
			if (u + 1 < PPL) { const uint32_t se = (uint32_t)__builtin_amdgcn_readlane((int)Ew[u + 1 < PPL ? u + 1 : u], 0); if (lane == 63) en = se; }
			else if (lane == 63) en = Ew[u];
			if (u == 0 && lane == 0) { em = en; om = Ow[u]; }
			const s16x2 e0 = as_s(Ew[u]), o0 = as_s(Ow[u]), em1 = as_s(em), om1 = as_s(om), e1 = as_s(en);
			const s16x2 r = e0 * (s16x2)(short)6 + ((om1 + o0) << 1) - (em1 + e1);
			s16x2 a = e0 + e1;
			a = a + (a & (em1 + e0) & as_s((k & 1) ? 0x00010001u : 0u));
			const s16x2 pp = o0 - (a >> 1), tail = o0 - e0;
			s16x2 l, h;
			if (left) {
				uint32_t rp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)as_w(r), 0x138, 0xF, 0xF, false);
				if (lane == 0) rp = rlast;
				const s16x2 carry = k > 0 ? pk_diffuse(as_s(rp)) : (s16x2)(short)0;
				rlast = (uint32_t)__builtin_amdgcn_readlane((int)as_w(r), 63);
				l = pk_rnd_half_away(r + carry, 6);
				h = k < HLF - 1 ? pk_rnd_half_away(pp, 3) : (tail >> 3);
			} else {
				l = pk_rnd_half_away(r, 4);
				h = k < HLF - 1 ? pk_rnd_half_away(pp, 1) : ((tail + (s16x2)(short)1) >> 1);   /* pp > 0 ? (pp + 1) >> 1 : pp >> 1 is rounding half away at shift 1 */
			}
			lo[u][0] = l.x; lo[u][1] = l.y; hi[u][0] = h.x; hi[u][1] = h.y;
		}
	} else {
	int rlast[2] = { 0, 0 };                                    /* r of cell 63 of the half before (the seam of the carry) */
#pragma unroll
		for (int u = 0; u < PPL; u++) {
			const int k = lane + 64 * u;
			uint32_t em = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Ew[u], 0x138, 0xF, 0xF, false), om = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Ow[u], 0x138, 0xF, 0xF, false);
			uint32_t en = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Ew[u], 0x130, 0xF, 0xF, false);
			if (u > 0) {
				const uint32_t se = (uint32_t)__builtin_amdgcn_readlane((int)Ew[u > 0 ? u - 1 : 0], 63), so_ = (uint32_t)__builtin_amdgcn_readlane((int)Ow[u > 0 ? u - 1 : 0], 63);
				if (lane == 0) { em = se; om = so_; }
			}
			if (u + 1 < PPL) { const uint32_t se = (uint32_t)__builtin_amdgcn_readlane((int)Ew[u + 1 < PPL ? u + 1 : u], 0); if (lane == 63) en = se; }
			else if (lane == 63) en = Ew[u];                           /* x[S] = x[S - 2] */
			if (u == 0 && lane == 0) { em = en; om = Ow[u]; }           /* x[-2] = x[2], x[-1] = x[1] */
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int e0 = h ? (int)Ew[u] >> 16 : (int16_t)(Ew[u] & 0xFFFF), o0 = h ? (int)Ow[u] >> 16 : (int16_t)(Ow[u] & 0xFFFF);
				const int em1 = h ? (int)em >> 16 : (int16_t)(em & 0xFFFF), om1 = h ? (int)om >> 16 : (int16_t)(om & 0xFFFF), e1 = h ? (int)en >> 16 : (int16_t)(en & 0xFFFF);
				const int r = 6 * e0 + 2 * (om1 + o0) - (em1 + e1);
				int a = e0 + e1;
				if ((k & 1) && (a & 1) && ((em1 + e0) & 1)) a++;
				const int pp = o0 - (a >> 1), tail = o0 - e0;          /* the predicted odd sample; the last one: x[S-1] - x[S-2] */
				if (left) {
					int rp = __builtin_amdgcn_update_dpp(0, r, 0x138, 0xF, 0xF, false);   /* the cell before: its carry comes in (filters.c:203-287) */
					if (lane == 0) rp = rlast[h];
					const int carry = k > 0 ? diffuse(rp) : 0;
					rlast[h] = __builtin_amdgcn_readlane(r, 63);
					lo[u][h] = rnd_half_away((int16_t)(r + carry), 6);
					hi[u][h] = k < HLF - 1 ? rnd_half_away(pp, 3) : (tail >> 3);
				} else {
					lo[u][h] = rnd_half_away(r, 4);
					hi[u][h] = k < HLF - 1 ? (pp > 0 ? (pp + 1) >> 1 : pp >> 1) : ((tail + 1) >> 1);
				}
			}
		}
	}
}

template <int S>
__global__ __launch_bounds__(S * 4) void k_dwt_ana(int16_t *__restrict__ jpegb, int16_t *__restrict__ procb, size_t plane_stride, int stride, int final_level,
                                                   int16_t *__restrict__ saveb, size_t save_plane, int save_row, int save_kind /* 1: copy of the S x S coefficient block, 2: of the LL quadrant copied back */, int n,
                                                   const uint8_t *__restrict__ src8b, size_t src8_plane /* the block as S x S bytes (a 4:2:0 chroma plane: nhw_encoder.c:2257-2263 widens it first), or null */,
                                                   int drop_t /* the transposed first-direction plane is not stored: nothing reads it behind a chroma analysis (the dequantiser simulation rewrites every cell) */,
                                                   const int16_t *__restrict__ altb, size_t alt_plane, int alt_stride /* the block is read from another int16 plane (the first level-2 analysis of the luma takes the LL rows from ll1: the front no longer copies them into the work plane), or null */)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	constexpr int LS = S + 2, HLF = S / 2, PPL = HLF / 64, NT_ = S * 4, NPRE = S * (S / 8) / NT_;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	int16_t *A = smem;
	/* The block fills the LDS a CU has (S = 256), so a CU holds one workgroup and its load, filter and store phases would follow one another with
	 * the memory system idle in between.  A workgroup therefore works through several images and has the next block on its way, in registers,
	 * while it filters the present one.  The barriers order LDS traffic only (no thread reads global memory another one wrote). */
	uint4 pre[NPRE];
	const int sstride = altb ? alt_stride : stride;
	if ((int)blockIdx.x < n) {
		const int16_t *src = altb ? altb + (size_t)blockIdx.x * alt_plane : jpegb + (size_t)blockIdx.x * plane_stride;
#pragma unroll
		for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = ana_piece(src, src8b ? src8b + (size_t)blockIdx.x * src8_plane : nullptr, v / (S / 8), v % (S / 8), sstride, S); }
	}
	for (int img = blockIdx.x; img < n; img += gridDim.x) {
	int16_t *jpeg = jpegb + (size_t)img * plane_stride, *proc = procb + (size_t)img * plane_stride;
	int16_t *save = saveb ? saveb + (size_t)img * save_plane : nullptr;
#pragma unroll
	for (int u = 0; u < NPRE; u++) {
		const int v = t + u * NT_, row = v / (S / 8), o = v % (S / 8);
		uint32_t *d = reinterpret_cast<uint32_t *>(A + row * LS + 8 * o);
		d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
	}
	lds_barrier();
	if (img + (int)gridDim.x < n) {
		const int16_t *src = altb ? altb + (size_t)(img + gridDim.x) * alt_plane : jpegb + (size_t)(img + gridDim.x) * plane_stride;
#pragma unroll
		for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = ana_piece(src, src8b ? src8b + (size_t)(img + gridDim.x) * src8_plane : nullptr, v / (S / 8), v % (S / 8), sstride, S); }
	}
	/* Both directions read a line two cells to a dword, a lane its own pair (cells 2k, 2k+1); the pair on the left and the first cell on the right
	 * come over the lanes (DPP shifts by one lane, the seam between the two halves of a 256-cell line through a readlane): one LDS read per lane
	 * and output pair where there were five 16-bit ones -- the kernel was bound by its LDS instructions. */
	for (int i = 0; i < 16; i++) {                                 /* first direction (filters.c:40-86): un-normalised taps */
		int16_t *x = A + (wv * 16 + i) * LS;
		uint32_t Dw[PPL];
		int lo[PPL], hi[PPL];
#pragma unroll
		for (int u = 0; u < PPL; u++) Dw[u] = reinterpret_cast<const uint32_t *>(x)[lane + 64 * u];
#pragma unroll
		for (int u = 0; u < PPL; u++) {
			uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Dw[u], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
			uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Dw[u], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
			if (u > 0) { const uint32_t seam = (uint32_t)__builtin_amdgcn_readlane((int)Dw[u > 0 ? u - 1 : 0], 63); if (lane == 0) pv = seam; }
			if (u + 1 < PPL) { const uint32_t seam = (uint32_t)__builtin_amdgcn_readlane((int)Dw[u + 1 < PPL ? u + 1 : u], 0); if (lane == 63) nx = seam; }
			else if (lane == 63) nx = Dw[u];                           /* x[S] = x[S - 2] */
			const int e0 = (int16_t)(Dw[u] & 0xFFFF), o0 = (int)Dw[u] >> 16, e1 = (int16_t)(nx & 0xFFFF);
			int em1 = (int16_t)(pv & 0xFFFF), om1 = (int)pv >> 16;
			if (u == 0 && lane == 0) { em1 = e1; om1 = o0; }           /* x[-2] = x[2], x[-1] = x[1] */
			lo[u] = 6 * e0 + 2 * (om1 + o0) - (em1 + e1);
			hi[u] = (o0 << 1) - (e0 + e1);                             /* the last one: (x[S-1] - x[S-2]) << 1, which is what e1 = e0 gives */
		}
#pragma unroll
		for (int u = 0; u < PPL; u++) { x[lane + 64 * u] = (int16_t)lo[u]; x[HLF + lane + 64 * u] = (int16_t)hi[u]; }
	}
	lds_barrier();
	if (!drop_t)
	for (int v = t; v < S * HLF; v += NT_) {                       /* the source plane keeps the transposed first-direction plane */
		const int i = v / HLF, j = 2 * (v % HLF);
		if (!final_level && i < HLF && j < HLF) continue;           /* (its LL quadrant is overwritten below) */
		*reinterpret_cast<uint32_t *>(jpeg + (size_t)i * stride + j) = (uint16_t)A[j * LS + i] | ((uint32_t)(uint16_t)A[(j + 1) * LS + i] << 16);
	}
	lds_barrier();
	for (int i = 0; i < 8; i++) {                                  /* second direction along the columns (filters.c:88-287), two columns (one dword) at a time */
		const int c = wv * 16 + 2 * i;
		const bool left = c < HLF;                                 /* the same for both columns and for the whole wavefront */
		uint32_t Ew[PPL], Ow[PPL];
		int lo[PPL][2], hi[PPL][2];
#pragma unroll
		for (int u = 0; u < PPL; u++) {
			const int k = lane + 64 * u;
			Ew[u] = *reinterpret_cast<const uint32_t *>(A + (2 * k) * LS + c); Ow[u] = *reinterpret_cast<const uint32_t *>(A + (2 * k + 1) * LS + c);
		}
		ana_col_pair<PPL, HLF>(Ew, Ow, left, lane, lo, hi);
#pragma unroll
		for (int h = 0; h < 2; h++) {
			int16_t *o = proc + (size_t)(c + h) * stride;
#pragma unroll
			for (int u = 0; u < PPL; u++) {
				const int k = lane + 64 * u;
				o[k] = (int16_t)lo[u][h]; o[HLF + k] = (int16_t)hi[u][h];
				if (save_kind == 1) { save[(size_t)(c + h) * save_row + k] = (int16_t)lo[u][h]; save[(size_t)(c + h) * save_row + HLF + k] = (int16_t)hi[u][h]; }
			}
		}
		if (!final_level && left) {                                  /* LL, parked in the columns' own cells (every lane has read its taps by now) */
#pragma unroll
			for (int u = 0; u < PPL; u++) *reinterpret_cast<uint32_t *>(A + (lane + 64 * u) * LS + c) = (uint32_t)(uint16_t)lo[u][0] | ((uint32_t)(uint16_t)lo[u][1] << 16);
		}
	}
	if (!final_level) {
	lds_barrier();
	for (int v = t; v < HLF * (HLF / 2); v += NT_) {               /* LL copied back in natural orientation (wavelet_filterbank.c:172-184) */
		const int k = v / (HLF / 2), c = 2 * (v % (HLF / 2));
		const uint32_t w = (uint16_t)A[k * LS + c] | ((uint32_t)(uint16_t)A[k * LS + c + 1] << 16);
		*reinterpret_cast<uint32_t *>(jpeg + (size_t)k * stride + c) = w;
		if (save_kind == 2) *reinterpret_cast<uint32_t *>(save + (size_t)k * save_row + c) = w;
	}
	}
	lds_barrier();                                                 /* the block is done with before the next one moves in */
	}
}

/* The level-1 analysis of a 4:2:0 chroma plane (256 x 256 bytes -> the coefficient plane, the LL quadrant in natural orientation and its copy), a
 * QUARTER of the block to a workgroup: quarter (xh, part) owns the 64 first-direction outputs 64 part .. 64 part + 63 of the low-pass (xh = 0) or
 * high-pass (xh = 1) half of every row -- it reads its 132 bytes of each row straight from the byte plane, filters them on the way into LDS
 * (256 rows x 64 cells, 33 KB: four workgroups a CU), runs the second direction down its 64 columns and writes 64 whole rows of the coefficient
 * plane.  k_dwt_ana<256> holds the block in one 1024-thread workgroup a CU, whose load, filter and store phases follow one another: 0.36 ms a
 * launch for 1.07 GB; four independent workgroups a CU keep the memory system busy through each other's filter phases.
 * Block b -> picture ((b >> 5) << 3) | (b & 7), quarter (b >> 3) & 3: the four quarters of a picture sit on one XCD (b mod 8) and read the
 * picture's bytes through one L2. */
#define CQ_LS 66
#define CQ_ROW(r) ((((r) & 1) * 128 + ((r) >> 1)) * CQ_LS)   /* even rows first, then the odd ones: the column pass reads a lane's even row and odd row at a pitch of 33 dwords each -- no bank conflicts (k_dwt_ana's 129-dword pitch puts rows 2k on 16 banks) */
__global__ __launch_bounds__(256) void k_chroma_l1q(const uint8_t *__restrict__ src8b, size_t src8_plane, int16_t *__restrict__ procb, int16_t *__restrict__ jpegb, size_t plane_stride,
                                                    int16_t *__restrict__ saveb, size_t save_plane, int save_row, int n, int ll_to_jpeg /* 0: the LL quadrant only goes to its copy (the level-2 analysis reads it there) */)
{
	constexpr int S = 256, HLF = 128, PPL = 2, stride = 256;
	__shared__ __attribute__((aligned(16))) int16_t A[S * CQ_LS];
	__shared__ uint16_t s_halo[S];
	const int b = blockIdx.x, img = ((b >> 5) << 3) | (b & 7), qd = (b >> 3) & 3;
	if (img >= n) return;
	const int xh = qd >> 1, part = qd & 1;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const uint8_t *src = src8b + (size_t)img * src8_plane;
	int16_t *proc = procb + (size_t)img * plane_stride, *jpeg = jpegb + (size_t)img * plane_stride;
	/* what lies beside the quarter's 128 bytes of a row: bytes 126, 127 on the left of the second quarter, byte 128 on the right of the first (the
	 * row's own ends mirror: x[-2] = x[2], x[-1] = x[1], x[256] = x[254]) */
	s_halo[t] = part ? *reinterpret_cast<const uint16_t *>(src + (size_t)t * S + 126) : (uint16_t)src[(size_t)t * S + 128];
	/* first direction (filters.c:40-86): a lane four bytes = two outputs of a row, half a wavefront a row; un-normalised taps as byte dot products
	 * (v_dot4_u32_u8: the positive and the negative taps apart).  The bytes on either side of the lane's four come over DPP; the quarter's first and
	 * last lane of a row take them from the halo or the row's mirror. */
	const int rsub = lane >> 5, kk = lane & 31;
	constexpr int CQ_B = 32;                                       /* rows' loads in flight a lane: all of them (8 at a time the workgroup waited for memory four times over: 0.59 ms for the two launches, 16 or 32: 0.51) */
	uint32_t w[CQ_B];
	auto first_dir = [&](auto xh_tag) {
		constexpr int XH = decltype(xh_tag)::value;
#pragma unroll 1
		for (int it0 = 0; it0 < 32; it0 += CQ_B) {
#pragma unroll
			for (int j = 0; j < CQ_B; j++) w[j] = *reinterpret_cast<const uint32_t *>(src + (size_t)((it0 + j) * 8 + wv * 2 + rsub) * S + 128 * part + 4 * kk);
			if (it0 == 0) __syncthreads();                              /* the halo is in place */
#pragma unroll
			for (int j = 0; j < CQ_B; j++) {
				const int row = (it0 + j) * 8 + wv * 2 + rsub;
				const uint32_t x = w[j];
				uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
				uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
				const uint32_t hl = s_halo[row];
				/* bytes 2, 3 of pv = the two cells on my left, byte 0 of nx = the cell on my right */
				const uint32_t pv_edge = part ? hl << 16 : __builtin_amdgcn_perm(x, x, 0x01020000u);   /* mirror: x[-2] = x[2] (my byte 2), x[-1] = x[1] (my byte 1) */
				const uint32_t nx_edge = part ? (x >> 16) & 255u : hl & 255u;                              /* mirror: x[256] = x[254] (my byte 2) */
				pv = kk == 0 ? pv_edge : pv;
				nx = kk == 31 ? nx_edge : nx;
				uint32_t o0, o1;
				if (XH == 0) {
					o0 = __builtin_amdgcn_udot4(x, 0x00000206u, __builtin_amdgcn_udot4(pv, 0x02000000u, 0u, false), false) - __builtin_amdgcn_udot4(x, 0x00010000u, __builtin_amdgcn_udot4(pv, 0x00010000u, 0u, false), false);
					o1 = __builtin_amdgcn_udot4(x, 0x02060200u, 0u, false) - __builtin_amdgcn_udot4(x, 0x00000001u, __builtin_amdgcn_udot4(nx, 0x00000001u, 0u, false), false);
				} else {
					o0 = __builtin_amdgcn_udot4(x, 0x00000200u, 0u, false) - __builtin_amdgcn_udot4(x, 0x00010001u, 0u, false);
					o1 = __builtin_amdgcn_udot4(x, 0x02000000u, 0u, false) - __builtin_amdgcn_udot4(x, 0x00010000u, __builtin_amdgcn_udot4(nx, 0x00000001u, 0u, false), false);
				}
				*reinterpret_cast<uint32_t *>(A + CQ_ROW(row) + 2 * kk) = (o0 & 0xFFFFu) | (o1 << 16);
			}
		}
	};
	if (xh == 0) first_dir(std::integral_constant<int, 0>()); else first_dir(std::integral_constant<int, 1>());
	__syncthreads();
	/* second direction down the quarter's columns, two at a time; a wavefront 16 columns */
	const bool left = xh == 0;
	const int cbase = HLF * xh + 64 * part;                            /* the quarter's first row of the coefficient plane */
	for (int i = 0; i < 8; i++) {
		const int c = wv * 16 + 2 * i;
		uint32_t Ew[PPL], Ow[PPL];
		int lo[PPL][2], hi[PPL][2];
#pragma unroll
		for (int u = 0; u < PPL; u++) {
			const int k = lane + 64 * u;
			Ew[u] = *reinterpret_cast<const uint32_t *>(A + CQ_ROW(2 * k) + c); Ow[u] = *reinterpret_cast<const uint32_t *>(A + CQ_ROW(2 * k + 1) + c);
		}
		ana_col_pair<PPL, HLF, true>(Ew, Ow, left, lane, lo, hi);   /* first-pass cells of a byte plane: -510 .. 2550 */
#pragma unroll
		for (int h = 0; h < 2; h++) {
			int16_t *o = proc + (size_t)(cbase + c + h) * stride;
#pragma unroll
			for (int u = 0; u < PPL; u++) { const int k = lane + 64 * u; o[k] = (int16_t)lo[u][h]; o[HLF + k] = (int16_t)hi[u][h]; }
		}
		if (left) {                                                  /* LL, parked in the columns' own cells (every lane has read its taps by now) */
#pragma unroll
			for (int u = 0; u < PPL; u++) *reinterpret_cast<uint32_t *>(A + CQ_ROW(lane + 64 * u) + c) = (uint32_t)(uint16_t)lo[u][0] | ((uint32_t)(uint16_t)lo[u][1] << 16);
		}
	}
	if (!left) return;
	__syncthreads();
	int16_t *save = saveb ? saveb + (size_t)img * save_plane : nullptr;
	for (int v = t; v < HLF * 32; v += 256) {                          /* LL copied back in natural orientation (wavelet_filterbank.c:172-184), and its copy */
		const int k = v >> 5, c = 2 * (v & 31);
		const uint32_t wd = *reinterpret_cast<const uint32_t *>(A + CQ_ROW(k) + c);
		if (ll_to_jpeg) *reinterpret_cast<uint32_t *>(jpeg + (size_t)k * stride + 64 * part + c) = wd;
		if (save) *reinterpret_cast<uint32_t *>(save + (size_t)k * save_row + 64 * part + c) = wd;
	}
}

template <int S>
__global__ __launch_bounds__(S * 4) void k_dwt_syn(int16_t *__restrict__ jpegb, int16_t *__restrict__ procb, size_t plane_stride, int stride, int n,
                                                   int drop_nat /* the reconstruction in natural orientation is not stored (nothing reads it: the chroma loops, the second luma loop below q22) */,
                                                   const uint16_t *__restrict__ verb_list, size_t verb_list_stride /* bytes */, const int *__restrict__ verb_len, size_t verb_len_stride /* bytes */
                                                   /* second luma loop, production: the LL2 samples the LL coder sent verbatim keep their exact value (nhw_encoder.c:2728-2735) -- the work plane's
                                                    * sample goes over the block's before the block is filtered (the dequantiser simulation did this while the coder ran in front of it) */)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	constexpr int LS = S + 2, HLF = S / 2, PPL = HLF / 64, NT_ = S * 4, NPRE = S * (S / 8) / NT_;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	int16_t *A = smem;
	uint4 pre[NPRE];                                               /* several images per workgroup, the next block on its way while this one is filtered (see k_dwt_ana) */
	if ((int)blockIdx.x < n) {
		const int16_t *src = jpegb + (size_t)blockIdx.x * plane_stride;
#pragma unroll
		for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(v / (S / 8)) * stride + 8 * (v % (S / 8))); }
	}
	for (int img = blockIdx.x; img < n; img += gridDim.x) {
	int16_t *jpeg = jpegb + (size_t)img * plane_stride, *proc = procb + (size_t)img * plane_stride;
#pragma unroll
	for (int u = 0; u < NPRE; u++) {
		const int v = t + u * NT_, row = v / (S / 8), o = v % (S / 8);
		uint32_t *d = reinterpret_cast<uint32_t *>(A + row * LS + 8 * o);
		d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
	}
	lds_barrier();
	if (verb_list) {
		const int nm = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(verb_len) + (size_t)img * verb_len_stride);
		const uint16_t *vl = reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(verb_list) + (size_t)img * verb_list_stride);
		for (int i = t; i < nm; i += NT_) { const int idx = vl[i], r = idx >> 7, cc = idx & 127; A[r * LS + cc] = proc[(size_t)r * stride + cc]; }
		if (nm) lds_barrier();
	}
	if (img + (int)gridDim.x < n) {
		const int16_t *src = jpegb + (size_t)(img + gridDim.x) * plane_stride;
#pragma unroll
		for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(v / (S / 8)) * stride + 8 * (v % (S / 8))); }
	}
	for (int i = 0; i < 16; i++) {                                 /* first direction, un-normalised */
		int16_t *x = A + (wv * 16 + i) * LS;
		int e[PPL], o[PPL];
#pragma unroll
		for (int u = 0; u < PPL; u++) syn_pair<S>(x, 1, lane + 64 * u, false, &e[u], &o[u]);
#pragma unroll
		for (int u = 0; u < PPL; u++) reinterpret_cast<uint32_t *>(x)[lane + 64 * u] = (uint32_t)(uint16_t)e[u] | ((uint32_t)(uint16_t)o[u] << 16);
	}
	lds_barrier();
	for (int i = 0; i < 16; i++) {                                 /* second direction along the columns, normalised: row c of the work plane */
		const int c = wv * 16 + i;
		int16_t *x = A + c;
		int e[PPL], o[PPL];
#pragma unroll
		for (int u = 0; u < PPL; u++) syn_pair<S>(x, LS, lane + 64 * u, true, &e[u], &o[u]);
		uint32_t *dst = reinterpret_cast<uint32_t *>(proc + (size_t)c * stride);
#pragma unroll
		for (int u = 0; u < PPL; u++) {
			const int k = lane + 64 * u;
			dst[k] = (uint32_t)(uint16_t)e[u] | ((uint32_t)(uint16_t)o[u] << 16);
			x[(2 * k) * LS] = (int16_t)e[u]; x[(2 * k + 1) * LS] = (int16_t)o[u];
		}
	}
	lds_barrier();
	if (!drop_nat)
	for (int v = t; v < S * (S / 8); v += NT_) {                   /* and its transpose, the reconstruction in natural orientation */
		const int row = v / (S / 8), o = v % (S / 8);
		const uint32_t *d = reinterpret_cast<const uint32_t *>(A + row * LS + 8 * o);
		*reinterpret_cast<uint4 *>(jpeg + (size_t)row * stride + 8 * o) = make_uint4(d[0], d[1], d[2], d[3]);
	}
	lds_barrier();
	}
}

#define DWT_WGS 256                  /* one resident workgroup per CU for the 256 x 256 blocks */
/* The kernels that need more dynamic LDS than the default limit are opted in PER DEVICE, when a handle is created on it
 * (nhw_enc_create, after hipSetDevice): the attribute belongs to the device's copy of the function.  Returns the first failing call. */
int nhw_front_set_attrs(const char **where)
{
#define SETATTR(fn, bytes) do { const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
                                if (e_ != hipSuccess) { *where = "hipFuncSetAttribute(" #fn ", MaxDynamicSharedMemorySize)"; return (int)e_; } } while (0)
	SETATTR((k_front_plain<0, 0>), FP_LDS_BYTES); SETATTR((k_front_plain<1, 0>), FP_LDS_BYTES); SETATTR((k_front_image<1, 1, 0>), FI_LDS_BYTES);
	SETATTR((k_front_image<1, 1, 1>), FI_LDS_BYTES); SETATTR((k_front_image<1, 1, 2>), FI_LDS_BYTES);
	SETATTR(k_dwt_ana<256>, 256 * 258 * sizeof(int16_t)); SETATTR(k_dwt_syn<256>, 256 * 258 * sizeof(int16_t));
	SETATTR(k_dwt_ana<128>, 128 * 130 * sizeof(int16_t)); SETATTR(k_dwt_syn<128>, 128 * 130 * sizeof(int16_t));
#undef SETATTR
	return 0;
}

void nhw_launch_copy_block(const int16_t *src, size_t src_plane, int src_row, int16_t *dst, size_t dst_plane, int dst_row, int rows, int cols, int n, hipStream_t s);

/* save (optional): a second destination for the block the reference copies right after the transform -- the S x S coefficient block
 * (save_kind 1) or the LL quadrant in natural orientation (save_kind 2) -- written by the fused kernels, by a block copy otherwise */
void nhw_launch_analysis(int16_t *jpeg, int16_t *proc, int n, size_t plane_stride, int stride, int size, int final_level,
                         hipStream_t s, int16_t *save, size_t save_plane, int save_row, int save_kind, const uint8_t *src8, size_t src8_plane, int drop_t,
                         const int16_t *alt, size_t alt_plane, int alt_stride)
{
	if (!save) save_kind = 0;
	static const int quarters = getenv("NHW_CHROMA_L1Q") ? atoi(getenv("NHW_CHROMA_L1Q")) : 1;
	if (size == 256 && src8 && !final_level && drop_t && save_kind != 1 && !alt && stride == 256 && quarters)   /* the encoder's chroma level 1 from the byte plane */
		k_chroma_l1q<<<4 * ((n + 7) & ~7), 256, 0, s>>>(src8, src8_plane, proc, jpeg, plane_stride, save_kind == 2 ? save : nullptr, save_plane, save_row, n, !(drop_t == 2 && save_kind == 2));
	else if (size == 256) k_dwt_ana<256><<<n < DWT_WGS ? n : DWT_WGS, 1024, 256 * 258 * sizeof(int16_t), s>>>(jpeg, proc, plane_stride, stride, final_level, save, save_plane, save_row, save_kind, n, src8, src8_plane, drop_t, alt, alt_plane, alt_stride);
	else if (size == 128) k_dwt_ana<128><<<n, 512, 128 * 130 * sizeof(int16_t), s>>>(jpeg, proc, plane_stride, stride, final_level, save, save_plane, save_row, save_kind, n, nullptr, 0, drop_t, alt, alt_plane, alt_stride);
	else {   /* size 512 is the front kernels' (nhw_launch_front_fused); a caller with any other size would get stale planes: stop loudly */
		fprintf(stderr, "nhw_launch_analysis: no kernel for transform size %d (256 and 128 only; 512 is nhw_launch_front_fused)\n", size);
		abort();
	}
}

void nhw_launch_synthesis(int16_t *jpeg, int16_t *proc, int n, size_t plane_stride, int stride, int size, hipStream_t s, int drop_nat,
                          const uint16_t *verb_list, size_t verb_list_stride, const int *verb_len, size_t verb_len_stride)
{
	if (size == 256) k_dwt_syn<256><<<n < DWT_WGS ? n : DWT_WGS, 1024, 256 * 258 * sizeof(int16_t), s>>>(jpeg, proc, plane_stride, stride, n, drop_nat, verb_list, verb_list_stride, verb_len, verb_len_stride);
	else if (size == 128) k_dwt_syn<128><<<n, 512, 128 * 130 * sizeof(int16_t), s>>>(jpeg, proc, plane_stride, stride, n, drop_nat, nullptr, 0, nullptr, 0);
	else {
		fprintf(stderr, "nhw_launch_synthesis: no kernel for transform size %d (256 and 128 only)\n", size);
		abort();
	}
}

/* the front launch group = ONE kernel.
 * bgr != nullptr: quality 17..23, everything from the BGR bytes (colour, 4:2:0 planes pu / pv, pre-filter for q <= 21, level-1 analysis): k_front_image
 *                 with the pre-filter, k_front_plain without;
 * bgr == nullptr: the luma plane y is the input (quality 1..16 behind their own pre-filter; the analysis stage entry point): k_front_plain.
 * st (optional): the carry at the start of every image row, for the compatibility mode's replay of a few rows (k_front_stale).
 * switches bit 0: every carry segment takes its exact replay (tests); bit 1: a stage check is going to read every plane (nothing is left out). */
static float color_yq(int q, int *family)
{
	static const int k_qtz[17] = { 0, 15900, 16500, 17100, 18000, 18820, 19670, 20640, 21540, 23540, 25570, 27522, 27830, 27607, 28786, 31262, 32375 };
	if (q >= 20) { *family = 0; return 0.f; }
	if (q >= 18) { *family = 1; return q == 19 ? 0.975f : 0.93f; }
	if (q == 17) { *family = 2; return 0.f; }
	*family = 3;
	return __builtin_bit_cast(float, k_qtz[q < 1 ? 1 : q]);
}
void nhw_launch_front_fused(const uint8_t *bgr, int q, uint8_t *pu, uint8_t *pv, size_t c_stride, const int16_t *y, size_t y_stride, int with_prefilter,
                            uint8_t *st, size_t s_stride, int16_t *proc, int16_t *jpeg, size_t plane_stride, int16_t *ll1, size_t ll1_stride,
                            int16_t *keep, size_t keep_stride, int n, hipStream_t s, int switches)
{
	int fam = 0;
	const float yq = bgr ? color_yq(q, &fam) : 0.f;
	int fl = (switches & 1) | ((switches & 2) ? 0x100000 : 0);
#ifdef NHW_DEV
	{ const char *e = getenv("NHW_BAND_STOP"); if (e) fl |= atoi(e) << 8; }
	if (getenv("NHW_FRONT_PROF")) fl |= 0x10000;
	{ const char *e = getenv("NHW_FRONT_SKIP"); if (e) fl |= atoi(e) & 12; }     /* 4: no stores of the level-1 plane, 8: none of the LL rows (timing experiments) */
	if (getenv("NHW_FRONT_DUMP") && y && bgr && with_prefilter) { fl |= 2 | (atoi(getenv("NHW_FRONT_DUMP")) << 4); keep = const_cast<int16_t *>(y); keep_stride = y_stride / 2; }
#endif
#define FI_ARGS(srcp, sstride) srcp, sstride, yq, pu, pv, c_stride, st, s_stride, proc, jpeg, plane_stride, ll1, ll1_stride, keep, keep_stride, fl
#define FP_ARGS(srcp, sstride) srcp, sstride, yq, pu, pv, c_stride, proc, jpeg, plane_stride, ll1, ll1_stride, keep, keep_stride, fl
	if (!bgr) k_front_plain<0, 0><<<n, FI_NT, FP_LDS_BYTES, s>>>(FP_ARGS((const void *)y, y_stride));
	else if (!with_prefilter) k_front_plain<1, 0><<<n, FI_NT, FP_LDS_BYTES, s>>>(FP_ARGS((const void *)bgr, (size_t)0));
	else if (fam == 0) k_front_image<1, 1, 0><<<n, FI_NT, FI_LDS_BYTES, s>>>(FI_ARGS((const void *)bgr, (size_t)0));
	else if (fam == 1) k_front_image<1, 1, 1><<<n, FI_NT, FI_LDS_BYTES, s>>>(FI_ARGS((const void *)bgr, (size_t)0));
	else k_front_image<1, 1, 2><<<n, FI_NT, FI_LDS_BYTES, s>>>(FI_ARGS((const void *)bgr, (size_t)0));
#undef FP_ARGS
#undef FI_ARGS
#ifdef NHW_DEV
	if (getenv("NHW_FRONT_PROF") && bgr && with_prefilter) {
		unsigned long long h[16];
		hipStreamSynchronize(s);
		hipMemcpyFromSymbol(h, HIP_SYMBOL(nhw::g_fi_prof), sizeof h);
		static const char *nm[16] = { "loop top (hold rows, last barrier)", "barrier after phase 0 (+ prefetch issue)", "chroma vertical + barrier", "contrast + barrier", "entry states + barrier(s)", "replay + barrier", "pair rules + barrier", "horizontal + barrier", "vertical + barrier", "phase 0 work (wait for rows, colour, LDS stores)", "vertical: keep stores", "vertical: row copies + column loads", "vertical: arithmetic", "vertical: stores issued" };
		unsigned long long tot = 0; for (int i = 0; i < 14; i++) tot += h[i];
		fprintf(stderr, "k_front_image q%d: thread-0 clock ticks per image (sum over bands), %d images\n", q, n);
		for (int i = 0; i < 14; i++) fprintf(stderr, "  %-52s %10.0f  %5.1f %%\n", nm[i], (double)h[i] / n, 100.0 * h[i] / (tot ? tot : 1));
		memset(h, 0, sizeof h); hipMemcpyToSymbol(HIP_SYMBOL(nhw::g_fi_prof), h, sizeof h);
	}
#endif
}

/* Compatibility mode (NHW_COMPAT_GLIBC_ONESHOT) only: the kernel-map cells whose memory the stock binary's malloc hands out again as
 * res256's slack (row 128, columns 0..3) and as tree1 (from byte 262176 of the map on: rows 272..280 cover what is read before it is
 * written).  A lane replays one row from the row's entry state; out: [0..3] row 128, then 9 rows of 512. */
__global__ void k_front_stale(const int16_t *__restrict__ yb, size_t y_stride, const uint8_t *__restrict__ st, size_t s_stride, int16_t *__restrict__ stale, size_t stale_stride)
{
	const int img = blockIdx.x, lane = threadIdx.x;
	if (lane >= 10) return;
	const int16_t *y = (const int16_t *)((const uint8_t *)yb + (size_t)img * y_stride);
	int16_t *out = (int16_t *)((uint8_t *)stale + (size_t)img * stale_stride);
	const int row = lane ? 271 + lane : 128, ncols = lane ? W : 4;
	int16_t *dst = lane ? out + 4 + (lane - 1) * W : out;
	int carry = st[(size_t)img * s_stride + row] & 15;
	dst[0] = 0;                                                     /* column 0 (and 511) of the map are never written */
	for (int c = 1; c < ncols; c++) {
		int k = 0;
		if (c <= W - 2) {
			const int16_t *p = y + (size_t)row * W + c;
			const int ctr = p[0];
			int sum = 0, mag = 0;
			for (int dy = -1; dy <= 1; dy++)
				for (int dx = -1; dx <= 1; dx++) {
					if (!dy && !dx) continue;
					const int d = ctr - p[dy * W + dx];
					sum += d; mag += iabs(d);
				}
			if (sum == 0) carry = 0;
			else { const int acc = 15 * iabs(sum) + mag + ((carry + 2) >> 2); k = sum < 0 ? -(acc >> 4) : (acc >> 4); carry = acc & 15; }
		}
		dst[c] = (int16_t)k;
	}
}
void nhw_launch_front_stale(const int16_t *y, size_t y_stride, const uint8_t *st, size_t s_stride, int16_t *stale, size_t stale_stride, int n, hipStream_t s)
{
	k_front_stale<<<n, 64, 0, s>>>(y, y_stride, st, s_stride, stale, stale_stride);
}


void nhw_launch_synth(uint8_t *bgr, int n, uint32_t seed_base, hipStream_t s)
{
	k_synth<<<(n + 63) / 64, 64, 0, s>>>(bgr, n, seed_base);
}
