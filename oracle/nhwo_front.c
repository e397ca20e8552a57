/*
 * nhwo_front.c -- oracle, front half of the path: colour conversion + 4:2:0, the luma pre-filter,
 * and the separable integer 5/3 filterbank (analysis and synthesis).  TEST INFRASTRUCTURE ONLY.
 * See nhwo.h for the rules.  Compile with -ffp-contract=off.
 */
#include <stdlib.h>
#include <string.h>
#include "nhwo.h"

static inline int iabs(int v) { return v < 0 ? -v : v; }

/* ------------------------------------------------------------------------------------------
 * a1  colour conversion + chroma 4:2:0     reference: encoder/colorspace.c:55-260
 * ------------------------------------------------------------------------------------------ */

static inline uint8_t clip_u8(int v) { return (v >> 8) != 0 ? (v < 0 ? 0 : 255) : (uint8_t)v; } /* colorspace.c:83-94 */

/* chroma term: float intermediate, +128.5f / +128.4f by sign, truncation (colorspace.c:73-79) */
static inline int chroma_round(float cb) { return cb >= 0 ? (int)(cb + 128.5f) : (int)(cb + 128.4f); }

/* integer BT.601 scale table for q<=16 (colorspace.c:174-189), indexed by quality 1..16 */
static const int k_qtz[17] = { 0, 15900, 16500, 17100, 18000, 18820, 19670, 20640, 21540, 23540, 25570,
                               27522, 27830, 27607, 28786, 31262, 32375 };

void nhwo_color(const uint8_t *bgr, int quality, int16_t *y, uint8_t *u, uint8_t *v)
{
	/* full-resolution U and V (the reference overwrites bytes 1 and 2 of every pixel in place) */
	uint8_t *uf = (uint8_t *)malloc(2 * NHWO_DIM * NHWO_DIM);
	uint8_t *vf = uf + NHWO_DIM * NHWO_DIM;
	int p, r, c;

	for (p = 0; p < NHWO_DIM * NHWO_DIM; p++) {
		const int b0 = bgr[3 * p], b1 = bgr[3 * p + 1], b2 = bgr[3 * p + 2];
		int Y, U, V;
		if (quality >= 17) {
			/* double products summed left to right; Y is NOT clipped (colorspace.c:69,80) */
			double ly = 0.299 * b0 + 0.587 * b1 + 0.114 * b2;
			double lu = -0.1687 * b0 - 0.3313 * b1 + 0.5 * b2;
			double lv = 0.5 * b0 - 0.4187 * b1 - 0.0813 * b2;
			float cu, cv;
			if (quality >= 20) {                                /* colorspace.c:66-101 */
				Y = (int)(ly + 0.5f);
				cu = (float)lu; cv = (float)lv;
			} else if (quality >= 18) {                         /* colorspace.c:102-137 */
				float yq = (quality == 19) ? 0.975f : 0.93f;    /* float variable holding a double literal */
				Y = (int)(ly * yq + 0.5f);
				cu = (float)lu; cv = (float)lv;
			} else {                                            /* q17, colorspace.c:138-171 */
				Y = (int)(ly * 0.94 + 0.5f);
				cu = (float)(lu * 0.94); cv = (float)(lv * 0.94);
			}
			U = chroma_round(cu);
			V = chroma_round(cv);
		} else {                                                /* colorspace.c:172-214 */
			const int qz = k_qtz[quality < 1 ? 1 : quality];
			Y = (((66 * b0 + 129 * b1 + 25 * b2) * qz + 4194304) >> 23) + 16;
			U = (((-38 * b0 - 74 * b1 + 112 * b2) * qz + 4194304) >> 23) + 128;
			V = (((112 * b0 - 94 * b1 - 18 * b2) * qz + 4194304) >> 23) + 128;
		}
		y[p] = (int16_t)Y;
		uf[p] = clip_u8(U);
		vf[p] = clip_u8(V);
	}

	/* horizontal [1 2 1]/4 on even columns, taps are the untouched odd neighbours; column 0 is
	 * (c0+c1+1)>>1 (colorspace.c:220-234).  Then vertical [1 2 1]/4 onto even rows, row 0 is
	 * (r0+r1+1)>>1 (colorspace.c:241-256). */
	{
		uint8_t *hu = (uint8_t *)malloc(2 * NHWO_DIM * NHWO_HALF);
		uint8_t *hv = hu + NHWO_DIM * NHWO_HALF;
		for (r = 0; r < NHWO_DIM; r++) {
			const uint8_t *su = uf + r * NHWO_DIM, *sv = vf + r * NHWO_DIM;
			hu[r * NHWO_HALF] = (uint8_t)((su[0] + su[1] + 1) >> 1);
			hv[r * NHWO_HALF] = (uint8_t)((sv[0] + sv[1] + 1) >> 1);
			for (c = 1; c < NHWO_HALF; c++) {
				hu[r * NHWO_HALF + c] = (uint8_t)((su[2 * c - 1] + 2 * su[2 * c] + su[2 * c + 1] + 2) >> 2);
				hv[r * NHWO_HALF + c] = (uint8_t)((sv[2 * c - 1] + 2 * sv[2 * c] + sv[2 * c + 1] + 2) >> 2);
			}
		}
		for (c = 0; c < NHWO_HALF; c++) {
			u[c] = (uint8_t)((hu[c] + hu[NHWO_HALF + c] + 1) >> 1);
			v[c] = (uint8_t)((hv[c] + hv[NHWO_HALF + c] + 1) >> 1);
		}
		for (r = 1; r < NHWO_HALF; r++)
			for (c = 0; c < NHWO_HALF; c++) {
				const int up = (2 * r - 1) * NHWO_HALF + c, mid = up + NHWO_HALF, dn = mid + NHWO_HALF;
				u[r * NHWO_HALF + c] = (uint8_t)((hu[up] + 2 * hu[mid] + hu[dn] + 2) >> 2);
				v[r * NHWO_HALF + c] = (uint8_t)((hv[up] + 2 * hv[mid] + hv[dn] + 2) >> 2);
			}
		free(hu);
	}
	free(uf);
}

/* ------------------------------------------------------------------------------------------
 * a2  luma pre-filter, quality 17..21       reference: encoder/image_processing.c:558-2426
 *     pass A (:601-764): 8-neighbour contrast map with a 4-bit error carry that runs through the
 *     whole interior in raster order; pass B (:770-837 q>16 branch, :1927-1990): pixel pairs.
 * ------------------------------------------------------------------------------------------ */
int16_t nhwo_kernel_row128[4];   /* the four kernel-map values that sit behind res256 in the stock binary's heap (GLIBC_ONESHOT mode, nhwo_luma.c) */
int16_t nhwo_kernel_stale[16384]; /* kernel map from byte 262176 on: what the stock binary's malloc hands out as tree1 (same mode) */
int16_t nhwo_kernel_row256[4];   /* kernel map bytes 262160..262167 (row 256, columns 8..11): the 8 bytes of slack behind resIII in that heap (same mode, quality <= 13) */

void nhwo_prefilter_low(int16_t *y, int quality);   /* nhwo_prelow.c */

void nhwo_prefilter(int16_t *y, int quality)
{
	const int S = NHWO_DIM;
	int16_t *src = (int16_t *)malloc(sizeof(int16_t) * S * S);
	int16_t *kmap = (int16_t *)calloc(S * S, sizeof(int16_t)); /* borders never written: read as 0 */
	int r, c, carry = 0, prev_big = 0;

	if (quality <= 16) { free(kmap); free(src); nhwo_prefilter_low(y, quality); return; }
	memcpy(src, y, sizeof(int16_t) * S * S); /* image_processing.c:566 */

	for (r = 1; r < S - 1; r++)
		for (c = 1; c < S - 1; c++) {
			const int16_t *p = src + r * S + c;
			const int ctr = p[0];
			int sum = 0, mag = 0, dy, dx;
			for (dy = -1; dy <= 1; dy++)
				for (dx = -1; dx <= 1; dx++) {
					int d;
					if (!dy && !dx) continue;
					d = ctr - p[dy * S + dx];
					sum += d; mag += iabs(d);
				}
			if (sum == 0) {                         /* :759-764 */
				kmap[r * S + c] = 0; carry = 0;
			} else {                                /* :626-632, :686-692 */
				int acc = 15 * iabs(sum) + mag + ((carry + 2) >> 2);
				kmap[r * S + c] = (int16_t)(sum < 0 ? -(acc >> 4) : (acc >> 4));
				carry = acc & 15;
			}
		}

	for (c = 0; c < 4; c++) { nhwo_kernel_row128[c] = kmap[128 * S + c]; nhwo_kernel_row256[c] = kmap[256 * S + 8 + c]; }
	memcpy(nhwo_kernel_stale, kmap + 262176 / 2, sizeof nhwo_kernel_stale);

	for (r = 1; r < S - 1; r++)
		for (c = 1; c < S - 2; c += 2) {            /* pairs (1,2) (3,4) ... (509,510), :772-778 */
			int16_t *o = y + r * S + c;             /* o[0] = first of pair, o[1] = second */
			const int k0 = kmap[r * S + c], k1 = kmap[r * S + c + 1];
			int tag;

			/* :810-837 */
			if (k0 > 201) { o[0] -= 2; tag = 4; }
			else if (k0 < -201) { o[0] += 2; tag = 3; }
			else if (k0 > 176) { o[0]--; tag = 2; }
			else if (k0 < -176) { o[0]++; tag = 1; }
			else tag = 0;

			if (k1 > 201) { if (!tag || tag == 3) o[1] -= 2; else if (tag != 4) o[1]--; }
			else if (k1 < -201) { if (!tag || tag == 4) o[1] += 2; else if (tag != 3) o[1]++; }
			else if (k1 > 176) { if (tag != 4) o[1]--; }
			else if (k1 < -176) { if (tag != 3) o[1]++; }

			/* :1927-1990 (gate q>14 holds for every quality handled here) */
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
	(void)quality;
	free(kmap);
	free(src);
}

/* ------------------------------------------------------------------------------------------
 * a4..a6  1-D analysis filters             reference: encoder/filters.c:55-114, 203-287, 346-386
 *   All three share the taps; they differ in normalisation/rounding.  Symmetric (whole-sample)
 *   extension at both ends: x[-1]=x[1], x[-2]=x[2], x[n]=x[n-2].
 * ------------------------------------------------------------------------------------------ */
static inline int tap5(const int16_t *x, int n, int k)
{
	const int c = 2 * k;
	const int l1 = c >= 1 ? x[c - 1] : x[1], l2 = c >= 2 ? x[c - 2] : x[2];
	const int r1 = x[c + 1], r2 = (c + 2 < n) ? x[c + 2] : x[n - 2];
	return 6 * x[c] + 2 * (l1 + r1) - (l2 + r2);
}

/* pass 1, un-normalised: filters.c:346-386 (downfilter53IV) */
static void ana_raw(const int16_t *x, int n, int16_t *lo, int16_t *hi)
{
	const int h = n >> 1;
	int k;
	for (k = 0; k < h; k++) lo[k] = (int16_t)tap5(x, n, k);
	for (k = 0; k < h - 1; k++) hi[k] = (int16_t)((x[2 * k + 1] << 1) - (x[2 * k] + x[2 * k + 2]));
	hi[h - 1] = (int16_t)((x[n - 1] - x[n - 2]) << 1);
}

/* predict residual with the pair-coupled parity flag (filters.c:66-87, 212-234): outputs come in
 * pairs (2t, 2t+1); if the first sum of a pair is odd and the second sum is odd too, the second
 * sum is bumped by one before halving. */
static inline int pair_predict(const int16_t *x, int k)
{
	int a = x[2 * k] + x[2 * k + 2];
	if ((k & 1) && (a & 1) && ((x[2 * k - 2] + x[2 * k]) & 1)) a++;
	return x[2 * k + 1] - (a >> 1);
}

static inline int rnd_half_away(int v, int shift) /* (v + half)>>shift on the magnitude */
{
	const int half = 1 << (shift - 1);
	return v >= 0 ? (v + half) >> shift : -((-v + half) >> shift);
}

/* rows that are H in x: filters.c:55-114 (downfilter53) */
static void ana_hrow(const int16_t *x, int n, int16_t *lo, int16_t *hi)
{
	const int h = n >> 1;
	int k;
	for (k = 0; k < h; k++) lo[k] = (int16_t)rnd_half_away(tap5(x, n, k), 4);
	for (k = 0; k < h - 1; k++) {
		const int r = pair_predict(x, k);
		hi[k] = (int16_t)(r > 0 ? (r + 1) >> 1 : r >> 1);
	}
	hi[h - 1] = (int16_t)(((x[n - 1] - x[n - 2]) + 1) >> 1);
}

/* error-diffusion term of the normalised low-pass: filters.c:246-247, 267-276 */
static inline int diffuse(int r)
{
	if (r >= 0) { const int m = r & 63; return m < 32 ? (m >> 2) : -((64 - m) >> 2); }
	else { const int m = (-r) & 63; return m < 32 ? -(m >> 2) : ((64 - m) >> 2); }
}

/* rows that are L in x: filters.c:203-287 (downfilter53VI) */
static void ana_lrow(const int16_t *x, int n, int16_t *lo, int16_t *hi)
{
	const int h = n >> 1;
	int k, carry = 0;
	for (k = 0; k < h; k++) {
		const int r = tap5(x, n, k);
		const int16_t acc = (int16_t)(r + carry); /* the reference accumulates in the short output cell */
		lo[k] = (int16_t)rnd_half_away(acc, 6);
		carry = diffuse(r);
	}
	for (k = 0; k < h - 1; k++) hi[k] = (int16_t)rnd_half_away(pair_predict(x, k), 3);
	hi[h - 1] = (int16_t)((x[n - 1] - x[n - 2]) >> 3);
}

/* a3: wavelet_filterbank.c:52-302 */
void nhwo_analysis(int16_t *jpeg, int16_t *proc, int stride, int n, int final_level, int16_t *keep)
{
	const int h = n >> 1;
	int i, j;
	for (i = 0; i < n; i++) ana_raw(jpeg + i * stride, n, proc + i * stride, proc + i * stride + h); /* :71-75 */
	for (i = 0; i < n; i++)                                                                          /* :100-105 */
		for (j = 0; j < n; j++) jpeg[i * stride + j] = proc[j * stride + i];
	if (keep) memcpy(keep, jpeg, sizeof(int16_t) * 2 * NHWO_QSIZE);                                  /* :107-112 */
	for (i = 0; i < h; i++) ana_lrow(jpeg + i * stride, n, proc + i * stride, proc + i * stride + h); /* :118-125 */
	for (i = h; i < n; i++) ana_hrow(jpeg + i * stride, n, proc + i * stride, proc + i * stride + h); /* :147-154 */
	if (!final_level)                                                                                /* :172-184 */
		for (i = 0; i < h; i++)
			for (j = 0; j < h; j++) jpeg[i * stride + j] = proc[j * stride + i];
}

/* a7  1-D synthesis: filters.c:521-572 (upfilter53I, then III or VI accumulate on top) */
static void syn_row(const int16_t *lo, const int16_t *hi, int m, int16_t *out, int normalise)
{
	int k;
	for (k = 0; k < m; k++) {
		const int ln = (k + 1 < m) ? lo[k + 1] : lo[k];
		const int hp = k > 0 ? hi[k - 1] : hi[0];
		const int hn = (k + 1 < m) ? hi[k + 1] : hi[k];
		int16_t e = (int16_t)(lo[k] << 3);
		int16_t o = (int16_t)((lo[k] + ln) << 2);
		e = (int16_t)(e - ((hi[k] + hp) << 1));
		o = (int16_t)(o + (6 * hi[k] - hp - hn));
		if (normalise) {                         /* upfilter53VI: filters.c:549-572 */
			if (e > 0) e = (int16_t)(e + 32);
			e >>= 6;
			if (o > 0) o = (int16_t)(o + 32);
			o >>= 6;
		}
		out[2 * k] = e;
		out[2 * k + 1] = o;
	}
}

/* wavelet_filterbank.c:305-496, as called by the encoder (last_stage = 0: result ends in jpeg) */
void nhwo_synthesis(int16_t *jpeg, int16_t *proc, int stride, int n)
{
	const int h = n >> 1;
	int i, j;
	for (i = 0; i < n; i++) syn_row(jpeg + i * stride, jpeg + i * stride + h, h, proc + i * stride, 0); /* :324-346 */
	for (i = 0; i < n; i++)                                                                            /* :354-359 */
		for (j = 0; j < n; j++) jpeg[i * stride + j] = proc[j * stride + i];
	for (i = 0; i < n; i++) syn_row(jpeg + i * stride, jpeg + i * stride + h, h, proc + i * stride, 1); /* :375-381 */
	for (i = 0; i < n; i++)                                                                            /* :390-395 */
		for (j = 0; j < n; j++) jpeg[i * stride + j] = proc[j * stride + i];
}

/* SURVEY.md section 8d generator (not part of the reference) */
void nhwo_synth_image(uint32_t seed, uint8_t *bgr)
{
	uint32_t x = 0x9E3779B9u * (seed + 1u);
	int lat[17][17][3];
	int gy, gx, c, yy, xx;
	if (!x) x = 1;
#define NHWO_NEXT() (x ^= x << 13, x ^= x >> 17, x ^= x << 5, x)
	for (gy = 0; gy < 17; gy++)
		for (gx = 0; gx < 17; gx++)
			for (c = 0; c < 3; c++) lat[gy][gx][c] = (int)(NHWO_NEXT() >> 24);
	for (yy = 0; yy < 512; yy++)
		for (xx = 0; xx < 512; xx++)
			for (c = 0; c < 3; c++) {
				const int cy = yy >> 5, wy = (yy & 31) << 3, cx = xx >> 5, wx = (xx & 31) << 3;
				const int v = ((lat[cy][cx][c] * (256 - wx) + lat[cy][cx + 1][c] * wx) * (256 - wy) +
				               (lat[cy + 1][cx][c] * (256 - wx) + lat[cy + 1][cx + 1][c] * wx) * wy + 32768) >> 16;
				int b = v + (int)(NHWO_NEXT() % 13u) - 6;
				*bgr++ = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
			}
#undef NHWO_NEXT
}
