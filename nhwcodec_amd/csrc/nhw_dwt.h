/*
 * nhw_dwt.h -- what the block-resident filterbank kernels share (nhw_front.hip: k_dwt_ana / k_dwt_syn; nhw_tail.hip: k_l2_recon).
 */
#ifndef NHW_DWT_H
#define NHW_DWT_H
#include <hip/hip_runtime.h>
#include <stdint.h>

/* workgroup barrier that orders LDS traffic only: global loads stay in flight across it */
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* one output pair of the 5/3 synthesis (wavelet_filterbank.c:305-496; upfilter53I / III / VI, filters.c:521-572): x holds the low band in cells 0 .. S/2-1 and the
 * high band behind it, st apart; the second direction normalises */
template <int S>
__device__ __forceinline__ void syn_pair(const int16_t *x, int st, int k, bool normalise, int *e_out, int *o_out)
{
	constexpr int M = S / 2;
	const int16_t *lo = x, *hi = x + M * st;
	const int l0 = lo[k * st], ln = (k + 1 < M) ? lo[(k + 1) * st] : l0;
	const int h0 = hi[k * st], hp = k > 0 ? hi[(k - 1) * st] : hi[0], hn = (k + 1 < M) ? hi[(k + 1) * st] : h0;
	int16_t e = (int16_t)(l0 << 3);
	int16_t o = (int16_t)((l0 + ln) << 2);
	e = (int16_t)(e - ((h0 + hp) << 1));
	o = (int16_t)(o + (6 * h0 - hp - hn));
	if (normalise) {
		if (e > 0) e = (int16_t)(e + 32);
		e >>= 6;
		if (o > 0) o = (int16_t)(o + 32);
		o >>= 6;
	}
	*e_out = e; *o_out = o;
}

/* rounding of the analysis' second direction (filters.c:88-287) */
__device__ __forceinline__ int rnd_half_away(int v, int shift)
{
	/* v < 0: -((-v + half) >> shift) = ceil((v - half) / 2^shift) = (v + half - 1) >> shift -- no branch either way */
	return (v + (1 << (shift - 1)) + (v >> 31)) >> shift;
}
__device__ __forceinline__ int diffuse(int r)
{
	/* an odd function of r: |r| mod 64 read as a signed 6-bit number, divided by 4 towards zero, with the sign of r */
	const int s = r >> 31, a = (r ^ s) - s;
	const int t = (int)((unsigned)a << 26) >> 26;
	const int d = (t + ((t >> 31) & 3)) >> 2;
	return (d ^ s) - s;
}

/* the 5-tap low-pass numerator and the predicted odd sample of the analysis (filters.c:55-114, 203-287, 346-386), on cells st apart */
template <int S>
__device__ __forceinline__ int tap5s(const int16_t *x, int st, int k)
{
	const int c = 2 * k;
	const int l1 = c >= 1 ? x[(c - 1) * st] : x[st], l2 = c >= 2 ? x[(c - 2) * st] : x[2 * st];
	const int r1 = x[(c + 1) * st], r2 = (c + 2 < S) ? x[(c + 2) * st] : x[(S - 2) * st];
	return 6 * x[c * st] + 2 * (l1 + r1) - (l2 + r2);
}
__device__ __forceinline__ int pair_predict_s(const int16_t *x, int st, int k)
{
	int a = x[2 * k * st] + x[(2 * k + 2) * st];
	if ((k & 1) && (a & 1) && ((x[(2 * k - 2) * st] + x[2 * k * st]) & 1)) a++;
	return x[(2 * k + 1) * st] - (a >> 1);
}

#endif
