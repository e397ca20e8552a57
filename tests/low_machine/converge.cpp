/* Developer measurement (host): how fast does the pair machine of the quality 1..16 pre-filter (nhw_low_machine.h: t1..t44, w1..w8) forget its
 * past?  The machine is walked over whole images; then, for start rows r0 = 32, 64 .. 480, a second machine is started FROM RESET k rows
 * earlier and walked to r0: does it hold the true state there?  The smallest such k of {1, 2, 4, 8, 16, 32, 64} is recorded (once the two
 * agree they agree for good: same state, same input).  If a few rows were enough, several wavefronts could walk blocks of rows of one image
 * speculatively and be checked at the seams (k_low_machine is bound by the latency of ONE wavefront walking an image).
 * usage: converge <q_first> <q_last> <images> <class>      (classes as in product_side.cpp) */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define DEVI static inline
#define DEVN static
#define Q 65536
#include "../../nhwcodec_amd/csrc/nhw_low_machine.h"
extern "C" {
#include "../../oracle/nhwo.h"
void lm_params(int q, int *sharp, int *sharp2);
void lm_contrast_map(const int16_t *src, int16_t *km, int q);
}
static int iabs(int v) { return v < 0 ? -v : v; }
static uint32_t rng_s;
static uint32_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 17; rng_s ^= rng_s << 5; return rng_s; }
static void make_image(int cls, int seed, uint8_t *bgr)
{
	rng_s = 0x9E3779B9u * (uint32_t)(seed + 1) + (uint32_t)cls * 7919u; if (!rng_s) rng_s = 1;
	if (cls == 1) { for (int i = 0; i < NHWO_IMG_BYTES; i++) bgr[i] = (uint8_t)(rnd() >> 24); return; }
	nhwo_synth_image((uint32_t)seed, bgr);
	if (cls == 2) {
		for (int k = 0; k < 60; k++) {
			const int y0 = rnd() % 480, x0 = rnd() % 480, hh = 2 + rnd() % 120, ww = 2 + rnd() % 120, kind = rnd() % 3;
			const uint8_t col[3] = { (uint8_t)(rnd() >> 24), (uint8_t)(rnd() >> 24), (uint8_t)(rnd() >> 24) };
			for (int yy = y0; yy < y0 + hh && yy < 512; yy++) for (int xx = x0; xx < x0 + ww && xx < 512; xx++) for (int c = 0; c < 3; c++) {
				uint8_t *p = bgr + (yy * 512 + xx) * 3 + c;
				if (kind == 0) *p = col[c];
				else if (kind == 1) { const int v = *p + (int)(rnd() % 81) - 40; *p = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
				else if (((xx + yy) & 3) == 0) *p = col[c];
			}
		}
	}
}
int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s q_first q_last images class\n", argv[0]); return 2; }
	const int q0 = atoi(argv[1]), q1 = atoi(argv[2]), n = atoi(argv[3]), cls = atoi(argv[4]);
	const int S = 512;
	uint8_t *bgr = (uint8_t *)malloc(NHWO_IMG_BYTES), *u = (uint8_t *)malloc(65536), *v = (uint8_t *)malloc(65536);
	int16_t *y = (int16_t *)malloc(2 * S * S), *km = (int16_t *)malloc(2 * S * S);
	static uint8_t codes[512][256];
	static PfM truth[512];
	static const int ks[7] = { 1, 2, 4, 8, 16, 32, 64 };
	for (int q = q0; q <= q1; q++) {
		int sharp, s2;
		lm_params(q, &sharp, &s2);
		long hist[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, starts = 0, differing_fields = 0, never = 0;
		for (int s = 0; s < n; s++) {
			make_image(cls, s, bgr);
			nhwo_color(bgr, q, y, u, v);
			memset(km, 0, 2 * S * S);
			lm_contrast_map(y, km, q);
			for (int r = 1; r < S - 1; r++)
				for (int c = 1; c < S - 2; c += 2) {
					const int at = r * S + c, k0 = km[at], k1 = km[at + 1];
					codes[r][(c - 1) / 2] = (uint8_t)((iabs(k0) > sharp) | ((iabs(k1) > sharp) << 1) | ((iabs(k1) > s2) << 2) | ((iabs(k0) > sharp + 96) << 3));
				}
			PfM m; machine_reset(m);
			for (int r = 1; r < S - 1; r++) { truth[r] = m; for (int i = 0; i < 255; i++) machine_step(m, codes[r][i], r); }
			for (int r0 = 32; r0 <= 480; r0 += 32) {
				starts++;
				int found = 7;
				for (int ki = 0; ki < 7; ki++) {
					const int from = r0 - ks[ki];
					if (from < 1) break;
					PfM t; machine_reset(t);
					for (int r = from; r < r0; r++) for (int i = 0; i < 255; i++) machine_step(t, codes[r][i], r);
					if (!memcmp(&t, &truth[r0], sizeof t)) { found = ki; break; }
					if (ki == 6 || r0 - ks[ki + 1] < 1) { for (int f = 0; f < 45; f++) differing_fields += t.t[f] != truth[r0].t[f]; never++; }
				}
				hist[found]++;
			}
		}
		printf("q%-2d class %d: %ld seams; a machine started from reset k rows early holds the true state at the seam for k = 1: %ld, 2: %ld, 4: %ld, 8: %ld, 16: %ld, 32: %ld, 64: %ld, not within 64 (or the image's top): %ld",
		       q, cls, starts, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
		if (never) printf("  (counters t1..t44 still differing there: %.1f on average)", (double)differing_fields / never);
		printf("\n");
	}
	return 0;
}
