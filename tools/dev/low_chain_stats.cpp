/* Developer tool (host): what the chain of the q <= 16 pre-filter's pair machine spends its steps on, counted with the kernel's own machine
 * text (nhwcodec_amd/csrc/nhw_low_machine.h) over whole images as ONE stream of 510 x 255 pair codes.
 * build: see tools/dev/low_chain_stats.sh      usage: low_chain_stats <q> <images> <class> */
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
	if (argc < 4) return 2;
	const int q = atoi(argv[1]), n = atoi(argv[2]), cls = atoi(argv[3]);
	const int S = 512, NP = 510 * 255;
	uint8_t *bgr = (uint8_t *)malloc(NHWO_IMG_BYTES), *u = (uint8_t *)malloc(65536), *v = (uint8_t *)malloc(65536);
	int16_t *y = (int16_t *)malloc(2 * S * S), *km = (int16_t *)malloc(2 * S * S);
	uint8_t *codes = (uint8_t *)malloc(NP + 64);
	int sharp, s2;
	lm_params(q, &sharp, &s2);
	long firsts = 0, firsts_slow = 0, in_burst_slow = 0, in_burst_fast = 0, pairs = 0;
	long burst_starts = 0, std_start = 0, quiet_start = 0, w8z_start = 0, t4nz = 0, t1n1 = 0;
	long ends_cap = 0, ends_wrap = 0, ends_slow = 0, burst_len = 0, gate_open_pairs = 0;
	long hist_t10[3] = { 0, 0, 0 }, why_hist[6] = { 0, 0, 0, 0, 0, 0 };
	const bool trace = argc > 4;
	for (int s = 0; s < n; s++) {
		make_image(cls, s, bgr);
		nhwo_color(bgr, q, y, u, v);
		memset(km, 0, 2 * S * S);
		lm_contrast_map(y, km, q);
		for (int r = 1; r < S - 1; r++) for (int p = 0; p < 255; p++) {
			const int k0 = km[r * S + 1 + 2 * p], k1 = km[r * S + 2 + 2 * p];
			codes[(r - 1) * 255 + p] = (uint8_t)((iabs(k0) > sharp) | ((iabs(k1) > sharp) << 1) | ((iabs(k1) > s2) << 2) | ((iabs(k0) > sharp + 96) << 3));
		}
		PfM m; PfC c;
		machine_reset(m); machine_cache(m, c);
		bool in_burst = false;
		for (int i = 0; i < NP; i++) {
			const int row = 1 + i / 255;
			pairs++;
			const bool first = m.t[1] == 0;
			if (!first && !in_burst) {
				/* a burst begins (the pair behind a first pair, or behind a machine_step that left t1 != 0) */
				in_burst = true; burst_starts++;
				const bool quiet = burst_quiet(m, c), entry = burst_entry_ok(m, c);
				if (quiet) quiet_start++;
				if (c.w8z) w8z_start++;
				if (m.t[4] != 0) t4nz++;
				if (m.t[1] != 1) t1n1++;
				if (quiet && entry && !c.w8z && m.t[4] == 0 && m.t[1] == 1) std_start++;
				hist_t10[m.t[10] == 10 ? 0 : m.t[10] == 8 ? 1 : 2]++;
			}
			if (!first && !burst_quiet(m, c)) gate_open_pairs++;
			int a = machine_step_fast(m, c, codes[i]);
			if (a < 0 && !first) {                                      /* why is this pair machine_step's?  (the tests of machine_burst_fast, in its order) */
				const int fires = (codes[i] & 1) + ((codes[i] >> 1) & 1);
				const int t1 = m.t[1] + fires, t4 = m.t[4] + fires;
				const int win = (t4 == m.t[10]) & (t1 == m.t[11]);
				const int cyc = (t4 >= 10) & ((t4 > 10) | (t1 != 15));
				const int t17 = cyc ? (m.t[18] == 0) : win;
				const int why = c.t6bad ? 0 : t17 ? (cyc ? 1 : 2) : t1 > 2000003 ? 3 : t1 >= 15 ? 4 : 5;
				why_hist[why]++;
				if (trace && s == 0 && row >= 200 && row < 203) fprintf(stderr, "row %d pair %d: why %d  t1 %d t4 %d t10 %d t11 %d t18 %d t44 %d t6 %d t7 %d t14 %d t15 %d t16 %d t24 %d t29 %d w3 %d\n", row, i % 255, why, m.t[1], m.t[4], m.t[10], m.t[11], m.t[18], m.t[44], m.t[6], m.t[7], m.t[14], m.t[15], m.t[16], m.t[24], m.t[29], m.w[3]);
			}
			if (a < 0) {
				a = machine_step(m, codes[i], row); machine_cache(m, c);
				if (first) firsts_slow++; else in_burst_slow++;
			} else if (!first) in_burst_fast++;
			if (first) firsts++;
			else burst_len++;
			if (m.t[1] == 0 && !first) { in_burst = false; }
		}
	}
	const double rows = 510.0 * n;
	printf("q%d class %d, %d images: per row: first pairs %.1f (slow %.2f), bursts %.1f (standard start %.1f, quiet %.1f, w8z %.1f, t4!=0 %.1f, t1!=1 %.1f), pairs in bursts %.1f (through machine_step %.2f), gate open at %.1f pairs; window at burst start 10/15: %.1f 8/12: %.1f other: %.1f\n",
	       q, cls, n, firsts / rows, firsts_slow / rows, burst_starts / rows, std_start / rows, quiet_start / rows, w8z_start / rows, t4nz / rows, t1n1 / rows,
	       burst_len / rows, in_burst_slow / rows, gate_open_pairs / rows, hist_t10[0] / rows, hist_t10[1] / rows, hist_t10[2] / rows);
	printf("   machine_step inside bursts, per row: t6 %.2f, t17 by t18 rotation %.2f, t17 by window %.2f, forced end %.2f, cap schedules %.2f, idle schedules / one-time / re-arm %.2f\n",
	       why_hist[0] / rows, why_hist[1] / rows, why_hist[2] / rows, why_hist[3] / rows, why_hist[4] / rows, why_hist[5] / rows);
	return 0;
}
