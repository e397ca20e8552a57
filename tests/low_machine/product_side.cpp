/* Test harness, product side (C++): nhwcodec_amd/csrc/nhw_low_machine.h compiled for the host -- the very text the HIP kernel runs --
 * walked over whole images next to the oracle's machine: machine_step must leave the counters exactly as the oracle's machine_pair
 * does, machine_step_fast must either decline (-1, counters untouched) or give machine_step's answer and counters.
 * usage: machine_check <q_first> <q_last> <images> <class>   (class 0: SURVEY 8d synthetic; 1: white noise; 2: synthetic + patches; 3: stripes) */
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
int lm_machine_size(void);
void lm_machine_reset(void *m);
void lm_params(int q, int *sharp, int *sharp2);
void lm_contrast_map(const int16_t *src, int16_t *km, int q);
void lm_machine_pair(void *m, int q, int row, int16_t *km, int16_t *y, uint8_t *so);
}
static int iabs(int v) { return v < 0 ? -v : v; }
static uint32_t rng_s;
static uint32_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 17; rng_s ^= rng_s << 5; return rng_s; }
static void make_image(int cls, int seed, uint8_t *bgr)
{
	rng_s = 0x9E3779B9u * (uint32_t)(seed + 1) + (uint32_t)cls * 7919u; if (!rng_s) rng_s = 1;
	if (cls == 1) { for (int i = 0; i < NHWO_IMG_BYTES; i++) bgr[i] = (uint8_t)(rnd() >> 24); return; }
	nhwo_synth_image((uint32_t)seed, bgr);
	if (cls == 2) {                     /* rectangles of flat colour, noise patches and dot grids over the synthetic image */
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
	} else if (cls == 3) {              /* gradient + stripes + light noise */
		for (int yy = 0; yy < 512; yy++) for (int xx = 0; xx < 512; xx++) for (int c = 0; c < 3; c++) {
			const int v = xx / 2 + yy / 3 + ((xx * (seed % 7 + 3) / 8) % 32) * 3 + (int)(rnd() % 9) - 4;
			bgr[(yy * 512 + xx) * 3 + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
		}
	}
}


/* the row walk of k_low_machine on the host: bursts through burst_lane / burst_commit (a loop over j stands in for the lanes and their
 * ballots), everything else pair by pair.  Returns the number of pairs that went through burst_commit. */
static int row_hop(PfM &m, PfC &c, const uint8_t *codes, uint8_t *acts, int row)
{
	int pos = 0, bursted = 0;
	int hits_prefix[256];                                               /* inclusive prefix sums of the pairs' hits */
	{ int h = 0; for (int i = 0; i < 255; i++) { h += (codes[i] & 1) + ((codes[i] >> 1) & 1); hits_prefix[i] = h; } hits_prefix[255] = h; }
	memset(acts, 0, 255);
	bool give_up = false;                                               /* a burst that was declined is walked pair by pair: to its end, or to the next pair machine_step takes */
	int cut_at = 255;                                                   /* where a burst's pairs end: the row's end, or the pair that ends the burst through t17 */
	while (pos < 255) {
		if (m.t[1] == 0) give_up = false;
		else if (!give_up && pos != cut_at && burst_entry_ok(m, c)) {
			PfBurstMasks k = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
			const int base = pos ? hits_prefix[pos - 1] : 0;
			for (int j = 0; j < 64; j++) {
				const int i = pos + j < 255 ? pos + j : 255;
				const PfBurstLane b = burst_lane(j, m.t[1], m.t[4], m.t[44], hits_prefix[i] - base, m.t[10], m.t[11], c.exT);
				const unsigned long long bit = 1ull << j;
				if (b.cap) k.cap |= bit; if (b.wrap) k.wrap |= bit; if (b.win) k.win |= bit; if (b.cyc) k.cyc |= bit; if (b.i6) k.i6 |= bit;
				if (b.iS) k.iS |= bit; if (b.cnt) k.cnt |= bit; if (b.g13) k.g13 |= bit; if (b.e15) k.e15 |= bit; if (b.eT) k.eT |= bit;
			}
			auto hits_to = [&](int e) { return hits_prefix[pos + e < 255 ? pos + e : 255] - base; };
			const int n = burst_quiet(m, c) ? burst_commit_quiet(m, c, (unsigned)k.cap, (unsigned)k.wrap, (unsigned)k.win, (unsigned)k.cyc, c.w8z ? (unsigned)k.i6 : 0u, cut_at - pos, hits_to)
			                                : burst_commit(m, c, k, cut_at - pos, hits_to);
			if (n > 0) { pos += n; bursted += n; continue; }
			if (cut_at == 255) {                                            /* declined for a pair that ends it through t17: taken up to that pair */
				const int w = burst_t17_pair(m, k.cap, k.wrap, k.win, k.cyc, 255 - pos);
				if (w >= 1 && w < 255 - pos) { cut_at = pos + w; continue; }
			}
			cut_at = 255;
			give_up = true;
		}
		int a = pos == cut_at ? -1 : machine_step_fast(m, c, codes[pos]);
		if (a < 0) { a = machine_step(m, codes[pos], row); machine_cache(m, c); give_up = false; cut_at = 255; }
		acts[pos++] = (uint8_t)a;
	}
	return bursted;
}

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s q_first q_last images class\n", argv[0]); return 2; }
	const int q0 = atoi(argv[1]), q1 = atoi(argv[2]), n = atoi(argv[3]), cls = atoi(argv[4]);
	if (lm_machine_size() != (int)sizeof(PfM)) { fprintf(stderr, "machine layouts differ\n"); return 2; }
	const int S = 512;
	uint8_t *bgr = (uint8_t *)malloc(NHWO_IMG_BYTES), *u = (uint8_t *)malloc(65536), *v = (uint8_t *)malloc(65536), *so = (uint8_t *)malloc(S * S);
	int16_t *y = (int16_t *)malloc(2 * S * S), *src = (int16_t *)malloc(2 * S * S), *km = (int16_t *)malloc(2 * S * S);
	long bad = 0;
	for (int q = q0; q <= q1; q++) {
		long steps = 0, fast = 0, bad_fast = 0, bad_step = 0, bad_hop = 0, bursted = 0;
		int sharp, s2;
		lm_params(q, &sharp, &s2);
		for (int s = 0; s < n; s++) {
			make_image(cls, s, bgr);
			nhwo_color(bgr, q, y, u, v);
			memcpy(src, y, 2 * S * S); memset(km, 0, 2 * S * S); memset(so, 0, S * S);
			lm_contrast_map(src, km, q);
			PfM ref, mach, trial, hop; PfC cache, hcache;
			uint8_t codes[256], acts_ref[256], acts_hop[256];
			lm_machine_reset(&ref); machine_reset(mach); machine_cache(mach, cache); machine_reset(hop); machine_cache(hop, hcache);
			if (memcmp(&ref, &mach, sizeof ref)) { fprintf(stderr, "reset states differ\n"); return 1; }
			for (int r = 1; r < S - 1; r++) {
			for (int c = 1; c < S - 2; c += 2) {
				const int at = r * S + c, k0 = km[at], k1 = km[at + 1];
				const int code = (iabs(k0) > sharp) | ((iabs(k1) > sharp) << 1) | ((iabs(k1) > s2) << 2) | ((iabs(k0) > sharp + 96) << 3);
				codes[(c - 1) / 2] = (uint8_t)code;
				lm_machine_pair(&ref, q, r, km + at, y + at, so + at);
				trial = mach;
				const int act = machine_step(mach, code, r);
				acts_ref[(c - 1) / 2] = (uint8_t)act;
				if (memcmp(&ref, &mach, sizeof ref)) { if (bad_step++ < 3) fprintf(stderr, "q%d image %d row %d col %d: machine_step leaves other counters than the oracle\n", q, s, r, c); mach = ref; }
				const PfM before = trial;
				const int fa = machine_step_fast(trial, cache, code);
				steps++;
				if (fa >= 0) {
					fast++;
					if (fa != act || memcmp(&trial, &mach, sizeof mach)) { if (bad_fast++ < 3) fprintf(stderr, "q%d image %d row %d col %d: machine_step_fast differs (answer %d, machine_step %d)\n", q, s, r, c, fa, act); }
				} else {
					if (memcmp(&trial, &before, sizeof before)) { if (bad_fast++ < 3) fprintf(stderr, "q%d image %d row %d col %d: machine_step_fast declined but touched the counters\n", q, s, r, c); }
					machine_cache(mach, cache);
				}
			}
			bursted += row_hop(hop, hcache, codes, acts_hop, r);
			if (memcmp(acts_ref, acts_hop, 255) || memcmp(&hop, &mach, sizeof mach)) {
				if (bad_hop++ < 3) { fprintf(stderr, "q%d image %d row %d: the burst walk differs from the pair-by-pair walk (answers %s)\n", q, s, r, memcmp(acts_ref, acts_hop, 255) ? "differ" : "equal");
					for (int i = 0; i < 45; i++) if (hop.t[i] != mach.t[i]) fprintf(stderr, "   t%d: burst walk %d, pairs %d\n", i, hop.t[i], mach.t[i]); }
				hop = mach; machine_cache(hop, hcache);
			}
			}
		}
		printf("q%d class %d: %ld pairs, fast form %.1f %%, in whole bursts %.1f %%, fast mismatches %ld, step mismatches %ld, burst-walk mismatches %ld\n", q, cls, steps, 100.0 * fast / steps, 100.0 * bursted / steps, bad_fast, bad_step, bad_hop);
		bad += bad_fast + bad_step + bad_hop;
	}
	return bad ? 1 : 0;
}
