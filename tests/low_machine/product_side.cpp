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
static long g_cov[64];                 /* the probes in the schedule branches no picture has been seen to reach (PF_COV in the header) */
#define PF_COV(n) (g_cov[n]++)
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

/* the burst table of a picture's code stream (k_low_table on the host): table_entries with the hits' prefix sums behind its two accessors */
static void table_build(const uint8_t *codes, int np, uint16_t *tab)
{
	int *hp = (int *)malloc(sizeof(int) * (np + 80));                   /* hp[i + 1]: hits of pairs 0 .. i */
	hp[0] = 0;
	for (int i = 0; i < np + 79; i++) hp[i + 1] = hp[i] + (i < np ? (codes[i] & 1) + ((codes[i] >> 1) & 1) : 0);
	for (int p = 0; p < np; p++) {
		const int s = p + 1;
		auto g = [&](int j) { return hp[s + j + 1] - hp[s]; };
		auto first_ge = [&](int K) { int j = 0; while (j < 32 && g(j) < K) j++; return j; };
		unsigned out[4];
		table_entries(g, first_ge, np - s, codes[p], out);
		for (int v = 0; v < 4; v++) tab[4 * p + v] = (uint16_t)out[v];
	}
	free(hp);
}
/* every entry against the burst walked pair by pair with burst_lane_d; returns the number of entries that differ */
static long table_check(const uint8_t *codes, int np, const uint16_t *tab)
{
	long bad = 0;
	int *hp = (int *)malloc(sizeof(int) * (np + 80));
	hp[0] = 0;
	for (int i = 0; i < np + 79; i++) hp[i + 1] = hp[i] + (i < np ? (codes[i] & 1) + ((codes[i] >> 1) & 1) : 0);
	for (int p = 0; p < np; p++) for (int v = 0; v < 4; v++) {
		const int s = p + 1;
		int e = -1, ncyc = 0, win = 0, i6 = 0, cap = 0, ge = 0;
		for (int j = 0; j < 31; j++) {
			const int hits = hp[s + j + 1] - hp[s];
			const PfLaneD d = burst_lane_d(j, 1, 0, v, hits, 8, 12);
			ncyc += d.cyc; win |= d.win;
			if (d.end) { e = j; cap = d.cap; ge = hits; if (!d.cap) i6 |= d.i6; break; }
			i6 |= d.i6;
		}
		unsigned want;
		if (e < 0 || e >= np - s || ncyc > 7 || ((codes[p] & 9) == 9)) want = TAB_NONE;
		else want = (unsigned)e | ((unsigned)cap << 5) | ((unsigned)ncyc << 7) | ((unsigned)win << 10) | ((unsigned)i6 << 11);
		if (cap && !ge) want = 0xFFFFu;                                 /* (a cap without a hit cannot be: the burst would have wrapped at the pair before) */
		const unsigned got = tab[4 * p + v] & 0xFFFu;
		const bool same = want == TAB_NONE ? (got & TAB_NONE) != 0 : got == want;
		if (!same || (tab[4 * p + v] >> 12) != codes[p]) { if (bad++ < 5) fprintf(stderr, "table entry of pair %d, v %d: %03x, the burst walked pair by pair says %03x\n", p, v, got, want); }
	}
	free(hp);
	return bad;
}
/* the chain's walk with the table (k_low_chain on the host): a first pair and its burst through table_take, the rest as in stream_hop */
static long table_hop(PfM &m, PfC &c, const uint8_t *codes, const uint16_t *tab, uint8_t *acts, int np)
{
	long fast = 0;
	int pos = 0;
	int *hp = (int *)malloc(sizeof(int) * (np + 64));
	{ int h = 0; for (int i = 0; i < np + 64; i++) { if (i < np) h += (codes[i] & 1) + ((codes[i] >> 1) & 1); hp[i] = h; } }
	memset(acts, 0, np);
	long gen_pairs = 0;
	while (pos < np) {
		if (m.t[1] == 0) {
			{
				int pairs = 0;
				const unsigned flags = tab_flags(m, c);
				const int a = table_take(m, flags, tab[4 * pos + (m.t[44] & 3)], pairs);
				if (a >= 0) { acts[pos] = (uint8_t)a; pos += pairs; fast += pairs; continue; }
			}
		}
		else if (burst_entry_ok(m, c)) {
			/* the burst's longest clean prefix, decided "in the lanes" (a loop over j stands in for them, running counts for v_mbcnt) */
			const int base = pos ? hp[pos - 1] : 0;
			PfGenU g = { m.t[1], m.t[4], m.t[44], m.t[10], m.t[11], m.t[18], m.t[29] > 0, m.t[30], m.t[33], np - pos };
			int cyc_before = 0, cnt_before = 0, s = -1; unsigned w = 0;
			for (int j = 0; j < 64 && s < 0; j++) {
				const int hits = hp[pos + j] - base, prev = j ? hp[pos + j - 1] - base : 0;
				const PfGen1 d = gen_lane1(j, g, c, hits, hits - prev);
				const PfGen2 r = gen_lane2(j, g, c, d, cyc_before, cnt_before);
				if (r.stop) { s = j; w = gen_word(g, d, r); }
				cyc_before += d.cyc; cnt_before += d.counting;
			}
			const int n = gen_take(m, s, w);
			pos += n; gen_pairs += n;
			if (!(w & 1u)) continue;                                        /* a clean run: the burst is over, or goes on behind the lanes' reach */
			const int a = machine_step(m, codes[pos], 1 + pos / 255);       /* the pair that stopped it */
			machine_cache(m, c);
			acts[pos++] = (uint8_t)a;
			continue;
		}
		int a = machine_step_fast(m, c, codes[pos]);
		if (a < 0) { a = machine_step(m, codes[pos], 1 + pos / 255); machine_cache(m, c); }
		acts[pos++] = (uint8_t)a;
	}
	free(hp);
	return fast;
}

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s q_first q_last images class\n", argv[0]); return 2; }
	const int q0 = atoi(argv[1]), q1 = atoi(argv[2]), n = atoi(argv[3]), cls = atoi(argv[4]);
	if (lm_machine_size() != (int)sizeof(PfM)) { fprintf(stderr, "machine layouts differ\n"); return 2; }
	for (int idx = 0; idx < 512; idx++) for (int t14 = 0; t14 < 6; t14++) {  /* first_lut against machine_first_fast: every index, every t14 it stands for */
		PfM m; PfC c;
		machine_reset(m);
		m.t[3] = (idx >> 4) & 3; m.t[8] = ((idx >> 6) & 1) ? 1 : 2; m.t[12] = (idx >> 7) & 1; m.t[14] = t14; m.t[13] = 0; m.t[27] = 5;
		machine_cache(m, c);
		if (c.t14_045 != ((idx >> 8) & 1) || c.fb14) continue;
		const int a = machine_first_fast(m, c, idx & 15);
		const unsigned lv = first_lut(idx);
		if ((idx & 9) == 9) { if (a != -1) { fprintf(stderr, "first_lut: index %d should decline\n", idx); return 1; } continue; }
		if (a != (int)(lv & 7u) || m.t[3] != (int)((lv >> 4) & 3u)) { fprintf(stderr, "first_lut differs from machine_first_fast at index %d (t14 %d)\n", idx, t14); return 1; }
	}
	const int S = 512;
	uint8_t *bgr = (uint8_t *)malloc(NHWO_IMG_BYTES), *u = (uint8_t *)malloc(65536), *v = (uint8_t *)malloc(65536), *so = (uint8_t *)malloc(S * S);
	int16_t *y = (int16_t *)malloc(2 * S * S), *src = (int16_t *)malloc(2 * S * S), *km = (int16_t *)malloc(2 * S * S);
	long bad = 0;
	for (int q = q0; q <= q1; q++) {
		long steps = 0, fast = 0, bad_fast = 0, bad_step = 0, bad_hop = 0, bursted = 0, tabled = 0;
		static uint16_t s_tab[4 * 510 * 255];
		static uint8_t s_codes[510 * 255], s_acts_ref[510 * 255], s_acts[510 * 255];
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
			memcpy(s_codes + (r - 1) * 255, codes, 255); memcpy(s_acts_ref + (r - 1) * 255, acts_ref, 255);
			bursted += row_hop(hop, hcache, codes, acts_hop, r);
			if (memcmp(acts_ref, acts_hop, 255) || memcmp(&hop, &mach, sizeof mach)) {
				if (bad_hop++ < 3) { fprintf(stderr, "q%d image %d row %d: the burst walk differs from the pair-by-pair walk (answers %s)\n", q, s, r, memcmp(acts_ref, acts_hop, 255) ? "differ" : "equal");
					for (int i = 0; i < 45; i++) if (hop.t[i] != mach.t[i]) fprintf(stderr, "   t%d: burst walk %d, pairs %d\n", i, hop.t[i], mach.t[i]); }
				hop = mach; machine_cache(hop, hcache);
			}
			}
			{                                                               /* the whole picture again as one stream with the burst table, the way k_low_chain walks it */
				PfM sm; PfC sc;
				machine_reset(sm); machine_cache(sm, sc);
				table_build(s_codes, 510 * 255, s_tab);
				bad_hop += table_check(s_codes, 510 * 255, s_tab);
				tabled += table_hop(sm, sc, s_codes, s_tab, s_acts, 510 * 255);
				if (memcmp(s_acts, s_acts_ref, 510 * 255) || memcmp(&sm, &mach, sizeof mach)) {
					if (bad_hop++ < 3) { int i = 0; while (i < 510 * 255 && s_acts[i] == s_acts_ref[i]) i++;
						fprintf(stderr, "q%d image %d: the table walk differs from the pair-by-pair walk (first answer that differs: pair %d of %d)\n", q, s, i, 510 * 255); }
				}
			}
		}
		printf("q%d class %d: %ld pairs, fast form %.1f %%, in whole bursts %.1f %%, through the burst table %.1f %%, fast mismatches %ld, step mismatches %ld, burst-walk mismatches %ld\n", q, cls, steps, 100.0 * fast / steps, 100.0 * bursted / steps, 100.0 * tabled / steps, bad_fast, bad_step, bad_hop);
		bad += bad_fast + bad_step + bad_hop;
	}
	{ long hit = 0; for (int i = 0; i < 64; i++) hit += g_cov[i] != 0;
	  printf("schedule probes reached: %ld%s\n", hit, hit ? "  (a picture that reaches the late schedules: pin it against oracle/_ref and add it to the goldens)" : ""); }
	return bad ? 1 : 0;
}
