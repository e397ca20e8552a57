/* Developer tool (host): a coverage-guided search for pictures that drive the quality 1..16 pair machine into the schedule branches no
 * picture of the test classes reaches (oracle/nhwo_prelow.c:246-298, 410-424 = reference encoder/image_processing.c:1504-1873, 1875-1900).
 * The pictures are grey, built of horizontal bands of identical rows (a row = a sequence of runs: flat, impulse, step, ramp), so a band's
 * rows repeat one code sequence; fitness = the probes (PF_COV in nhw_low_machine.h) a picture's walk hits, then how far the counters that
 * lead there (t32, t36, t28, t8, t5) got.  Found pictures are written as raw 512 x 512 grey planes.
 * build: g++ -O2 -std=c++17 -o /tmp/prelow_fuzz tools/dev/prelow_fuzz.cpp <oracle_side.o> -Loracle -l:liboracle.so
 * usage: prelow_fuzz <quality> <seconds> <seed> <outdir> */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>
#include <algorithm>
#define DEVI static inline
#define DEVN static
#define Q 65536
static long g_cov[64];
#define PF_COV(n) (g_cov[n]++)
#include "../../nhwcodec_amd/csrc/nhw_low_machine.h"
extern "C" {
#include "../../oracle/nhwo.h"
void lm_params(int q, int *sharp, int *sharp2);
void lm_contrast_map(const int16_t *src, int16_t *km, int q);
}
static int iabs(int v) { return v < 0 ? -v : v; }
static uint64_t rs;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
static int rint_(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

struct Run { int kind, len, a, b; };                                   /* 0 flat a; 1 impulses of amplitude b every a-th pixel... see paint */
struct Band { int height; int base; std::vector<Run> runs; };
struct Rect { int x, y, w, h, kind, level, amp; };                       /* kind 0 flat, 1 noise of +-amp, 2 dots every 4th pixel, 3 vertical stripes of period amp, 4 horizontal ramp */
struct Pic { std::vector<Band> bands; int synth_seed; std::vector<Rect> rects; };

static void paint_row(const Band &bd, uint8_t *row)
{
	int x = 0, level = bd.base;
	for (const Run &r : bd.runs) {
		for (int i = 0; i < r.len && x < 512; i++, x++) {
			int v = level;
			switch (r.kind) {
			case 0: v = level; break;                                       /* flat */
			case 1: v = level + ((i % (r.a < 2 ? 2 : r.a)) == 0 ? r.b : 0); break;   /* impulses of height b every a pixels */
			case 2: v = level + (i * r.b) / (r.len > 1 ? r.len - 1 : 1); break;      /* ramp by b over the run */
			case 3: v = level + (((i / (r.a < 1 ? 1 : r.a)) & 1) ? r.b : 0); break;  /* square wave of period 2a, height b */
			case 4: v = level + ((i & 1) ? r.b : -r.b); break;             /* checker along x */
			}
			row[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
		}
		if (r.kind == 2) level += r.b;
		if (r.kind == 0 && r.b) level = r.a;                               /* a flat run may set a new level */
	}
	for (; x < 512; x++) row[x] = (uint8_t)(level < 0 ? 0 : level > 255 ? 255 : level);
}
static uint32_t hash32(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }
static void paint(const Pic &p, uint8_t *grey)
{
	if (p.synth_seed >= 0) {                                            /* the SURVEY 8d generator's picture (its green channel as grey), rectangles over it */
		static uint8_t bgr[3 * 512 * 512];
		nhwo_synth_image((uint32_t)p.synth_seed, bgr);
		for (int i = 0; i < 512 * 512; i++) grey[i] = bgr[3 * i + 1];
		for (const Rect &r : p.rects)
			for (int yy = r.y; yy < r.y + r.h && yy < 512; yy++) for (int xx = r.x; xx < r.x + r.w && xx < 512; xx++) {
				int v = r.level;
				if (r.kind == 1) v = grey[yy * 512 + xx] + (int)(hash32((uint32_t)(yy * 512 + xx) * 2654435761u + (uint32_t)r.amp) % (uint32_t)(2 * r.amp + 1)) - r.amp;
				else if (r.kind == 2) v = ((xx + yy) & 3) == 0 ? r.level : grey[yy * 512 + xx];
				else if (r.kind == 3) v = r.level + (((xx / (r.amp < 1 ? 1 : r.amp)) & 1) ? 24 : 0);
				else if (r.kind == 4) v = r.level + (xx - r.x) * r.amp / (r.w > 1 ? r.w : 1);
				grey[yy * 512 + xx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
			}
		return;
	}
	int y = 0;
	uint8_t row[512];
	for (const Band &b : p.bands) { paint_row(b, row); for (int i = 0; i < b.height && y < 512; i++, y++) memcpy(grey + 512 * y, row, 512); }
	for (; y < 512; y++) memcpy(grey + 512 * y, grey + 512 * (y ? y - 1 : 0), 512);
}
static Run rand_run()
{
	Run r;
	r.kind = rint_(0, 4); r.len = 1 << rint_(0, 7); if (rnd() & 1) r.len = rint_(1, 96);
	r.a = rint_(1, 12); r.b = rint_(-60, 60); if (r.kind == 0) { r.a = rint_(20, 235); r.b = rnd() & 1; }
	return r;
}
static Band rand_band()
{
	Band b; b.height = rnd() & 1 ? rint_(1, 12) : rint_(8, 160); b.base = rint_(40, 215);
	const int n = rint_(1, 14);
	for (int i = 0; i < n; i++) b.runs.push_back(rand_run());
	return b;
}
static Rect rand_rect() { Rect r; r.x = rint_(0, 500); r.y = rint_(0, 500); r.w = rint_(2, 300); r.h = rint_(1, 200); r.kind = rint_(0, 4); r.level = rint_(10, 245); r.amp = rint_(1, 40); return r; }
static Pic rand_pic()
{
	Pic p; p.synth_seed = -1;
	if (rnd() & 1) { p.synth_seed = rint_(0, 4000); const int n = rint_(0, 40); for (int i = 0; i < n; i++) p.rects.push_back(rand_rect()); return p; }
	const int n = rint_(1, 10); for (int i = 0; i < n; i++) p.bands.push_back(rand_band()); return p;
}
static void mutate(Pic &p)
{
	if (p.synth_seed >= 0) {
		const int k = rint_(0, 6);
		if (p.rects.empty() || k == 0) { p.rects.push_back(rand_rect()); return; }
		Rect &r = p.rects[rnd() % p.rects.size()];
		switch (k) {
		case 1: r.x = std::max(0, r.x + rint_(-9, 9)); r.y = std::max(0, r.y + rint_(-5, 5)); break;
		case 2: r.w = std::max(1, r.w + rint_(-12, 12)); r.h = std::max(1, r.h + rint_(-6, 6)); break;
		case 3: r.level = std::min(250, std::max(5, r.level + rint_(-10, 10))); r.amp = std::max(1, r.amp + rint_(-3, 3)); break;
		case 4: r = rand_rect(); break;
		case 5: if (p.rects.size() > 1) p.rects.erase(p.rects.begin() + rnd() % p.rects.size()); break;
		case 6: p.synth_seed = rint_(0, 4000); break;
		}
		return;
	}
	const int k = rint_(0, 9);
	if (p.bands.empty()) { p.bands.push_back(rand_band()); return; }
	Band &b = p.bands[rnd() % p.bands.size()];
	switch (k) {
	case 0: b.height = std::max(1, b.height + rint_(-8, 8)); break;
	case 1: b.base = std::min(235, std::max(20, b.base + rint_(-12, 12))); break;
	case 2: if (!b.runs.empty()) { Run &r = b.runs[rnd() % b.runs.size()]; r.b += rint_(-6, 6); } break;
	case 3: if (!b.runs.empty()) { Run &r = b.runs[rnd() % b.runs.size()]; r.len = std::max(1, r.len + rint_(-6, 6)); } break;
	case 4: if (!b.runs.empty()) { Run &r = b.runs[rnd() % b.runs.size()]; r.a = std::max(1, r.a + rint_(-2, 2)); } break;
	case 5: b.runs.insert(b.runs.begin() + (b.runs.empty() ? 0 : rnd() % b.runs.size()), rand_run()); break;
	case 6: if (b.runs.size() > 1) b.runs.erase(b.runs.begin() + rnd() % b.runs.size()); break;
	case 7: p.bands.insert(p.bands.begin() + rnd() % p.bands.size(), rand_band()); break;
	case 8: if (p.bands.size() > 1) p.bands.erase(p.bands.begin() + rnd() % p.bands.size()); break;
	case 9: if (!b.runs.empty()) b.runs[rnd() % b.runs.size()] = rand_run(); break;
	}
}

struct Score { int probes; long progress; };
static bool better(const Score &a, const Score &b) { return a.probes != b.probes ? a.probes > b.probes : a.progress > b.progress; }

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s quality seconds seed outdir\n", argv[0]); return 2; }
	const int q = atoi(argv[1]), secs = atoi(argv[2]); rs = 0x9E3779B97F4A7C15ull * (uint64_t)(atoi(argv[3]) + 1);
	const char *outdir = argv[4];
	const int S = 512, NP = 510 * 255;
	std::vector<uint8_t> grey(S * S), bgr(3 * S * S), u(65536), v(65536), codes(NP);
	std::vector<int16_t> y(S * S), km(S * S);
	int sharp, s2; lm_params(q, &sharp, &s2);
	long global_seen[64] = { 0 };
	auto eval = [&](const Pic &p, Score &sc, long cov[64]) {
		paint(p, grey.data());
		for (int i = 0; i < S * S; i++) { bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = grey[i]; }
		nhwo_color(bgr.data(), q, y.data(), u.data(), v.data());
		std::fill(km.begin(), km.end(), 0);
		lm_contrast_map(y.data(), km.data(), q);
		memset(g_cov, 0, sizeof g_cov);
		PfM m; machine_reset(m);
		int mx32 = 0, mx36 = 0, mx28 = 0, mx8 = 0, mx5 = 0, mx37 = 0, mx38 = 0, mx43 = 0, mx31 = 0, mx33 = 0, any5 = 0;
		for (int r = 1; r < S - 1; r++) for (int pp = 0; pp < 255; pp++) {
			const int k0 = km[r * S + 1 + 2 * pp], k1 = km[r * S + 2 + 2 * pp];
			const int code = (iabs(k0) > sharp) | ((iabs(k1) > sharp) << 1) | ((iabs(k1) > s2) << 2) | ((iabs(k0) > sharp + 96) << 3);
			machine_step(m, code, r);
			const int t32 = m.t[32] > 100 ? 9 : m.t[32];
			mx32 = std::max(mx32, t32); mx36 = std::max(mx36, m.t[36]); mx28 = std::max(mx28, m.t[28]); mx8 = std::max(mx8, m.t[8]); mx5 = std::max(mx5, m.t[5]);
			mx37 = std::max(mx37, m.t[37] < 0 ? 20 : m.t[37]); mx31 = std::max(mx31, m.t[31]); mx33 = std::max(mx33, m.t[33] > 0 ? 1 : 0); any5 |= m.t[14] == 5; mx38 = std::max(mx38, m.t[38]); mx43 = std::max(mx43, m.t[43]);
		}
		sc.probes = 0;
		for (int i = 0; i < 64; i++) { cov[i] = g_cov[i]; if (g_cov[i]) sc.probes++; }
		sc.progress = 5000L * mx31 + 20000L * mx33 + 20000L * any5 + 1000L * mx32 + 20L * std::min(mx36, 120) + 300L * mx28 + 2000L * std::min(mx8, 8) + 50L * std::min(mx5, 40) + 30L * std::min(mx37, 20) + 100L * std::min(mx38, 11) + 100L * std::min(mx43, 25);
	};
	struct Ent { Pic p; Score s; };
	std::vector<Ent> pool;
	const time_t t_end = time(nullptr) + secs;
	long evals = 0; int saved = 0;
	while (time(nullptr) < t_end) {
		Pic cand;
		if (pool.size() < 24 || rnd() % 16 == 0) cand = rand_pic();
		else { cand = pool[rnd() % pool.size()].p; const int nm = rint_(1, 4); for (int i = 0; i < nm; i++) mutate(cand); }
		Score sc; long cov[64];
		eval(cand, sc, cov);
		evals++;
		bool fresh = false;
		for (int i = 0; i < 64; i++) if (cov[i] && !global_seen[i]) { global_seen[i] = 1; fresh = true; }
		if (fresh) {
			char fn[256]; snprintf(fn, sizeof fn, "%s/q%d_%03d.grey", outdir, q, saved++);
			paint(cand, grey.data());
			FILE *f = fopen(fn, "wb"); if (f) { fwrite(grey.data(), 1, S * S, f); fclose(f); }
			printf("[%ld evals] %s: probes", evals, fn);
			for (int i = 0; i < 64; i++) if (cov[i]) printf(" %d", i);
			printf("  (progress %ld)\n", sc.progress); fflush(stdout);
		}
		if (pool.size() < 48) pool.push_back({ cand, sc });
		else {
			size_t worst = 0;
			for (size_t i = 1; i < pool.size(); i++) if (better(pool[worst].s, pool[i].s)) worst = i;
			if (better(sc, pool[worst].s) || fresh) pool[worst] = { cand, sc };
		}
	}
	int total = 0; for (int i = 0; i < 64; i++) total += global_seen[i] != 0;
	printf("q%d: %ld evaluations, %d probes reached:", q, evals, total);
	for (int i = 0; i < 64; i++) if (global_seen[i]) printf(" %d", i);
	printf("\n");
	return 0;
}
