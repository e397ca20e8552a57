// Developer tool (host): a coverage-guided search over CODE streams (any sequence of the 16 pair codes, a superset of what pictures produce): does the
// q <= 16 pair machine ever reach the schedule branches no test picture reaches (PF_COV probes), or a burst end between the two one-time w8 steps?
// build: g++ -O2 -std=c++17 -o /tmp/codestream_search tools/dev/codestream_search.cpp     usage: codestream_search <seconds> <seed>
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
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
struct Seg { int len; uint8_t p0, p1, pg, ph; int mode; };   // mode 0: random with probabilities, 1: constant code, 2: alternating two codes
typedef std::vector<Seg> Genome;
static const int NP = 510 * 255;
static Seg rseg() { Seg s; s.len = 1 << (rnd() % 13); if (rnd() & 1) s.len = 1 + rnd() % 40; s.p0 = rnd(); s.p1 = rnd(); s.pg = rnd(); s.ph = (rnd() & 3) ? 0 : rnd(); s.mode = rnd() % 3; return s; }
struct Res { int probes; long prog; bool straddle; long cov[64]; };
static Res eval(const Genome &g, uint64_t seed)
{
	memset(g_cov, 0, sizeof g_cov);
	PfM m; machine_reset(m);
	uint64_t r = seed; auto rr = [&]() { r ^= r << 13; r ^= r >> 7; r ^= r << 17; return (uint32_t)(r >> 24) & 255; };
	size_t si = 0; int left = g.empty() ? NP : g[0].len; long prog = 0; bool straddle = false; int mx8 = 0, mx31 = 0, mx5 = 0;
	for (int i = 0; i < NP; i++) {
		while (left <= 0 && si + 1 < g.size()) { si++; left = g[si].len; }
		const Seg &s = g.empty() ? Seg{NP, 0, 0, 0, 0, 0} : g[si]; left--;
		int code;
		if (s.mode == 1) code = s.p0 & 15; else if (s.mode == 2) code = (i & 1) ? (s.p0 & 15) : (s.p1 & 15);
		else { const int f0 = rr() < s.p0, f1 = rr() < s.p1, g1 = f1 && rr() < s.pg, h0 = f0 && rr() < s.ph; code = f0 | (f1 << 1) | (g1 << 2) | (h0 << 3); }
		if ((code & 4) && !(code & 2)) code &= ~4; if ((code & 8) && !(code & 1)) code &= ~8;
		const int t44b = m.t[44], t1b = m.t[1], t14b = m.t[14];
		machine_step(m, code, 1 + i / 255);
		if (t44b < -90000 && m.t[44] < -90000) straddle = true;          // the pair behind the first W8 step did not take the second
		mx8 = std::max(mx8, m.t[8]); mx31 = std::max(mx31, m.t[31]); mx5 = std::max(mx5, m.t[5]);
		(void)t1b; (void)t14b;
	}
	Res R; R.probes = 0; for (int i = 0; i < 64; i++) { R.cov[i] = g_cov[i]; if (g_cov[i]) R.probes++; }
	R.prog = 100L * mx8 + 1000L * mx31 + 10L * mx5 + (m.w[8] ? 5 : 0) + (straddle ? 100000 : 0); R.straddle = straddle;
	return R;
}
int main(int argc, char **argv)
{
	const int secs = argc > 1 ? atoi(argv[1]) : 60; rs ^= (uint64_t)(argc > 2 ? atoi(argv[2]) : 1) * 0x9E3779B97F4A7C15ull;
	std::vector<std::pair<Genome, long>> pool;
	long evals = 0, seen[64] = { 0 }; bool any_straddle = false; int maxt8 = 0;
	const time_t end = time(nullptr) + secs;
	while (time(nullptr) < end) {
		Genome g;
		if (pool.size() < 16 || rnd() % 8 == 0) { const int n = 1 + rnd() % 60; for (int i = 0; i < n; i++) g.push_back(rseg()); }
		else { g = pool[rnd() % pool.size()].first; const int nm = 1 + rnd() % 3;
			for (int k = 0; k < nm; k++) { const int op = rnd() % 5; if (g.empty()) { g.push_back(rseg()); continue; }
				const size_t at = rnd() % g.size();
				if (op == 0) g[at] = rseg(); else if (op == 1) g.insert(g.begin() + at, rseg()); else if (op == 2 && g.size() > 1) g.erase(g.begin() + at);
				else if (op == 3) g[at].len = std::max(1, g[at].len + (int)(rnd() % 21) - 10); else { g[at].p0 += rnd() % 9 - 4; g[at].p1 += rnd() % 9 - 4; } } }
		const Res R = eval(g, rnd() | 1); evals++;
		bool fresh = false; for (int i = 0; i < 64; i++) if (R.cov[i] && !seen[i]) { seen[i] = 1; fresh = true; printf("[%ld] probe %d reached\n", evals, i); fflush(stdout); }
		if (R.straddle && !any_straddle) { any_straddle = true; printf("[%ld] straddle reached\n", evals); fflush(stdout); }
		const long score = 1000000L * R.probes + R.prog;
		if (pool.size() < 32) pool.push_back({ g, score });
		else { size_t w = 0; for (size_t i = 1; i < pool.size(); i++) if (pool[i].second < pool[w].second) w = i; if (score > pool[w].second || fresh) pool[w] = { g, score }; }
		(void)maxt8;
	}
	int tot = 0; for (int i = 0; i < 64; i++) tot += seen[i] != 0;
	printf("%ld code streams: %d probes reached, straddle %s, best score %ld\n", evals, tot, any_straddle ? "REACHED" : "never", pool.empty() ? 0 : std::max_element(pool.begin(), pool.end(), [](auto &a, auto &b) { return a.second < b.second; })->second);
}
