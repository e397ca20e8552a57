// Host-side walk of the class tables of Y22 / Y23 (nhwcodec_amd/csrc/nhw_residual_rules.h, the text the HIP kernels compile) over their whole
// domains, against the comparison chains they are filled from (nhw_encoder.c:1077-1420 as restated in classify_kind / lh_rule / code_step_reg).
#include <cstdio>
#include <cstring>
#include "../../nhwcodec_amd/csrc/nhw_residual_rules.h"

int main()
{
	long checked = 0, bad = 0;
	for (int q = 13; q <= 23; q++) {
		const int rs = q >= 20 ? 3 : q >= 18 ? 4 : q >= 15 ? 6 : 8;   // nhw_encoder.c:1075-1079
		static uint8_t ktab[CK_TABLE_BYTES + 8], ytab[Y23_TAB_BYTES + 8];
		memset(ktab, 0xAA, sizeof ktab); memset(ytab, 0xAA, sizeof ytab);
		for (int tid = 0; tid < 7; tid++) { classify_table_fill(ktab, q, rs, tid, 7); code_table_fill(ytab, q, rs, tid, 7); }   // (a thread count that divides nothing)
		// Y22: the kind of every triple of differences
		for (int res = -40; res <= 40; res++) for (int a = -20; a <= 20; a++) for (int d2 = -20; d2 <= 20; d2++, checked++)
			if (classify_lookup(ktab, res, a, d2) != classify_kind(q, rs, res, a, d2)) { if (bad++ < 5) printf("q%d kind(%d,%d,%d)\n", q, res, a, d2); }
		// Y22: what a rule does to an LH1 coefficient (rule 0: nothing)
		for (int rule = 0; rule <= CK_LH_RULES; rule++) for (int v = -200; v <= 200; v++) for (int before = -40; before <= 40; before++, checked++) {
			const int bc = (before < -9 ? -9 : before > 8 ? 8 : before) + 9;
			const int got = (int16_t)(v + reinterpret_cast<const int8_t *>(ktab + CK_LHT_OFF)[(rule * 9 + lh_class(v)) * 18 + bc]);
			if (got != lh_rule(rule, v, before)) { if (bad++ < 5) printf("q%d lh(%d,%d,%d) %d != %d\n", q, rule, v, before, got, lh_rule(rule, v, before)); }
		}
		// the word of "no kind" does nothing
		{ const uint32_t aw = reinterpret_cast<const uint32_t *>(ktab + CK_ACT_OFF)[CK_NONE]; checked++;
		  if ((aw & 0xFFFF) || ((aw >> 16) & 7) != 2 || ((aw >> 19) & 7) != 2 || (aw >> 22)) { bad++; printf("q%d none-word %08x\n", q, aw); } }
		// Y23: cell and coefficient from the table = from the chain
		for (int res = -20; res <= 20; res++) for (int lv0 = -300; lv0 <= 300; lv0++) for (int vm1 = -20; vm1 <= 20; vm1++, checked++) {
			int la = lv0, lb = lv0;
			const int ca = code_step_reg(q, rs, 100 + res, 100, la, vm1), cb = code_step_tab(ytab, 100 + res, 100, lb, vm1);
			if (ca != cb || la != lb) { if (bad++ < 5) printf("q%d y23(res %d, lv %d, vm1 %d): %d/%d != %d/%d\n", q, res, lv0, vm1, ca, la, cb, lb); }
		}
		static const int codes[10] = { 14000, 14500, 12200, 12100, 12300, 12400, 14100, 12500, 12600, 14900 };
		for (int k = 0; k < 10; k++, checked++) {
			int la = 7, lb = 7;
			if (code_step_reg(q, rs, 55, codes[k], la, 3) != code_step_tab(ytab, 55, codes[k], lb, 3) || la != lb) { bad++; printf("q%d code %d\n", q, codes[k]); }
		}
	}
	// Y21: the cell rule on value codes = the reference's chain of comparisons (nhw_encoder.c:970-1073), for every value a cell or a neighbour can hold
	{
		auto ref = [](int pass, int x, int lv, int rv, int &own, int &force, int &trip) {
			auto in47 = [](int v) { return v > 3 && v <= 7; };
			auto inm = [](int v) { return v < -3 && v >= -7; };
			own = x; force = 0; trip = 0;
			if (x > 4 && x < 8) { if (in47(lv) && in47(rv)) { own = 12700; force = 10100; trip = 1; } }
			else if (x < -4 && x > -8) { if (inm(lv) && inm(rv)) { own = 12900; force = 10100; trip = 1; } }
			else if (x == 8) { if ((lv & 0xFFFE) == 6 || (rv & 0xFFFE) == 6) own = 10; else if (!pass && rv == 8) { own = 9; force = 9; } }
			else if (x == -8) { if (((-lv) & 0xFFFE) == 6 || ((-rv) & 0xFFFE) == 6) own = -9; else if (!pass && rv == -8) { own = -9; force = -9; } }
		};
		int vals[128], nv = 0;
		for (int v = -40; v <= 40; v++) vals[nv++] = v;
		const int extra[] = { 10100, 12700, 12900, 10204, 10300, 127, -127, 300, -300, 2047, -2047, 12100, 12200 };
		for (int v : extra) vals[nv++] = v;
		for (int pass = 0; pass < 2; pass++) for (int a = 0; a < nv; a++) for (int b = 0; b < nv; b++) for (int c3 = 0; c3 < nv; c3++, checked++) {
			const int x = vals[a], lv = vals[b], rv = vals[c3];
			int o1, f1, t1, o2, f2; bool t2;
			ref(pass, x, lv, rv, o1, f1, t1);
			tag_rule(pass, true, x, tag_code(x), tag_code(lv), tag_code(rv), o2, f2, t2);
			if (o1 != o2 || f1 != f2 || t1 != (int)t2) { if (bad++ < 5) printf("y21 pass %d (%d,%d,%d): %d %d %d != %d %d %d\n", pass, x, lv, rv, o1, f1, t1, o2, f2, (int)t2); }
			// a value a cell can be set or forced to never fires itself (what lets the kernel give such a cell the code 0)
			if (o1 != x && (tag_code(o1) & 15)) { if (bad++ < 5) printf("y21 result %d has a class\n", o1); }
		}
		// four cells at once: "nothing fires" is exactly "no cell fires with its neighbours as they are"
		unsigned rng = 12345;
		auto next = [&]() { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 16) % 23) - 11; };   // -11 .. 11: dense in the values that matter
		for (int pass = 0; pass < 2; pass++) for (int it = 0; it < 2000000; it++, checked++) {
			const int l = next(), r = next(); int x[4], e[4];
			for (int k = 0; k < 4; k++) { x[k] = next(); e[k] = tag_code(x[k]); }
			const uint32_t act = 0x01010101u & ~((it & 7) == 0 ? 1u : 0u) & ~((it & 15) == 1 ? 1u << 24 : 0u);   // now and then the first / last cell is outside the pass
			const uint32_t fires = tag_fires4(pass, e, tag_code(l), tag_code(r), act);
			bool any = false;
			for (int k = 0; k < 4; k++) {
				if (!((act >> (8 * k)) & 1)) continue;
				int o1, f1, t1;
				ref(pass, x[k], k ? x[k - 1] : l, k < 3 ? x[k + 1] : r, o1, f1, t1);
				const bool f = o1 != x[k] || f1 || t1;
				any |= f;
				if (f != (((fires >> (8 * k)) & 1) != 0)) { if (bad++ < 5) printf("y21 fires4 pass %d cell %d (%d | %d %d %d %d | %d)\n", pass, k, l, x[0], x[1], x[2], x[3], r); }
			}
			if (any != (fires != 0)) bad++;
		}
	}
	printf("checked %ld, mismatches %ld\n", checked, bad);
	return bad != 0;
}
