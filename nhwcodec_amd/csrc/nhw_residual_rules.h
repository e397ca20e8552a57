/* nhw_residual_rules.h -- the rules of Y22 / Y23 (nhw_encoder.c:1077-1420) as pure functions and the class tables the column sweep
 * (residuals_fused_par, nhw_tail_par.h) runs them from.  Compiled by hipcc into the kernels and by g++ into tests/test_residual_rules.py, which
 * walks the tables' whole domains against the comparison chains they stand for. */
#ifndef NHW_RESIDUAL_RULES_H
#define NHW_RESIDUAL_RULES_H
#include <stdint.h>
#ifndef __HIPCC__                                                 /* the host build of the test */
#define DEV static inline
#define RULE_FN static inline
static inline int __mul24(int a, int b) { return a * b; }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int mult8_or_7(int m) { return !(m & 7) || (m & 7) == 7; }
#else
#define RULE_FN __device__ __forceinline__
#endif
/* ---------------------------------------------------------------- Y22 / Y23 (C) */
/* one column of Y22 (:1084-1325).  sp/so: where the reads of column j+1 (and of the recon sample (j,255) that
 * serves as lh[-1] at r = 0) are taken from: a snapshot for columns 0..254, the live planes for column 255,
 * which the reference visits last. */
/* One step of the Y22 column walk (nhw_encoder.c:1077-1325) at row r of column j.  pr / orow point at the column's
 * recon sample / LL1 cell of row r (row strides ps / os: the planes themselves, an LDS tile, or a packed copy of
 * the column), lh at the LH1 coefficient the step may nudge, lhm1 is the one before it (as this walk left it).
 * sp / so: where the right-hand neighbour column is read (the values from before the pass). */
/* One step of the Y22 column walk (nhw_encoder.c:1077-1325) decides on three small differences -- this row's residual,
 * the next row's, the one after -- through a chain of some forty comparisons.  Every comparison only asks which of
 * 16 x 9 x 12 value classes the triple is in, so the chain is evaluated once per class (classify_kind, by the threads
 * of the workgroup, into an LDS table) and a step is a table lookup plus a short switch: with 64 columns in a
 * wavefront the chain itself was most of the divergent instruction stream. */
enum { CK_NONE, CK_MARKP, CK_MARKN, CK_S12100, CK_S12500, CK_S12200, CK_S12600, CK_INC, CK_DEC, CK_Q18P_NEXT, CK_Q18P_CELL, CK_Q18P_COPY,
       CK_Q18N_NEXT, CK_Q18N_CELL, CK_Q18N_COPY, CK_NBP, CK_NBN, CK_PREVGE0, CK_PREVLE0, CK_C14500, CK_NUP, CK_NM2, CK_NM3,
       CK_LARGE4, CK_LARGE56, CK_LARGE7, CK_LARGE8, CK_KINDS };   /* CK_LARGE*: by the residual (-4; -5, -6; -7; below) */
#define CK_CLASS_BYTES (17 * 9 * 12)
/* behind the class table: what a kind does, one word per kind (classify_action), and what its rule does to the LH1 coefficient by the
 * coefficient's class and the one before it (lh_table_fill) -- so that the 64 columns of a wavefront take ONE path through the step
 * whatever their kinds (the switch over the kinds was 200 of the step's 230 instructions: every step found most kinds among its lanes) */
#define CK_ACT_OFF CK_CLASS_BYTES
#define CK_LHT_OFF (CK_ACT_OFF + 4 * CK_KINDS)
#define CK_LH_RULES 5
#define CK_TABLE_BYTES ((CK_LHT_OFF + (CK_LH_RULES + 1) * 9 * 18 + 3) & ~3)   /* (rule 0: a row of zeros, so that the step needs no branch) */
DEV int classify_kind(int q, int res_setting, int res, int a, int d2)
{
	if (res == 2 && a == 2 && d2 >= 2) return (d2 < 5 || d2 > 6) ? CK_MARKP : CK_NONE;
	if (((res == 2 && a == 3) || (res == 3 && a == 2)) && d2 > 1 && d2 < 6) return CK_MARKP;
	if (res == 3 && a == 3) return (d2 > 0 && d2 < 6) ? CK_MARKP : (q >= 19 ? CK_S12100 : CK_NONE);
	if (a == -4 && (res == 2 || res == 3) && (d2 == 2 || d2 == 3)) return (res == 2 && d2 == 2) ? CK_INC : CK_MARKP;
	if (res == 1 && a == 3 && d2 == 2) return CK_PREVGE0;
	if ((res == 3 || res == 4 || res == 5 || res > 6) && (a == 3 || (a & 0xFFFE) == 4)) {
		if (res > 6) return CK_S12500;
		if (q >= 19) return CK_S12100;
		if (q == 18) return (res < 5 && a == 5) ? CK_Q18P_NEXT : (res >= 5 ? CK_Q18P_CELL : ((res == 3 && a >= 4) ? CK_Q18P_NEXT : CK_Q18P_COPY));
		return CK_NONE;
	}
	if ((res == 2 || res == 3) && (a == 2 || a == 3)) return (d2 == 0 || d2 == 1) ? CK_NBP : CK_NONE;
	if (a == 4 && (res == -2 || res == -3) && (d2 == -2 || d2 == -3)) return (res == -2 && d2 == -2) ? CK_DEC : CK_MARKN;
	if ((res == -3 || res == -4 || res == -5 || res < -7) && (a == -3 || a == -4 || a == -5)) {
		if (res < -7) return CK_S12600;
		if (q >= 19) return CK_S12200;
		if (q == 18) return (res > -5 && a == -5) ? CK_Q18N_NEXT : (res <= -5 ? CK_Q18N_CELL : ((res == -3 && a <= -4) ? CK_Q18N_NEXT : CK_Q18N_COPY));
		return CK_NONE;
	}
	if (a == -2 || a == -3) {
		if (res == -2 || res == -3) {
			if (d2 < 0) return CK_MARKN;
			if (res == -3 && q >= 21) return CK_C14500;
			if (d2 == 0) return CK_NBN;
			return res == -2 ? CK_NM2 : CK_NM3;
		}
		if (res == -1 && a == -3 && d2 == -2) return CK_PREVLE0;
		if (res == -1) return d2 == -3 ? CK_MARKN : CK_NUP;
		if (res == -4) return (d2 < -1 && d2 > -4) ? CK_MARKN : CK_LARGE4;
		return CK_NONE;
	}
	if (!res || res == -1) return CK_NUP;
	if (res == -2) return CK_NM2;
	if (res == -3) return CK_NM3;
	if (res < -res_setting) return res == -4 ? CK_LARGE4 : res >= -6 ? CK_LARGE56 : res == -7 ? CK_LARGE7 : CK_LARGE8;
	return CK_NONE;
}
/* what a kind does (:1084-1325), as a word: bits 0..15 the code its LL1 cell takes (0: none), 16..18 / 19..21 what is added to the recon
 * samples one / two rows down (+2), 22 "the sample one row down becomes its LL1 cell", 23..24 that cell first becomes 14100 (1) / 14000 (2)
 * (q18 only), 25.. the rule applied to the LH1 coefficient (0: none) */
enum { LHR_NONE, LHR_NUP, LHR_NM2, LHR_NM3, LHR_L4, LHR_L6 };
DEV uint32_t ck_word(int code, int d1, int d2, int snap, int next, int rule) { return (uint32_t)code | (uint32_t)(d1 + 2) << 16 | (uint32_t)(d2 + 2) << 19 | (uint32_t)snap << 22 | (uint32_t)next << 23 | (uint32_t)rule << 25; }
DEV uint32_t classify_action(int kind, int q)
{
	switch (kind) {
	case CK_MARKP: return ck_word(12400, -2, -2, 0, 0, 0);
	case CK_MARKN: return ck_word(12300, 2, 2, 0, 0, 0);
	case CK_S12100: return ck_word(12100, 0, 0, 1, 0, 0);
	case CK_S12500: return ck_word(12500, 0, 0, 1, 0, 0);
	case CK_S12200: return ck_word(12200, 0, 0, 1, 0, 0);
	case CK_S12600: return ck_word(12600, 0, 0, 1, 0, 0);
	case CK_INC: return ck_word(0, 1, 0, 0, 0, 0);
	case CK_DEC: return ck_word(0, -1, 0, 0, 0, 0);
	case CK_Q18P_NEXT: return ck_word(0, 0, 0, 1, 1, 0);
	case CK_Q18P_CELL: return ck_word(14100, 0, 0, 1, 0, 0);
	case CK_Q18P_COPY: return ck_word(0, 0, 0, 1, 0, 0);
	case CK_Q18N_NEXT: return ck_word(0, 0, 0, 1, 2, 0);
	case CK_Q18N_CELL: return ck_word(14000, 0, 0, 1, 0, 0);
	case CK_Q18N_COPY: return ck_word(0, 0, 0, 1, 0, 0);
	case CK_C14500: return ck_word(14500, 0, 0, 0, 0, 0);
	case CK_NUP: return ck_word(0, 0, 0, 0, 0, LHR_NUP);
	case CK_NM2: return ck_word(0, 0, 0, 0, 0, LHR_NM2);
	case CK_NM3: return q >= 21 ? ck_word(14500, 0, 0, 0, 0, 0) : ck_word(0, 0, 0, 0, 0, LHR_NM3);
	case CK_LARGE4: return ck_word(14000, 0, 0, 0, 0, LHR_L4);
	case CK_LARGE56: return ck_word(14000, 0, 0, 0, 0, 0);
	case CK_LARGE7: return ck_word(14000, 0, 0, 0, 0, LHR_L6);
	case CK_LARGE8: return q >= 21 ? ck_word(14900, 0, 0, 0, 0, 0) : ck_word(14000, 0, 0, 0, 0, LHR_L6);
	default: return ck_word(0, 0, 0, 0, 0, 0);                  /* CK_NONE; the four kinds that look at more cells become one of the above first */
	}
}
/* the rules on the LH1 coefficient v of the cell, given the one before it (as the walk left it) */
DEV int lh_rule(int rule, int v, int before)
{
	switch (rule) {
	case LHR_NUP:
		if (v == 7) { if (before >= 0 && before < 8) v += 2; } else if (v == 8) { if (before >= -2 && before < 8) v += 2; }
		break;
	case LHR_NM2:
		if (v < -14) { if (mult8_or_7(-v)) v++; } else if (v == 7 || (v & 0xFFFE) == 8) { if (before >= -2) v += 3; }
		break;
	case LHR_NM3:
		if (v < -14) { if (mult8_or_7(-v)) v++; }
		else if (v >= 0 && ((v + 2) & 0xFFFC) == 8) { if (before >= -2) v = 10; }
		else if (v > 14 && (v & 7) == 7) v++;
		break;
	case LHR_L4:
		if (v == -7 || v == -8) { if (before < 2 && before > -8) v = -9; }
		break;
	case LHR_L6:
		if (v < -14) { if (mult8_or_7(-v)) v++; } else if (v == 7 || v == 8) { if (before >= -1 && before < 8) v += 3; }
		break;
	default: break;
	}
	return v;
}
/* the rules only tell these coefficients apart: below -14 on a multiple of 8 or one short of it (1), -8, -7 (2, 3), 6 .. 9 (4 .. 7), above 14
 * and 7 modulo 8 (8), anything else (0: no rule moves it); and of the coefficient before, where it lies in -9 (or less) .. 8 (or more) */
DEV int lh_class(int v)
{
	const int m = (1 - v) & 7;                                     /* 0 or 1: -v is a multiple of 8 or one short of it */
	int c = (v < -14 && m < 2) ? 1 : 0;
	c = (v > 14 && (v & 7) == 7) ? 8 : c;
	c = (unsigned)(v + 8) <= 1u ? v + 10 : c;
	c = (unsigned)(v - 6) <= 3u ? v - 2 : c;
	return c;
}
/* value classes: residual <= -9, -8 .. 6, >= 7 (the chain compares it with -7 and with -res_setting >= -8); next residual -5 .. -2, 2 .. 5, anything else; third <= -4, -3 .. 6, >= 7 */
DEV void classify_table_fill(uint8_t *tab, int q, int res_setting, int tid, int nt /* threads that share the filling */)
{
	for (int idx = tid; idx < CK_CLASS_BYTES; idx += nt) {
		const int rc = idx / (9 * 12), ac = (idx / 12) % 9, dc = idx % 12;
		const int a = ac < 4 ? ac - 5 : (ac < 8 ? ac - 2 : 0);
		tab[idx] = (uint8_t)classify_kind(q, res_setting, rc - 9, a, dc - 4);
	}
	for (int k = tid; k < CK_KINDS; k += nt) reinterpret_cast<uint32_t *>(tab + CK_ACT_OFF)[k] = classify_action(k, q);
	for (int idx = tid; idx < (CK_LH_RULES + 1) * 9 * 18; idx += nt) {
		const int rule = idx / (9 * 18), lc = (idx / 18) % 9, before = idx % 18 - 9;
		const int v = lc == 0 ? 0 : lc == 1 ? -16 : lc == 8 ? 15 : lc < 4 ? lc - 10 : lc + 2;   /* one coefficient of the class */
		reinterpret_cast<int8_t *>(tab + CK_LHT_OFF)[idx] = (int8_t)(lh_rule(rule, v, before) - v);
	}
}
DEV int classify_lookup(const uint8_t *tab, int res, int a, int d2)
{
	const int rc = (res < -9 ? -9 : (res > 7 ? 7 : res)) + 9, dc = (d2 < -4 ? -4 : (d2 > 7 ? 7 : d2)) + 4;
	const int ac = (a >= -5 && a <= 5) ? (int)((0x76548883210ull >> (4 * (a + 5))) & 15) : 8;
	return tab[__mul24(__mul24(rc, 9) + ac, 12) + dc];
}

/* Y23's step (:1329-1420): pv the recon sample, cell the LL1 cell as Y22 left it, lv the LH1 coefficient (j, 256 + r), vm1 the one before as
 * THIS walk left it; returns what the cell becomes */
RULE_FN int code_step_reg(int q, int res_setting, int pv, int cell, int &lv, int vm1)
{
	if (cell < 12000) {
		const int res = pv - cell;
		int out = 0;
		if (!res || res == 1) { if (lv == -7 || lv == -8) { if (vm1 < 2 && vm1 > -8) lv = -9; } }
		else if (res == 2) {
			if (lv > 15 && !(lv & 7)) lv--;
			else if (lv == -7 || lv == -8) { if (vm1 <= 1) lv = -9; }
			else if (lv == -6) { if (vm1 <= -1 && vm1 > -8) lv = -9; }
		}
		else if (res == 3) {
			if (q >= 21) out = 144;
			else if (lv > 15 && !(lv & 7)) lv--;
			else if (lv <= 0 && (((-lv) + 2) & 0xFFFC) == 8) { if (vm1 <= 2) lv = -10; }
		}
		else if (res > res_setting) {
			out = 141;
			if (res == 4) { if (lv == 7 || (lv & 0xFFFE) == 8) { if (vm1 >= 0 && vm1 < 8) lv += 2; } }
			else if (res > 6) {
				if (res > 7 && q >= 21) out = 148;
				else if (lv > 15 && !(lv & 7)) lv--;
				else if (lv == -6 || lv == -7 || lv == -8) { if (vm1 < 0 && vm1 > -8) lv = -9; }
			}
		}
		return out;
	}
	switch (cell) {
	case 14000: return 140; case 14500: return 145; case 12200: return 122; case 12100: return 121; case 12300: return 123;
	case 12400: return 124; case 14100: return 141; case 12500: return 125; case 12600: return 126; case 14900: return 149;
	default: return cell;
	}
}
/* The same step from a table: what it does to the coefficient depends on the residual (below 0: nothing; 0 .. 8; above), on which of six
 * classes the coefficient is in (-9; -8, -7; -6; 7 .. 9; above 15 and a multiple of 8; anything else) and on where the coefficient before
 * lies (up to -8; -7 .. -1; 0, 1; 2; 3 .. 7; from 8) -- every comparison of code_step_reg is constant on these classes -- and is one of
 * five things (nothing, = -9, = -10, - 1, + 2); what the cell becomes depends on the residual alone.  The table is filled by running
 * code_step_reg on a representative of every class (the branches of the chain were most of the step's instructions: 64 columns
 * find most of them). */
#define Y23_OPS (11 * 6 * 6)
#define Y23_TAB_BYTES ((Y23_OPS + 11 + 3) & ~3)
DEV void code_table_fill(uint8_t *yt, int q, int res_setting, int tid, int nt)
{
	for (int idx = tid; idx < Y23_OPS + 11; idx += nt) {
		if (idx < Y23_OPS) {
			const int res = idx / 36 - 1, lcl = (idx / 6) % 6, vcl = idx % 6;
			const int lv0 = lcl == 0 ? 100 : lcl == 1 ? -9 : lcl == 2 ? -8 : lcl == 3 ? -6 : lcl == 4 ? 8 : 16;
			const int vm1 = vcl == 0 ? -8 : vcl == 1 ? -7 : vcl == 2 ? 0 : vcl == 3 ? 2 : vcl == 4 ? 3 : 8;
			int lv = lv0;
			code_step_reg(q, res_setting, res, 0, lv, vm1);
			yt[idx] = (uint8_t)(lv == lv0 ? 0 : lv == -9 ? 1 : lv == -10 ? 2 : lv == lv0 - 1 ? 3 : 4);
		} else {
			int lv = 100;
			yt[idx] = (uint8_t)code_step_reg(q, res_setting, idx - Y23_OPS - 1, 0, lv, 0);
		}
	}
}
RULE_FN int code_step_tab(const uint8_t *yt, int pv, int cell, int &lv, int vm1)
{
	const int res = pv - cell, rcl = (res < -1 ? -1 : res > 9 ? 9 : res) + 1;
	int lcl = (unsigned)(lv + 9) <= 3u ? (int)((0x3221u >> (4 * (lv + 9))) & 15u) : 0;
	lcl = (unsigned)(lv - 7) <= 2u ? 4 : lcl;
	lcl = (lv > 15 && !(lv & 7)) ? 5 : lcl;
	const int vcl = vm1 <= -8 ? 0 : vm1 <= -1 ? 1 : vm1 <= 1 ? 2 : vm1 == 2 ? 3 : vm1 <= 7 ? 4 : 5;
	const int op = yt[__mul24(__mul24(rcl, 6) + lcl, 6) + vcl];
	const int nl = op == 0 ? lv : op == 1 ? -9 : op == 2 ? -10 : op == 3 ? lv - 1 : lv + 2;
	const bool plain = cell < 12000;
	/* a code of Y22 (12100 .. 14900) becomes its hundredth (the switch of :1398-1416; from 12000 on a cell IS a code: see RF_LDS_BYTES) */
	lv = plain ? nl : lv;
	return plain ? (int)yt[Y23_OPS + rcl] : (cell * 5243) >> 19;
}

/* ---- Y21 (:970-1073): the cell rule on value codes.  A code says where a value lies: 4 .. 7 (bit 0: a neighbour of a triple), 5 .. 7 (bit 1: its
 * centre), 6, 7 (bit 2: what turns an 8 into a 10), 8 (bit 3), with its sign (bit 4). */
DEV int tag_code(int v)
{
	int a = v < 0 ? -v : v;
	a = a > 9 ? 9 : a;
	return (int)((0x877310000ull >> (4 * a)) & 15u) | (v < 0 ? 16 : 0);
}
/* one cell: x its value (or the value forced on it: then ex = 0), ex / el / er the codes of the cell, of its left neighbour as the walk left it
 * and of its right neighbour as it was; pass 0: LH1, pass 1: HL1 (no pairs of 8s).  ow: what the cell becomes, force: what it forces on the
 * next cell (0: nothing), t3: it is the centre of a triple (the cell before it is marked as well) */
RULE_FN void tag_rule(int pass, bool act, int x, int ex, int el, int er, int &ow, int &force, bool &t3)
{
	const bool sl = !((ex ^ el) & 16), sr = !((ex ^ er) & 16), neg = (ex & 16) != 0;
	t3 = act && (ex & 2) && (el & 1) && sl && (er & 1) && sr;
	const bool is8 = act && (ex & 8);
	const bool near = is8 && (((el & 4) && sl) || ((er & 4) && sr));
	const bool pair = !pass && is8 && !near && (er & 8) && sr;
	ow = x;
	ow = t3 ? (neg ? 12900 : 12700) : ow;
	ow = near ? (neg ? -9 : 10) : ow;
	ow = pair ? (neg ? -9 : 9) : ow;
	force = t3 ? 10100 : pair ? ow : 0;
}
/* does any of four cells in a row fire with its neighbours as they are?  A code a byte, all four at once; act_bytes: bit 0 of the bytes of the
 * cells that take part.  Zero = none fires, and then a walk over them changes nothing (a cell only sees another left neighbour, or a forced
 * value, behind a cell that fired). */
RULE_FN uint32_t tag_fires4(int pass, const int e[4], int e_left, int e_right, uint32_t act_bytes)
{
	const uint32_t E = (uint32_t)e[0] | (uint32_t)e[1] << 8 | (uint32_t)e[2] << 16 | (uint32_t)e[3] << 24;
	const uint32_t L = E << 8 | (uint32_t)e_left, R = E >> 8 | (uint32_t)e_right << 24;
	const uint32_t sl = ~((E ^ L) >> 4), sr = ~((E ^ R) >> 4);       /* bit 0 of a byte: the same sign */
	const uint32_t t3 = (E >> 1) & L & R & sl & sr, near = (E >> 3) & (((L >> 2) & sl) | ((R >> 2) & sr)), pair = pass ? 0u : (E >> 3) & (R >> 3) & sr;
	return (t3 | near | pair) & act_bytes;
}
#endif
