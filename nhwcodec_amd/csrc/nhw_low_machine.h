/*
 * nhw_low_machine.h -- the pair machine of the quality 1..16 luma pre-filter (rcanut/nhwcodec encoder/image_processing.c:770-1925) as a
 * pure automaton: its counters, the step that follows the reference line by line (machine_step) and the branch-free form of the steps it
 * spends its time on (machine_step_fast).  No device intrinsics in here: nhw_low.hip includes it with DEVI / DEVN = device functions;
 * tests/test_low_machine.py compiles the same text for the host and walks every form over whole images next to the CPU restatement of the
 * reference's machine.
 * The includer defines DEVI (inlined), DEVN (not inlined) and Q (the reference's IM_SIZE).
 */
#ifndef NHW_LOW_MACHINE_H
#define NHW_LOW_MACHINE_H
#ifndef PF_COV
#define PF_COV(n) ((void)0)      /* a developer tool (tools/dev/prelow_fuzz.cpp) counts which of the rare schedule branches a picture reaches */
#endif

/* ------------------------------------------------------------------------------------------------ pass B: the pair machine (:770-1992) */
/* The counters are numbered as the reference numbers its variables (t1..t44, w1..w8): they have no documented meaning, and a reader
 * can put the two side by side.  Constant indices only, so the arrays live in registers. */
struct PfM { int t[45], w[9]; };
#define T(n) (m.t[n])
#define Wv(n) (m.w[n])

DEVI void machine_reset(PfM &m)
{
	for (int i = 0; i < 45; i++) m.t[i] = 0;
	for (int i = 0; i < 9; i++) m.w[i] = 0;
	T(6) = 8; T(10) = 10; T(11) = 15; T(18) = 8; T(44) = 2; Wv(3) = 20;
}
DEVI void set_window(PfM &m, int wide) { if (wide) { T(10) = 10; T(11) = 15; } else { T(10) = 8; T(11) = 12; } }

/* schedule walked once the burst counter t7 has reached 4 (:1203-1448) */
DEVN void long_schedule(PfM &m)
{
	switch (T(16)) {
	case 0:
		set_window(m, 1); T(16) = 1;
		if ((Wv(7) == 2 || Wv(7) == 4) && T(24) == 14) { if (Wv(7) == 2) T(1) = 2000005; }
		else { T(4) = 1000000; T(1) = 9; }
		break;
	case 1:
		set_window(m, 0); T(16) = 2; Wv(5)++;
		if (Wv(5) == 3 && T(1) > 0 && T(1) < 30) T(1) = (-T(1)) >> 2;
		else { T(4) = 10; T(1) += 2; }
		break;
	case 2:
		set_window(m, 1); T(16) = 3; T(4) = 1000000; Wv(6)++;
		if (Wv(6) == 6 || Wv(6) == 10) T(1) = 10;
		break;
	case 3: set_window(m, 0); T(16) = 4; T(4) = 8; T(1) -= 4; break;
	case 4: set_window(m, 1); T(16) = 5; break;
	case 5: set_window(m, 1); T(16) = 6; T(4) = 10; T(1) = 2000000; break;
	case 6: set_window(m, 0); T(16) = 7; T(4) = 8; T(1) = 3000000; break;
	case 7: set_window(m, 0); T(16) = 8; T(4) = 1000000; break;
	case 8: {
		set_window(m, 0);
		const int s = T(24);
		if (s >= 0 && s < 14) {
			/* sub-position -> next position; the t4 / t1 presets are sparse: written out */
			int n16 = 1;
			switch (s) {
			case 0: n16 = 1; T(4) = 1000000; break;
			case 1: n16 = 2; break;
			case 2: n16 = 1; T(4) = 1000000; break;
			case 3: n16 = 2; break;
			case 4: n16 = 1; T(1) = 2999998; break;
			case 5: n16 = 0; break;
			case 6: n16 = 3; break;
			case 7: n16 = 3; T(1) = 7; break;
			case 8: n16 = 1; break;
			case 9: n16 = 8; T(4) = 1000000; break;
			case 10: n16 = 1; T(4) = 8; T(1) = 11; break;
			case 11: n16 = 0; break;
			case 12: n16 = 1; break;
			default: n16 = 0; break;     /* 13 */
			}
			T(16) = n16; T(24) = s + 1;
		}
		else if (s == 14) { T(16) = 1; T(24) = 15; Wv(7)++; T(1) = Wv(2) == 0 ? 1999978 : Wv(2) == 1 ? 1999982 : 1999993; }
		else if (s == 15) { T(16) = 0; T(24) = 12; T(1) = (Wv(2) == 1 || Wv(2) == 3) ? -5 : 2000005; Wv(2)++; }
		break;
	}
	default: break;
	}
}

/* end of a burst (:1053-1456) */
DEVN void burst_end(PfM &m)
{
	if (!T(6)) {
		T(6) = 1; T(14) = 0;
		if (!T(22)) T(7)++;
		if (T(22) == 1) T(22) = 0;
	} else {
		T(6)++; T(1)++;
		if (T(4) > 900000 && T(1) == 12) T(4) = 8;
		if (T(1) > 3000000) { T(1) = 12; T(4) = 8; }
		else if (T(1) > 2000006 && T(1) < 2500000) { T(1) = 14; T(4) = 10; }
		if (!T(15)) { T(14) = 1; T(15) = 1; }
		else { T(14) = 0; T(15)++; if (T(15) > 9) T(15) = 0; }
		if (T(6) > 15 && T(7) < 4) { T(6) = 0; if (T(19) > 0) T(20)++; }
	}
	if (T(4) == 8 || (T(4) == 10 && Wv(3) > 16)) {
		if (Wv(3) < 21) { T(4) = 0; Wv(3)++; }
		else if (T(4) == 8) Wv(3) = 0;
		else if (Wv(4) < 2) { T(4) = 8; T(1) = 12; Wv(4)++; }
		else { T(4) = 0; Wv(4) = 0; }
	}
	else T(4) = 0;
	T(8) = 0; T(5) = 0; T(12) = 0;
	if (T(7) == 3) set_window(m, !T(6));
	else if (T(7) == 1) {
		set_window(m, T(9) < 2);
		T(9)++;
		if (T(9) >= 3 && T(10) == 8) T(9) = 0;
	}
	else if (T(7) == 2) set_window(m, 0);
	else if ((T(6) == 10 || T(6) == 11) && !T(7)) { T(10) = 6; T(11) = 9; }
	else if (T(7) >= 4) long_schedule(m);
	else { T(10) = T(10) == 8 ? 10 : 8; T(11) = T(11) == 12 ? 15 : 12; }
}

/* a pair inside a burst that neither ends it nor sits at its cap (:1504-1873) */
DEVN void burst_idle(PfM &m)
{
	if (T(1) == 6 && !Wv(8)) { T(1)++; Wv(8)++; T(44) = -100000; }
	else if (T(44) < -90000) { T(1)++; Wv(8)++; T(44) = 0; }
	else if (T(44) < 3) T(44)++;
	else { T(1) += 3; T(44) = 0; }

	if (!(T(29) > 0 && (T(14) == 4 || T(14) == 5 || T(39) == 2 || T(41) > 0))) return;

	if (T(4) < 2 && T(1) == 15 && (T(14) == 4 || (T(14) == 5 && T(32) > 2))) {
		PF_COV(0);
		if (T(32) == 0 || T(32) == 2 || T(32) == 3 || (T(32) > 7 && T(32) < 500000)) {
			if (T(32) > 7 && T(14) == 5) { PF_COV(1); T(14) = 1; T(32) = 1000000; }
			else if (!T(34)) { PF_COV(2); T(34) = 1; }
			else { PF_COV(3); T(14) = 5; T(34) = 0; }
		}
		if (!T(32)) T(14) = 5;
		T(32)++;
	}
	else if (T(32) == 4 || T(32) == 5 || T(32) == 7) {
		PF_COV(4);
		if (T(37) == 4) { PF_COV(5); T(14) = 3; }
		else if (T(37) == 15) { PF_COV(6); T(14) = 3; T(32)++; }
		else if (T(32) == 7 && T(37) > -345000) {
			PF_COV(7);
			if (T(14) == 4) {
				PF_COV(8);
				if (!T(42)) T(37) -= 10000;
				if (T(38) > 0) {
					PF_COV(9);
					T(42)++;
					if (T(42) > 0 || (!T(42) && T(43) > 3)) {
						PF_COV(10);
						if (!T(42)) { PF_COV(11); T(14) = T(43) == 14 ? 3 : T(43) == 24 ? 4 : 1; }
						else T(14) = 1;
						T(39) = 0;
						if (T(42) > 5) { PF_COV(12); T(42) = -1; T(43)++; }
					}
					else if (T(42) == -1) { PF_COV(13); T(14) = 3; T(39) = 2; T(40) = -2; T(42) = 0; }
					else { PF_COV(14); T(39) = 0; }
				}
				else { PF_COV(15); T(14) = 5; T(39) = 1; T(42) = 0; }
			}
			else if (T(39) >= 1) {
				PF_COV(16);
				T(38)++;
				if (T(39) < 2) T(39) = (T(38) == 2 || T(38) == 4 || T(38) == 6 || T(38) == 9) ? 2 : 0;
				else {
					T(40)++;
					if (T(38) == 8) { T(39) = 0; T(40) = 0; }
					if (T(40) > 2) { T(40) = 0; T(39) = 0; }
				}
				if (T(38) >= 1 && T(38) <= 10) T(14) = 4;
			}
			else { PF_COV(17); T(40) = 1; if (T(38) == 1) T(39) = 2; }
		}
		if (T(37) >= 0) T(37)++;
	}
	else if (T(32) == 6 && T(36) < 118) {
		PF_COV(18);
		if (T(14) == 4 || T(14) == 5 || T(41) == 0 || T(41) > 3) T(36)++;
		if (T(41) > 3 && T(36) < 8) T(41) = 0;
		switch (T(36)) {                 /* t36 -> t14; t41 is reset, counted up or set to 4 */
		case 1: T(14) = 1; T(41) = 0; break;   case 2: T(14) = 2; T(41) = 0; break;   case 3: T(14) = 1; T(41) = 0; break;
		case 4: T(14) = 3; T(41) = 0; break;   case 5: T(14) = 3; T(41)++; break;     case 6: T(14) = 0; T(41) = 0; break;
		case 7: T(14) = 2; T(41) = 0; break;   case 8: T(14) = 2; T(41) = 4; break;   case 15: T(14) = 1; T(41) = 0; break;
		case 31: T(14) = 3; T(41)++; break;    case 47: T(14) = 2; T(41) = 0; break;  case 100: T(14) = 0; T(41)++; break;
		case 116: T(14) = 2; T(41) = 0; break;
		default: break;
		}
	}

	if (T(28) < 14 && T(1) > 7) {                        /* :1711-1871 */
		const int st = T(28);
		if (T(14) == 5 && !st && !T(33) && T(1) > 13 && T(31) > 0) { T(30) = 1; T(33) = 2; }
		else T(30)++;
		const int ahead = T(30) - T(33);
		int late_d = 0x7fffffff, l14 = 0, l15 = 0, l1 = 0, l4 = 0;   /* stages 6..12 fire once t30 has run far enough past t33 */
		switch (st) {
		case 6: late_d = 54; l14 = 2; l15 = 3; l1 = 3; break;    case 7: late_d = 57; l14 = 2; l15 = 8; l1 = 8; break;
		case 8: late_d = 84; l14 = 2; l15 = 7; l1 = 7; break;    case 9: late_d = 111; l14 = 2; l15 = 3; l1 = 7; break;
		case 10: late_d = 116; l14 = 1; l15 = 0; l1 = 1; l4 = 8; break;
		case 11: late_d = 185; l14 = 0; l15 = 4; l1 = -17; break; case 12: late_d = 187; l14 = 3; l15 = 3; l1 = -19; break;
		default: break;
		}
		if (!st && ahead > 10 && T(33) > 0 && T(14) == 4) { PF_COV(20); T(14) = 3; T(15) += 6; T(28)++; }
		else if (st == 1 && ahead > 70 && T(14) == 4 && T(1) == 11) { PF_COV(21); T(15) = 1; T(1) = 13; T(28)++; }
		else if (st == 2 && T(31) > 2 && T(1) == 15 && T(15) > 1) { PF_COV(22); T(15) = 15; T(33) = T(30); T(1) = 6; T(28)++; }
		else if (st == 3 && ahead > 3 && T(31) > 2) { PF_COV(23); T(15) = 0; T(28)++; }
		else if (st == 5 && ahead > 22 && T(31) > 2 && T(1) == 12) { PF_COV(25); T(15) = 3; T(1) = 9; T(28)++; }
		else if (st == 4 && ahead > 6 && T(1) == 15) { PF_COV(24); T(14) = 1; T(15) += 6; T(1)++; T(28)++; }
		else if (st >= 6 && st <= 12 && ahead > late_d) { PF_COV(20 + st); T(14) = l14; T(15) = l15; T(1) = l1; if (l4) T(4) = l4; T(28)++; }
		else if (ahead == 9) { T(1) += (12 - T(4)) >> 2; T(4) = 10; }
		else if (st > 0 && T(1) == 15 && Wv(1) < 11) { if (T(4) != 10) { if (Wv(1) == 4 || Wv(1) == 10) T(4) = 10; Wv(1)++; } }
		else if (st == 13 && ahead > 188) { PF_COV(33); T(14) = 0; T(15) = 3; T(1) = -30; T(28)++; }
	}
}

/* one pixel pair of the sharpening machine (:838-1925) WITHOUT its data: the machine never looks at a contrast value itself, only at four
 * threshold tests of the pair (code: bit 0 |k0| > sharp, bit 1 |k1| > sharp, bit 2 |k1| > sharp2, bit 3 |k0| > sharp + 96), and what it does
 * to the picture is decided by three bits of its answer (ACT_FIRST: the pair opens a burst -- strength 2 instead of 1; ACT_ZERO0: the
 * first map cell is cleared; ACT_SUBST: the second pixel goes through the marker substitution / weak-first rule of :893-917).  So the
 * wavefront computes the codes of a whole row in parallel, the chain lane walks 255 codes in registers (no loads, stores or compares of
 * picture data on the chain), and the wavefront applies the answers in parallel (pair_apply).  The counters change exactly as in the
 * reference. */
#define ACT_FIRST 1
#define ACT_ZERO0 2
#define ACT_SUBST 4
DEVI int machine_step(PfM &m, int code, int row)
{
	const bool f0 = code & 1, f1 = code & 2, g1 = code & 4, h0 = code & 8;
	int act = 0;
	if (!T(1)) {                                     /* first pair of a burst (:840-994) */
		act = ACT_FIRST;
		T(2) = 0;
		if (f0) {
			if (g1 || T(8) == 1) {
				act |= ACT_ZERO0;
				if ((T(19) < 4 * Q || (T(20) >= 3 && T(20) < 4 * Q)) && h0 && T(6) > 0 && row > 2) {
					if (T(20) >= 3 && T(19) >= 8 * Q) { T(6) = 7000000; T(20) = 8 * Q; }
					if (T(19) > 0 && T(19) < 4 * Q) {
						if (T(20) > 2 || (T(20) == 2 && T(6) > 3 && !T(23)) || (T(20) == 2 && T(6) > 14 && T(23) > 0)) {
							if (T(23) == 1) T(6) = 5000000;
							T(23)++; T(21)++;
							if (T(21) >= 2) T(19) = 8 * Q;
						}
					}
					if (!T(19)) { T(6)++; T(20) = 1; }
					T(19)++;
				}
			}
			T(2) = 1;
		}
		if (f1) {
			if ((T(2) == 1 || T(12) == 1) && (!T(14) || T(14) == 4 || T(14) == 5)) {
				if (!T(3) && T(2) == 1) { act |= ACT_SUBST; T(3) = 1; }
				else T(3) = T(3) == 1 ? 2 : T(3) == 2 ? 3 : 0;
			}
			if (T(14) == 2) { T(14) = 1; T(26) = 3; if (T(25) > 0) T(25)++; }
			if (T(14) == 1) { if (T(26) < 4) T(26)++; else { T(14) = 2; T(26) = 0; } }
		}
		if (f0 || f1) T(13) = 1;
		if (T(14) == 1 || T(14) == 2) T(27)++; else T(27) = 0;
		if (T(27) > 2) T(14) = 1;
		if (T(14) == 1) {
			T(14) = 4;
			if (!T(25)) { T(15)++; T(25) = 1; }
			else { T(25)++; if (T(25) > 3) T(25) = 0; }
		}
		T(1) = 1;
	} else {                                         /* inside a burst (:995-1910) */
		const int fires = (f0 ? 1 : 0) + (f1 ? 1 : 0);
		T(1) += fires; T(4) += fires;

		if (T(4) < 10) T(17) = (T(4) == T(10) && T(1) == T(11));
		else if (T(4) > 10 || T(1) != 15) {
			if (!T(18)) { T(17) = 1; T(18) = 1; }
			else { T(17) = 0; T(18)++; if (T(18) > 15) T(18) = 0; }
		}
		else T(17) = (T(4) == T(10) && T(1) == T(11));

		if (T(6) > 6000000) { T(6) = 0; T(22) = 0; }
		else if (T(6) > 4000000) { T(6) = 0; T(22) = (T(21) == 1); }

		if (T(17) == 1 || T(1) > 2000003) burst_end(m);
		else if (T(1) >= 15) {                       /* :1457-1503 */
			if (!T(4)) T(8)++; else { T(8) = 0; T(5) = 0; T(12) = 0; }
			T(1)++;
			if (T(4) < 2 && T(29) > 0 && T(14) == 4) {
				if (T(31) == 0 || T(31) == 1) { T(14) = 3; T(31)++; }
				else if (T(31) == 2) { T(14) = 0; T(15) = 0; T(31)++; }
			}
			if (T(14) == 5 && !T(35) && T(32) > 4 && T(32) < 8) { T(14) = 1; T(32)--; T(35)++; }
		}
		else burst_idle(m);

		if (T(8) > 6 && !T(4) && T(1) > 1 && T(1) < 15) {  /* :1875-1900 */
			PF_COV(34);
			T(5)++;
			if (T(5) < 35) {
				T(1) = 0;
				if (!T(13)) { PF_COV(35); T(12) = 1; T(13) = 1; }
				else { PF_COV(36); T(12) = 0; T(13)++; if (T(13) > 3) T(13) = 0; }
			}
			else { PF_COV(37); T(12) = 0; }
		}
		if (T(1) > 15 && T(1) < 1000000) { T(1) = 0; T(4) = 0; T(29)++; }
	}
	return act;
}

/* What machine_step_fast needs to know about the counters that only machine_step moves (the slow schedules' positions): worked out once
 * behind every machine_step, so that the fast step is a few dozen operations on the live counters. */
struct PfC {
	int fb14;      /* t14 is 1 or 2: a burst's first pair has bookkeeping to do (:931-947) */
	int t14_045;   /* t14 is 0, 4 or 5 (:884) */
	int gate14;    /* the gate of the slow schedules (:1532) but for t29 > 0 */
	int t6bad;     /* :1041-1051 is due */
	int capA, capB;        /* the cap's two schedule steps (:1466-1500): (t4 < 2 & t29 > 0 & capA) | capB */
	int schedA, schedB;    /* the t32 schedules (:1534-1709): (t4 < 2 & t1 == 15 & schedA) | schedB */
	int st_lt14;   /* the t28 schedule (:1711-1871) is not through */
	int arm0;      /* its arming step but for t1 > 13 (:1714) */
	int lim;       /* how far behind t33 the schedule's present stage waits */
	int exA, exB, exT;     /* what else it waits for: exA | (exB & t1 == exT) */
	int w8z;       /* w8 == 0 (:1506) */
	int wk;        /* the window (t10, t11): 1 = (8, 12), 2 = (10, 15) or (6, 9), 0 = anything else (the burst table knows the first three) */
};

DEVI void machine_cache(const PfM &m, PfC &c)
{
	if ((T(32) | T(28) | T(33) | T(35) | T(36) | T(39) | T(41)) == 0) {
		/* the slow schedules at rest (no picture has been seen to move them: DESIGN 2) -- what is left of the general form below */
		const int t14 = T(14), is4 = t14 == 4, is5 = t14 == 5;
		c.fb14 = (t14 == 1) | (t14 == 2);
		c.t14_045 = (t14 == 0) | is4 | is5;
		c.gate14 = is4 | is5;
		c.t6bad = T(6) > 4000000;
		c.capA = is4; c.capB = 0;
		c.schedA = is4; c.schedB = 0;
		c.st_lt14 = 1;
		c.arm0 = is5 & (T(31) > 0);
		c.lim = 10;
		c.exA = 0; c.exB = 0; c.exT = 12;
		c.w8z = Wv(8) == 0;
		c.wk = ((T(10) == 8) & (T(11) == 12)) | ((((T(10) == 10) & (T(11) == 15)) | ((T(10) == 6) & (T(11) == 9))) << 1);
		return;
	}
	const int t14 = T(14), t32 = T(32), st = T(28), t33 = T(33), t31 = T(31) > 2;
	const int is4 = t14 == 4, is5 = t14 == 5;
	c.fb14 = (t14 == 1) | (t14 == 2);
	c.t14_045 = (t14 == 0) | is4 | is5;
	c.gate14 = is4 | is5 | (T(39) == 2) | (T(41) > 0);
	c.t6bad = T(6) > 4000000;
	c.capA = is4;
	c.capB = is5 & (T(35) == 0) & (t32 > 4) & (t32 < 8);
	c.schedA = is4 | (is5 & (t32 > 2));
	c.schedB = (t32 == 4) | (t32 == 5) | (t32 == 7) | ((t32 == 6) & (T(36) < 118));
	c.st_lt14 = st < 14;
	c.arm0 = is5 & (st == 0) & (t33 == 0) & (T(31) > 0);
	{ const unsigned lim4 = st < 4 ? 0x03FF460Au : st < 8 ? 0x39361606u : st < 12 ? 0xB9746F54u : 0x0000BCBBu;   /* stages 0..13: 10 70 - 3 6 22 54 57 84 111 116 185 187 188 */
	  c.lim = (int)((lim4 >> (8 * (st & 3))) & 0xFF); }
	{ const int s0 = st == 0, s1 = st == 1, s3 = st == 3, s5 = st == 5;
	  c.exA = (s0 & (t33 > 0) & is4) | (s3 & t31) | !(s0 | s1 | s3 | s5);
	  c.exB = (s1 & is4) | (s5 & t31);
	  c.exT = s1 ? 11 : 12; }
	c.w8z = Wv(8) == 0;
	c.wk = ((T(10) == 8) & (T(11) == 12)) | ((((T(10) == 10) & (T(11) == 15)) | ((T(10) == 6) & (T(11) == 9))) << 1);
}

/* The pairs machine_step spends its time on, in short forms: each returns the answer, or -1 (counters untouched) where the pair needs
 * machine_step itself.  The chain runs on wave-uniform values, so these are scalar code with scalar branches.
 * Covered: a burst's first pair while t14 is not 1 or 2 and the strong-contrast bookkeeping (:850-873) is not due; inside a burst, a pair
 * that neither ends it (t17, :1053) nor makes one of the slow schedules of :1504-1873 move (with their gate open the t28 schedule may
 * count, :1714-1716).  On the synthetic images of the benchmark that is 99.6 % of the pairs at quality 10 and all of them at quality 1; on
 * white noise 80-90 %.  c: machine_cache() of the counters as they are. */
DEVI int machine_first_fast(PfM &m, const PfC &c, int code)          /* a burst's first pair (:840-994); t1 == 0 */
{
	const int f0 = code & 1, f1 = (code >> 1) & 1, g1 = (code >> 2) & 1, h0 = (code >> 3) & 1;
	if (c.fb14 | (f0 & h0)) return -1;
	int act = ACT_FIRST;
	if (f0 & (g1 | (T(8) == 1))) act |= ACT_ZERO0;
	if (f1 & (f0 | (T(12) == 1)) & c.t14_045) {                         /* the second pixel goes through the weak-first rule or its rotation (:884-930) */
		const int o3 = T(3);
		if ((o3 == 0) & f0) { act |= ACT_SUBST; T(3) = 1; }
		else T(3) = ((o3 == 1) << 1) | ((o3 == 2) * 3);                 /* 1 -> 2 -> 3 -> 0 */
	}
	T(2) = f0;
	if (f0 | f1) T(13) = 1;
	T(27) = 0;
	T(1) = 1;
	return act;
}
DEVI int machine_burst_fast(PfM &m, const PfC &c, int code)          /* inside a burst (:995-1910); t1 != 0 */
{
	const int fires = (code & 1) + ((code >> 1) & 1);
	int t1 = T(1) + fires, t4 = T(4) + fires, t8 = T(8), t5 = T(5), t12 = T(12), t44 = T(44), t29 = T(29), t30 = T(30);
	const int t18 = T(18);
	const int win = (t4 == T(10)) & (t1 == T(11));
	const int cyc = (t4 >= 10) & ((t4 > 10) | (t1 != 15));
	const int t17 = cyc ? (t18 == 0) : win;
	const int n18 = cyc ? (t18 == 0 ? 1 : t18 >= 15 ? 0 : t18 + 1) : t18;
	const int cap = t1 >= 15;
	const int few = (t4 < 2) & (t29 > 0);
	const int gate = (t29 > 0) & c.gate14;
	const int bad_cap = (few & c.capA) | c.capB;                      /* the cap (:1457-1503) without its two schedule steps */
	/* the plain idle step (:1504-1530); with the gate open (:1532) none of the three t32 schedules may be due, and the t28 schedule
	 * (:1711-1871) may only count */
	const int t1i = t44 < 3 ? t1 : t1 + 3;                              /* t1 behind the idle step */
	const int sched = ((t4 < 2) & (t1i == 15) & c.schedA) | c.schedB;
	const int counting = gate & c.st_lt14 & (t1i > 7);
	const int dist = t30 + 1 - T(33);
	const int stage = (c.arm0 & (t1i > 13)) | ((dist > c.lim) & (c.exA | (c.exB & (t1i == c.exT)))) | (dist == 9) | (t1i == 15);
	const int bad_idle = ((t1 == 6) & c.w8z) | (t44 < -90000) | (gate & sched) | (counting & stage);
	int bad_burst = c.t6bad | t17 | (t1 > 2000003) | (cap ? bad_cap : bad_idle);
	{
		const int keep = !cap | (t4 == 0);                            /* the cap clears t8, t5, t12 unless the burst had no hit */
		t8 = keep ? t8 + (cap & (t4 == 0)) : 0;
		t5 = keep ? t5 : 0;
		t12 = keep ? t12 : 0;
		t1 = cap ? t1 + 1 : t1i;
		t44 = cap ? t44 : (t44 < 3 ? t44 + 1 : 0);
		t30 += counting & !cap;
	}
	bad_burst |= (t8 > 6) & (t4 == 0) & (t1 > 1) & (t1 < 15);         /* the re-arm of :1875-1900 */
	const int wrap = (t1 > 15) & (t1 < 1000000);
	t1 = wrap ? 0 : t1; t4 = wrap ? 0 : t4; t29 += wrap;
	if (bad_burst) return -1;
	T(1) = t1; T(4) = t4; T(5) = t5; T(8) = t8; T(12) = t12; T(17) = 0; T(18) = n18; T(29) = t29; T(30) = t30; T(44) = t44;
	return 0;
}
DEVI int machine_step_fast(PfM &m, const PfC &c, int code) { return T(1) == 0 ? machine_first_fast(m, c, code) : machine_burst_fast(m, c, code); }

/* ---- a whole burst at a time ------------------------------------------------------------------------------------------------------
 * Inside a burst the fast form above does very little: hits add to t1 and t4, every fourth idle pair adds 3 to t1 (t44 counts them), and
 * the burst is over when t1 reaches 15 (the cap, :1457) or passes it (:1906).  Written out for the j-th pair behind the present one,
 * with h_j the hits of pairs 0..j:  t4_j = t4 + h_j,  t1_j = t1 + h_j + 3 floor((t44 + j) / 4)  -- no pair needs the one before it.  So
 * the pairs of a burst are evaluated side by side (burst_lane: one lane of the wavefront per pair, the hits from a prefix sum over the
 * row), the first pair that caps or wraps ends it (first set bit of a ballot), and everything machine_step_fast would have declined on the
 * way is a question about bit masks of those pairs (burst_commit).  A burst that is clean moves the counters in one go; one that is not
 * (or that runs past the end of the row) is left to the pair-by-pair forms.  Bit j of every mask = pair j behind the present one. */
struct PfBurstLane { int cap, wrap, win, cyc, i6, iS, cnt, g13, e15, eT; };
DEVI PfBurstLane burst_lane(int j, int t1_0, int t4_0, int v, int hits, int t10, int t11, int ext)
{
	PfBurstLane b;
	const int t4 = t4_0 + hits;
	const int t1 = t1_0 + hits + 3 * ((v + j) >> 2);                     /* behind the hits of pair j, before its idle step */
	const int t1i = t1 + ((((v + j) & 3) == 3) ? 3 : 0);                 /* behind its idle step */
	b.cap = t1 >= 15;
	b.wrap = t1i > 15;
	b.win = (t4 == t10) & (t1 == t11);                                  /* :1004 / :1039 */
	b.cyc = (t4 >= 10) & ((t4 > 10) | (t1 != 15));                      /* the pair rotates t18 instead of looking at the window (:1006-1037) */
	b.i6 = t1 == 6;                                                     /* :1506 */
	b.iS = (t4 < 2) & (t1i == 15);                                      /* :1534 */
	b.cnt = t1i > 7;                                                    /* :1711 */
	b.g13 = t1i > 13;                                                   /* :1714 */
	b.e15 = t1i == 15;
	b.eT = t1i == ext;
	return b;
}
struct PfBurstMasks { unsigned long long cap, wrap, win, cyc, i6, iS, cnt, g13, e15, eT; };
DEVI unsigned long long burst_low_bits(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }   /* bits 0 .. n-1 */
DEVI int burst_popc(unsigned long long x) { return __builtin_popcountll(x); }
/* can the counters go through burst_lane / burst_commit as they are?  (inside a burst, nothing pending that only machine_step knows) */
DEVI int burst_entry_ok(const PfM &m, const PfC &c)
{
	return (T(1) >= 1) & (T(1) < 15) & (T(4) >= 0) & (T(4) < 15) & (T(44) >= 0) & (T(44) <= 3) & !c.t6bad & (T(8) <= 6);
}
/* is the gate of the slow schedules (:1532) shut?  Then a burst can only be stopped by its window, by the t18 rotation or by the one-time
 * step of :1506, and burst_commit_quiet decides on five masks instead of ten. */
DEVI int burst_quiet(const PfM &m, const PfC &c) { return !((T(29) > 0) & c.gate14); }

/* k: the masks of burst_lane's answers for the pairs from the present one on; avail: how many of them exist in this row (1 .. 255);
 * hits_to(e): the hits of pairs 0 .. e (the caller has them per lane: one readlane).
 * Returns the number of pairs taken (counters moved), or 0 (counters untouched). */
template <class HITS>
DEVI int burst_commit(PfM &m, const PfC &c, const PfBurstMasks &k, int avail, HITS hits_to)
{
	const unsigned long long endm = k.cap | k.wrap;
	const int e_ = endm ? __builtin_ctzll(endm) : 64;
	const int open = e_ >= avail;                                       /* the row ends first: its last pairs are taken, the burst goes on in the next row */
	if (open && avail > 24) return 0;                                   /* (a burst is over within 21 pairs: not reached) */
	const int e = open ? avail - 1 : e_;
	const unsigned long long upto = burst_low_bits(e + 1);
	const int cap_end = open ? 0 : (int)((k.cap >> e) & 1);
	const unsigned long long idle = cap_end ? burst_low_bits(e) : upto; /* the pairs that take the idle step */
	const int t4e = T(4) + hits_to(e);                                  /* t4 behind pair e */
	const int v = T(44);
	if (k.win & ~k.cyc & upto) return 0;
	const int ncyc = burst_popc(k.cyc & upto), t18 = T(18);             /* pairs that rotate t18 (:1006-1037): 1 .. 15 -> 0, and at 0 the burst ends */
	if (ncyc && (t18 == 0 || ncyc > 16 - t18)) return 0;
	if (cap_end && ((((t4e < 2) & (T(29) > 0)) & c.capA) | c.capB)) return 0;
	if ((k.i6 & idle) && c.w8z) return 0;
	int counted = 0;
	if ((T(29) > 0) & c.gate14) {
		if (c.schedB) return 0;
		if ((k.iS & idle) && c.schedA) return 0;
		if (c.st_lt14) {
			const unsigned long long cm = k.cnt & idle;                 /* the pairs that count (t30) */
			if (cm) {
				const int n = burst_popc(cm), d0 = T(30) - T(33);       /* t30 - t33 at the i-th of them: d0 + i */
				if (c.arm0 && (cm & k.g13)) return 0;
				if (cm & k.e15) return 0;
				if (c.exA && d0 + n > c.lim) return 0;
				const unsigned long long mt = cm & k.eT;
				if (c.exB && mt) { const int last = 63 - __builtin_clzll(mt); if (d0 + burst_popc(cm & burst_low_bits(last + 1)) > c.lim) return 0; }
				if (9 - d0 >= 1 && 9 - d0 <= n) return 0;
				counted = n;
			}
		}
	}
	T(30) += counted;
	T(18) = (t18 + ncyc) & 15;
	T(17) = 0;
	if (open) {                                                         /* e + 1 idle pairs: t1 as burst_lane's t1i of pair e */
		T(1) += (t4e - T(4)) + 3 * ((v + e + 1) >> 2);
		T(4) = t4e;
		T(44) = (v + e + 1) & 3;
		return e + 1;
	}
	if (cap_end) {
		if (t4e == 0) T(8)++; else { T(8) = 0; T(5) = 0; T(12) = 0; }
		T(44) = (v + e) & 3;
	} else T(44) = 0;
	T(29)++; T(1) = 0; T(4) = 0;
	return e + 1;
}
/* A burst that was declined: the first of its pairs that ends it through t17 (:1053) -- a pair at the window (:1004 / :1039), or the pair that
 * finds the t18 rotation at 0 (:1006-1037: 16 - t18 rotating pairs take it there) -- or 64 if there is none before the burst's cap or wrap.
 * The pairs before that one can still go as an open burst (burst_commit with avail = the answer); the pair itself needs machine_step. */
DEVI int burst_t17_pair(const PfM &m, unsigned long long cap, unsigned long long wrap, unsigned long long win, unsigned long long cyc, int avail)
{
	const unsigned long long endm = cap | wrap;
	const int e_ = endm ? __builtin_ctzll(endm) : 63;
	const unsigned long long upto = burst_low_bits((e_ < avail ? e_ : avail - 1) + 1);
	const unsigned long long wm = win & ~cyc & upto;
	unsigned long long cy = cyc & upto;
	for (int skip = T(18) == 0 ? 0 : 16 - T(18); skip > 0 && cy; skip--) cy &= cy - 1;
	const int w1 = wm ? __builtin_ctzll(wm) : 64, w2 = cy ? __builtin_ctzll(cy) : 64;
	return w1 < w2 ? w1 : w2;
}
/* the same with the gate of the slow schedules shut (burst_quiet): masks of the first 32 pairs (a burst is over within 21) */
template <class HITS>
DEVI int burst_commit_quiet(PfM &m, const PfC &c, unsigned cap, unsigned wrap, unsigned win, unsigned cyc, unsigned i6, int avail, HITS hits_to)
{
	/* one decision, one exit: everything that can stop the burst goes into `bad` first (the compiler keeps the counters where they are
	 * instead of shuffling them at every early return) */
	const unsigned endm = cap | wrap;
	const int e_ = endm ? __builtin_ctz(endm) : 32;
	const int open = e_ >= avail;
	const int e = open ? avail - 1 : e_;
	const unsigned upto = 0xFFFFFFFFu >> ((31 - e) & 31);               /* bits 0 .. e */
	const int cap_end = open ? 0 : (int)((cap >> (e & 31)) & 1);
	const unsigned cy = cyc & upto;
	const int ncyc = __builtin_popcount(cy), t18 = T(18), v = T(44);
	int bad = open & (avail > 24);
	bad |= (win & ~cyc & upto) != 0;
	bad |= (ncyc != 0) & ((t18 == 0) | (ncyc > 16 - t18));
	bad |= c.w8z & ((i6 & (cap_end ? upto >> 1 : upto)) != 0);          /* the idle pairs: all of them, or all but the last */
	bad |= cap_end & c.capB;                                            /* (capA needs t29 > 0 and t14 == 4: the gate would be open) */
	if (bad) return 0;
	const int dh = hits_to(e);
	const int t4e = T(4) + dh;
	const int done = !open;                                             /* the burst ended: its cap or its wrap */
	const int zero_hit_cap = cap_end & (t4e == 0), hit_cap = cap_end & (t4e != 0);
	T(18) = (t18 + ncyc) & 15;
	T(17) = 0;
	T(8) = hit_cap ? 0 : T(8) + zero_hit_cap;
	T(5) = hit_cap ? 0 : T(5);
	T(12) = hit_cap ? 0 : T(12);
	T(44) = open ? (v + e + 1) & 3 : (cap_end ? (v + e) & 3 : 0);
	T(29) += done;
	T(1) = done ? 0 : T(1) + dh + 3 * ((v + e + 1) >> 2);
	T(4) = done ? 0 : t4e;
	return e + 1;
}

/* one pair of a burst in closed form, as the burst table sees it (table_entries below is checked against a walk with this, pair by pair):
 * lane j = pair j behind the burst's first one, hits = of pairs 0 .. j */
struct PfLaneD { int end, cap, wrap, cyc, win, i6; };
DEVI PfLaneD burst_lane_d(int j, int t1_0, int t4_0, int v, int hits, int t10, int t11)
{
	PfLaneD d;
	const int u = v + j;
	const int t4 = t4_0 + hits;
	const int t1 = t1_0 + hits + 3 * (u >> 2);                          /* behind the hits of pair j, before its idle step */
	const int t1i = t1 + (((u & 3) == 3) ? 3 : 0);                      /* behind its idle step */
	d.cap = t1 >= 15;
	d.wrap = t1i > 15;
	d.end = d.cap | d.wrap;
	d.cyc = (t4 >= 10) & ((t4 > 10) | (t1 != 15));
	d.win = (t4 == t10) & (t1 == t11) & !d.cyc;
	d.i6 = t1 == 6;
	return d;
}

/* ---- a burst's longest clean prefix, decided in the lanes (round 6) ---------------------------------------------------------------------
 * burst_commit above takes a whole burst or nothing: a burst with one pair that moves a slow schedule was walked pair by pair
 * (machine_step_fast, some sixty scalar instructions a pair), and with the gate of the slow schedules open that is most bursts -- the
 * pictures for which it stays open for long set the kernel's time.  Here every lane decides ITS pair exactly as machine_burst_fast would
 * find it (the counters in closed form, the counts of the lanes before it by v_mbcnt): does it end the burst, would it be declined; the
 * first lane that stops -- the burst's end, a pair that needs machine_step, the last pair before the stream's end -- packs the counters'
 * values into a word (before its pair if that pair is machine_step's, behind it otherwise), and the scalar unit reads that one word.
 * Every pair of a burst is then taken in a clean run or is a pair machine_step must see: nothing is walked pair by pair any more.
 *   gen_lane1    the pair's own tests (lane j = pair j behind the present one; hits: of pairs 0 .. j, own: of pair j alone)
 *   gen_lane2    with the counts of the lanes before it: is it declined, does it stop
 *   gen_word     bit 0 the pair is machine_step's (the word holds the counters BEFORE it), 1 the burst is over, 2 by its cap, 3 with a hit;
 *                4..8 t1, 9..13 t4, 14..15 t44, 16..19 t18, 20..26 what t30 has counted
 *   gen_take     the word into the counters; returns the number of pairs taken (0: the first pair already is machine_step's) */
#ifndef PF_BIT
#define PF_BIT(x) ((int)(x))           /* a test as 0 / 1 (the device pins it in a vector register: flags are combined on the vector unit, not as lane masks on the scalar unit) */
#endif
struct PfGenU { int t1_0, t4_0, v, t10, t11, t18, t29pos, t30, t33, avail; };   /* the counters' side (wave-uniform) */
struct PfGen1 { int t1, t1i, t4, u, own, cap, wrap, cyc, win, counting; };
DEVI PfGen1 gen_lane1(int j, const PfGenU &g, const PfC &c, int hits, int own)
{
	PfGen1 d;
	d.u = g.v + j;
	d.own = own;
	d.t4 = g.t4_0 + hits;
	d.t1 = g.t1_0 + hits + 3 * (d.u >> 2);                              /* behind the hits of pair j, before its idle step */
	d.t1i = g.t1_0 + hits + 3 * ((d.u + 1) >> 2);                       /* behind its idle step */
	d.cap = PF_BIT(d.t1 >= 15);
	d.wrap = PF_BIT(d.t1i > 15);
	d.cyc = PF_BIT(d.t4 > 10) | (PF_BIT(d.t4 == 10) & PF_BIT(d.t1 != 15));
	d.win = PF_BIT(d.t4 == g.t10) & PF_BIT(d.t1 == g.t11);
	const int gate = g.t29pos & c.gate14;
	d.counting = gate & c.st_lt14 & PF_BIT(d.t1i > 7) & (d.cap ^ 1);    /* an idle pair the t28 schedule counts (:1711) */
	return d;
}
struct PfGen2 { int bad, stop, t18j, cnt_before; };
DEVI PfGen2 gen_lane2(int j, const PfGenU &g, const PfC &c, const PfGen1 &d, int cyc_before, int cnt_before)
{
	PfGen2 r;
	r.t18j = (g.t18 + cyc_before) & 15;                                 /* t18 as the pair finds it, if no pair before it was stopped */
	r.cnt_before = cnt_before;
	const int t17 = d.cyc ? PF_BIT(r.t18j == 0) : d.win;                /* :1004-1039 */
	const int small4 = PF_BIT(d.t4 < 2);
	const int gate = g.t29pos & c.gate14;
	const int bad_cap = (small4 & g.t29pos & c.capA) | c.capB;          /* the cap's two schedule steps (:1466-1500) */
	const int e15 = PF_BIT(d.t1i == 15);
	const int sched = (small4 & e15 & c.schedA) | c.schedB;             /* the t32 schedules (:1534-1709) */
	const int dist = g.t30 + cnt_before + 1 - g.t33;
	const int stage = (c.arm0 & PF_BIT(d.t1i > 13)) | (PF_BIT(dist > c.lim) & (c.exA | (c.exB & PF_BIT(d.t1i == c.exT)))) | PF_BIT(dist == 9) | e15;   /* :1711-1871 */
	const int bad_idle = (PF_BIT(d.t1 == 6) & c.w8z) | (gate & sched) | (d.counting & stage);
	r.bad = t17 | (d.cap ? bad_cap : bad_idle);
	r.stop = r.bad | d.cap | d.wrap | PF_BIT(j == g.avail - 1) | PF_BIT(j == 63);
	return r;
}
DEVI unsigned gen_word(const PfGenU &g, const PfGen1 &d, const PfGen2 &r)
{
	const int end = (d.cap | d.wrap) & (r.bad ^ 1);
	/* the counters before the pair (it is machine_step's), or behind it (the burst's end: t1 = t4 = 0; a burst the stream cuts: as they stand) */
	const int t1 = r.bad ? d.t1 - d.own : end ? 0 : d.t1i;
	const int t4 = r.bad ? d.t4 - d.own : end ? 0 : d.t4;
	const int t44 = r.bad ? (d.u & 3) : d.cap ? (d.u & 3) : d.wrap ? 0 : ((d.u + 1) & 3);
	const int t18 = r.bad ? r.t18j : ((r.t18j + d.cyc) & 15);
	const int cnt = r.bad ? r.cnt_before : r.cnt_before + d.counting;
	return (unsigned)r.bad | ((unsigned)end << 1) | ((unsigned)(d.cap & end) << 2) | ((unsigned)PF_BIT(d.t4 != 0) << 3)
	     | ((unsigned)(t1 & 31) << 4) | ((unsigned)(t4 & 31) << 9) | ((unsigned)t44 << 14) | ((unsigned)t18 << 16) | ((unsigned)(cnt & 127) << 20);
}
/* can a burst go through the lanes as the counters are?  (burst_entry_ok: inside a burst, nothing pending that only machine_step knows) */
DEVI int gen_take(PfM &m, int lane_s, unsigned w)
{
	const int bad = (int)(w & 1u);
	const int n = lane_s + 1 - bad;
	if (n == 0) return 0;
	T(1) = (int)((w >> 4) & 31u); T(4) = (int)((w >> 9) & 31u); T(44) = (int)((w >> 14) & 3u); T(18) = (int)((w >> 16) & 15u); T(30) += (int)((w >> 20) & 127u);
	T(17) = 0;
	if (w & 2u) {
		if (w & 4u) { if (w & 8u) { T(8) = 0; T(5) = 0; T(12) = 0; } else T(8)++; }
		T(29)++;
	}
	return n;
}

/* ---- the burst table (round 6) -----------------------------------------------------------------------------------------------------
 * Nearly every burst starts the same way: behind a first pair, with t1 = 1, t4 = 0 and t44 = v in 0..3 -- and from there its course is a
 * function of the picture's codes alone: where it ends, whether by its cap, how often it rotates t18, whether a pair sits at the (8, 12)
 * window or at the one-time step of :1506.  So that function is TABULATED for every pair of the stream and all four v, in parallel, by a
 * kernel of its own on the vector units (k_low_table), and the chain (one wavefront a picture; what it pays for is scalar instructions,
 * 17 cycles each with sixteen chains to a CU's scalar unit, while a dependent LDS look-up costs 72) takes a first pair and its burst with
 * two look-ups and some forty scalar instructions.  Everything the table does not describe -- the gate of the slow schedules open, counters
 * elsewhere, an entry that says "not here" -- goes through the forms above.
 * Entry (16 bits) of first-pair position p and v; the burst's pairs are s = p + 1, s + 1, ..:
 *   bits 0..4  e: the burst's last pair is s + e
 *   bit 5 cap: it ends by its cap (t1 reaches 15 before the idle step; such a burst has a hit: t8, t5, t12 are cleared)
 *   bit 6 TAB_NONE: not described -- no end within the window, or behind the stream's end, or ncyc > 7, or the first pair itself has
 *         bookkeeping to do (its first cell above both thresholds, :850-873)
 *   bits 7..9 ncyc: how many of its pairs rotate t18 (:1006-1037);  bit 10: one of its pairs sits at the window (8, 12) (:1004 / :1039: it
 *   ends the burst through t17 if that is the window);  bit 11: one of its idle pairs has t1 == 6 (:1506)
 *   bits 12..15: the code of pair p itself (the first pair's)
 * The counters' side of the bargain is one word, tab_flags(): the bits of an entry that stop it as the counters are (TAB_NONE always; the
 * window bit if the window is (8, 12), the :1506 bit while w8 is 0, the cap bit while the cap's schedule step is due), or TAB_ALL if no
 * entry may be taken at all. */
#define TAB_NONE 64u
#define TAB_ALL 0xFFFFFFFFu
/* the four entries of one position.  g(j): hits of pairs s .. s + j (j in 0..31; whatever lies behind the stream's end counts as no hit);
 * first_ge(K): the first j in 0..31 with g(j) >= K, 32 if there is none;  navail: how many of the pairs s, s + 1, .. exist;  code: pair p's */
template <class G, class J>
DEVI void table_entries(G g, J first_ge, int navail, int code, unsigned out[4])
{
	/* Where the burst ends, without a walk: with f = (v + j) >> 2 the pair j caps if g(j) >= 14 - 3 f, and it wraps if g(j) >= 15 - 3 f' with
	 * f' = (v + j + 1) >> 2.  Inside one quarter (f fixed: four consecutive j) the first such pair is max(first_ge(14 - 3 f), the quarter's
	 * first j); the burst's end is the least of those over the quarters -- eleven look-ups that do not depend on v or on each other.  For v > 0
	 * the end is the end of v - 1 or the pair before it (the counters of (v, j) are those of (v - 1, j + 1) but for the hits of one pair). */
	const int j2 = first_ge(2), j3 = first_ge(3), j5 = first_ge(5), j6 = first_ge(6), j8 = first_ge(8), j9 = first_ge(9), j10 = first_ge(10);
	const int j11 = first_ge(11), j12 = first_ge(12), j14 = first_ge(14), j15 = first_ge(15);
	const int first_slow = (code & 1) & ((code >> 3) & 1);
	int e0 = 20;                                                        /* v = 0: pair 20 caps whatever the hits (f = 5) */
	{
		const int capk[5] = { j14, j11, j8, j5, j2 }, wrapk[6] = { j15, j12, j9, j6, j3, 0 };
		for (int f = 0; f < 5; f++) { const int lo = 4 * f, j = capk[f] > lo ? capk[f] : lo; e0 = (j <= lo + 3 && j < e0) ? j : e0; }
		for (int f = 0; f < 6; f++) { const int lo = 4 * f - 1 > 0 ? 4 * f - 1 : 0, j = wrapk[f] > lo ? wrapk[f] : lo; e0 = (j <= 4 * f + 2 && j < e0) ? j : e0; }
	}
	const int ga[4] = { g(e0), g(e0 > 0 ? e0 - 1 : 0), g(e0 > 1 ? e0 - 2 : 0), g(e0 > 2 ? e0 - 3 : 0) };   /* the hits up to the pairs the four ends can be */
	int e = e0;
	for (int v = 0; v < 4; v++) {
		if (v) {
			const int d = e0 - e + 1;                                   /* the pair before the end of v - 1 is e0 - d */
			const int gm = d == 1 ? ga[1] : d == 2 ? ga[2] : ga[3];
			const int jm = e - 1;
			const int pm = (e > 0) & (((1 + gm + 3 * ((v + jm) >> 2)) >= 15) | ((1 + gm + 3 * ((v + jm + 1) >> 2)) >= 16));
			e = pm ? e - 1 : e;
		}
		const int dd = e0 - e;
		const int ge = dd == 0 ? ga[0] : dd == 1 ? ga[1] : dd == 2 ? ga[2] : ga[3];
		const int t1e = 1 + ge + 3 * ((v + e) >> 2);
		const int cap = t1e >= 15;
		int ncyc = e >= j10 ? e - j10 + 1 : 0;
		ncyc -= (ge == 10) & (t1e == 15);
		const int min3a = (j9 - 1) < (7 - v) ? (j9 - 1) : (7 - v);
		const int win = (j8 > 4 - v ? j8 : 4 - v) <= (min3a < e ? min3a : e);
		const int idle_hi = e - cap;
		const int a_hi = (j6 - 1) < (3 - v) ? (j6 - 1) : (3 - v);
		const int b_hi = (j3 - 1) < (7 - v) ? (j3 - 1) : (7 - v);
		const int i6 = (j5 <= (a_hi < idle_hi ? a_hi : idle_hi)) | ((j2 > 4 - v ? j2 : 4 - v) <= (b_hi < idle_hi ? b_hi : idle_hi));
		const int none = (e >= navail) | (ncyc > 7) | first_slow;
		out[v] = (unsigned)e | ((unsigned)cap << 5) | (none ? TAB_NONE : 0u) | ((unsigned)(ncyc & 7) << 7) | ((unsigned)win << 10) | ((unsigned)i6 << 11) | ((unsigned)code << 12);
	}
}
/* which bits of an entry stop it as the counters are (t1 == 0: a first pair comes next) */
DEVI unsigned tab_flags(const PfM &m, const PfC &c)
{
	if (c.fb14 | c.t6bad | (T(8) > 6) | c.gate14 | !c.wk | (T(4) != 0) | ((unsigned)T(44) > 3u)) return TAB_ALL;   /* (the gate: whatever t29 is -- it is 0 only before the picture's first burst) */
	return TAB_NONE | (c.wk == 1 ? 1u << 10 : 0u) | (c.w8z ? 1u << 11 : 0u) | (c.capB ? 1u << 5 : 0u);
}
/* the first pair's rules (machine_first_fast without its decline: tab_flags and TAB_NONE cover that) as a table of 512 bytes:
 * index = the pair's code | t3 << 4 | (t8 == 1) << 6 | (t12 == 1) << 7 | t14_045 << 8;  value = the answer | t3 behind the pair << 4 */
DEVI unsigned first_lut(int idx)
{
	const int f0 = idx & 1, f1 = (idx >> 1) & 1, g1 = (idx >> 2) & 1, t3 = (idx >> 4) & 3, t8is1 = (idx >> 6) & 1, t12is1 = (idx >> 7) & 1, t14_045 = (idx >> 8) & 1;
	int act = ACT_FIRST | ((f0 & (g1 | t8is1)) ? ACT_ZERO0 : 0);
	const int rot = f1 & (f0 | t12is1) & t14_045;                        /* the second pixel goes through the weak-first rule or its rotation (:884-930) */
	const int subst = rot & (t3 == 0) & f0;
	const int n3 = !rot ? t3 : subst ? 1 : (((t3 == 1) << 1) | ((t3 == 2) * 3));
	act |= subst ? ACT_SUBST : 0;
	return (unsigned)act | ((unsigned)n3 << 4);
}
/* The chain's step at a first pair (t1 == 0) with the table: entry = the table's for this position and v = t44 (any entry if flags is
 * TAB_ALL), flags = tab_flags().  Returns -1 (counters untouched: the pair and its burst go through the other forms), or the first pair's
 * answer with the counters moved over the first pair and its whole burst; pairs: how many pairs that was. */
DEVI int table_take(PfM &m, unsigned flags, unsigned entry, int &pairs)
{
	if (entry & flags) return -1;
	const int ncyc = (int)((entry >> 7) & 7u), e = (int)(entry & 31u), code = (int)(entry >> 12);
	const int x = ((T(18) - 1) & 15) + ncyc;                            /* the t18 rotation would pass 0 (:1006-1037: that pair ends the burst through t17) */
	if (x > 15) return -1;
	const unsigned lv = first_lut(code | (T(3) << 4) | ((T(8) == 1) << 6) | ((T(12) == 1) << 7) | (((T(14) == 0) | (T(14) == 4) | (T(14) == 5)) << 8));
	T(3) = (int)((lv >> 4) & 3u);
	T(2) = code & 1;
	if (code & 3) T(13) = 1;
	T(27) = 0;
	T(18) = (x + 1) & 15;
	T(17) = 0;
	if (entry & 32u) { T(8) = 0; T(5) = 0; T(12) = 0; T(44) = (T(44) + e) & 3; }
	else T(44) = 0;
	T(29)++;
	pairs = 2 + e;
	return (int)(lv & 7u);
}

#undef T
#undef Wv

#endif
