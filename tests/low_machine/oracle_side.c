/* Test harness, oracle side (C): the reference-following pre-filter passes of oracle/nhwo_prelow.c, opened up for the machine check.
 * TEST INFRASTRUCTURE ONLY. */
#include "../../oracle/nhwo_prelow.c"

int lm_machine_size(void) { return (int)sizeof(pf_machine); }
void lm_machine_reset(void *m) { machine_reset((pf_machine *)m); }
void lm_params(int q, int *sharp, int *sharp2) { const pf_params p = params_for(q); *sharp = p.sharp; *sharp2 = p.sharp2; }
void lm_contrast_map(const int16_t *src, int16_t *km, int q) { const pf_params p = params_for(q); contrast_map_low(src, km, &p); }
/* one pair through the oracle's machine, picture side included (km / y / so: the pair's cells) */
void lm_machine_pair(void *m, int q, int row, int16_t *km, int16_t *y, uint8_t *so)
{
	const pf_params p = params_for(q);
	int k0 = km[0], k1 = km[1];
	machine_pair((pf_machine *)m, &p, row, &k0, &k1, km, y, so);
}
