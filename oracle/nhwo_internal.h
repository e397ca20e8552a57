/* nhwo_internal.h -- oracle internals.  TEST INFRASTRUCTURE ONLY (see nhwo.h). */
#ifndef NHWO_INTERNAL_H
#define NHWO_INTERNAL_H

#include <stdlib.h>
#include <string.h>
#include "nhwo.h"

#define W  512      /* luma plane row stride = reference 2*IM_DIM */
#define H  256      /* reference IM_DIM; chroma plane row stride */
#define Q  65536    /* reference IM_SIZE */
#define DEADZONE 8  /* reference `ratio` / m1 / m2: nhw_encoder_cli.c:177 select=8 */

static inline int iabs(int v) { return v < 0 ? -v : v; }

/* arena of zero-filled buffers, each between NHWO_GUARD zero bytes ("OOB = ZERO" model) */
typedef struct {
	uint8_t *base;
	size_t cap, used;
} nhwo_arena;

static inline void *arena_get(nhwo_arena *a, size_t bytes)
{
	size_t need = (bytes + 63) & ~(size_t)63;
	uint8_t *p;
	if (a->used + need + 2 * NHWO_GUARD > a->cap) return NULL;
	p = a->base + a->used + NHWO_GUARD;
	a->used += need + NHWO_GUARD; /* the trailing guard of one block is the leading guard of the next */
	return p;
}

/* position-list side stream (nhw_res1/3/5/6: list bytes, low-bit plane, payload words) */
typedef struct {
	uint8_t *list;  int list_len;
	uint8_t *bits;  int bits_len;
	uint8_t *word;  int word_len;
} nhwo_poslist;

typedef struct {
	int q;
	nhwo_arena arena;
	nhwo_trace *trace;

	/* planes (names follow the reference's image_buffer, codec.h:112-123) */
	int16_t *jpeg, *proc;           /* luma: 4*Q shorts each */
	int16_t *cjpeg, *cproc;         /* chroma: Q shorts each */
	uint8_t *pu, *pv;               /* 4:2:0 chroma planes, Q bytes each */
	int16_t *ll1;                   /* reference res256 (luma: Q shorts) */
	int16_t *l2save;                /* reference resIII (luma: Q shorts) */
	int16_t *cll1, *cl2save;        /* chroma res256 / resIII: Q/4 shorts */
	int16_t *keep;                  /* im_quality_setting: 2*Q shorts (q>=22) */
	int16_t *first_order;           /* im_wavelet_first_order: Q shorts (q>=22) */
	int16_t *band;                  /* im_wavelet_band: Q shorts (q>=22) */
	uint8_t *scan;                  /* im_nhw: 6*Q bytes */

	/* encode_state (codec.h:125-181) */
	uint8_t *ll_bytes;              /* first life of enc->tree1: 96*H+1 bytes of LL2 samples */
	uint8_t *ll_full;               /* enc->ch_res during the LL2 emission: Q/4 bytes */
	uint8_t *exw;  int exw_len;     /* exw_Y */
	uint8_t *res4; int res4_len;
	nhwo_poslist res1, res3, res5, res6;
	uint16_t *char_res1; int char_res1_len;
	uint32_t *qsetting3; int qsetting3_len;
	uint8_t *ll_comp;               /* highres_comp: Q/2 bytes */
	int ll_comp_y_len;              /* Y_res_comp */
	uint8_t *ll_word; int ll_word_len;        /* highres_word / highres_comp_len */
	uint16_t *ll_mem; int ll_mem_len;         /* highres_mem */
	uint8_t *ch_res;  int ch_res_len;         /* final ch_res / end_ch_res */
	uint8_t *res_u64, *res_v64;     /* 512 bytes each */
	int res_low, res_high;          /* setup->RES_LOW / RES_HIGH */
	int wavelet_type;               /* setup->wavelet_type after wavlts2packet (0 or 4) */
	int select1, select2;
	uint8_t *sel_word1, *sel_word2;
	uint32_t *packet;               /* enc->encode: 80000 words */
	int size_data1, size_data2;
	uint8_t *book1, *book2;         /* second life of tree1 / tree2: 708 bytes each */
	int size_book1, size_book2, tree_end;
	uint8_t book_tmp[600];          /* wavlts2packet's `codebook` scratch, shared by both parts */
	int res1_count, res3_count, res5_count; /* running nhw_resN_word_len counters of Y22/Y23 */
} nhwo_ctx;

/* trace */
void nhwo_trace_put(nhwo_trace *t, const char *name, int nblobs, const void **blobs, const uint32_t *lens);
static inline void trace_planes(nhwo_ctx *c, const char *name, const void *a, uint32_t la, const void *b, uint32_t lb)
{
	const void *bl[2]; uint32_t ln[2]; int n = 0;
	if (!c->trace) return;
	if (a) { bl[n] = a; ln[n++] = la; }
	if (b) { bl[n] = b; ln[n++] = lb; }
	nhwo_trace_put(c->trace, name, n, bl, ln);
}

/* stages */
void nhwo_prefilter_chroma(int16_t *plane, int quality);        /* pre_processing_UV */
void nhwo_dequant_sim_luma(nhwo_ctx *c, int part);              /* offsetY_recons256 */
void nhwo_dequant_sim_chroma(nhwo_ctx *c, int comp);            /* offsetUV_recons256 */
void nhwo_quantise_luma(nhwo_ctx *c);                           /* offsetY */
void nhwo_quantise_chroma(nhwo_ctx *c);                         /* offsetUV */
void nhwo_ll_code_luma(nhwo_ctx *c);                            /* Y_highres_compression */
void nhwo_ll_code_chroma(nhwo_ctx *c);                          /* highres_compression */
int  nhwo_packetise(nhwo_ctx *c);                               /* wavlts2packet */
void nhwo_band_recons(nhwo_ctx *c);                             /* im_recons_wavelet_band */
void nhwo_hq_settings(nhwo_ctx *c);                             /* wavelet_synthesis_high_quality_settings */
void nhwo_poslist_finish(nhwo_ctx *c, nhwo_poslist *pl, uint8_t *raw, int raw_len, const uint8_t *payload,
                         int payload_len, int word_mode);
int  nhwo_luma(nhwo_ctx *c);
int  nhwo_chroma(nhwo_ctx *c, int comp);
size_t nhwo_container(nhwo_ctx *c, uint8_t *out, size_t cap);   /* write_compressed_file */

#endif
