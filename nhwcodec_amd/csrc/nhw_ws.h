/* nhw_ws.h -- device workspace layout shared by the kernels of libnhwhip.so (gfx950 only). */
#ifndef NHW_WS_H
#define NHW_WS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define W  512      /* luma row stride  (reference 2*IM_DIM) */
#define H  256      /* chroma row stride (reference IM_DIM) */
#define Q  65536    /* reference IM_SIZE */
#define DEADZONE 8  /* reference `ratio`: nhw_encoder_cli.c:177 */
#define GUARD 4096  /* zero bytes kept behind every logical buffer: the reference's out-of-bounds reads
                        land here and return 0, the canonical-oracle semantics */

/* Per-image buffers are laid out structure-of-arrays over the batch: buffer b of image i lives at
 * base + off[b] + i * stride[b], stride[b] = size + GUARD rounded to 256 B, and GUARD zero bytes precede
 * image 0.  The guards are never written, so they stay zero between batches. */
enum {
	B_JPEG, B_PROC, B_PU, B_PV, B_CJPEG, B_CPROC, B_LL1, B_L2SAVE, B_CLL1, B_CL2SAVE, B_KEEP, B_FIRST, B_BAND,
	B_HS, B_KMAP, B_ROWMAP, B_ROWSTATE, B_SCAN, B_LLBYTES, B_LLFULL, B_EXW, B_LLCOMP, B_LLWORD, B_LLMEM, B_RES4,
	B_RAW, B_PAY, B_CC, B_HALF, B_TMP16,
	B_R1LIST, B_R1BITS, B_R1WORD, B_R3LIST, B_R3BITS, B_R3WORD, B_R5LIST, B_R5BITS, B_R5WORD,
	B_R6LIST, B_R6BITS, B_R6WORD, B_CHARRES, B_QSET3,
	B_RESU64, B_RESV64, B_PACKET, B_BOOK1, B_BOOK2, B_SEL1, B_SEL2, B_S1, B_S2, B_HIST, B_META, B_PROF, B_ROWFLAG, B_SEGMAP, B_STALE,
	B_NZQ, B_NZS, B_VOFF, B_VALS, B_CNZQ, B_CVALS,   /* the luma symbol stream as a list (nhw_tail_wave.h, wave_quantise_luma): non-zero map as the quantiser writes it, the map and the value offsets in stream order (Y31), the values; CNZQ / CVALS: the chroma part's map and values as the chroma quantiser leaves them */
	B_CJPEG_V, B_CPROC_V, B_CLL1_V, B_CL2SAVE_V, B_UBYTES,   /* the V plane's own copies of the four chroma work planes (production: NhwWs::split_chroma); UBYTES: where the chroma quantiser parks U's symbols until V's come (it used the band plane, and above q21 waited for Y29 to be through with it) */
	B_LOWTAB,   /* quality 1..16: the candidate masks pass A of the pre-filter leaves for its order-dependent cells (nhw_low.hip, k_low_pre -> k_low_mapfix): 320 bytes a row */
	B_COUNT
};

/* scalar per-image state (reference encode_state / codec_setup scalars), one struct per image in B_META */
struct NhwPosLens { int list_len, bits_len, word_len; };
struct NhwMeta {
	int exw_len, res4_len;
	NhwPosLens r1, r3, r5, r6;
	int char_res1_len, qsetting3_len;
	int ll_comp_y_len, ll_word_len, ll_mem_len, ch_res_len;
	int res_low, res_high, wavelet_type;
	int select1, select2;
	int size_data1, size_data2, size_book1, size_book2, tree_end;
	int status;
	int pad;
};

struct NhwWs {
	uint8_t *base;
	size_t off[B_COUNT];
	size_t stride[B_COUNT];
	int n;
	int q;
	int dbg;      /* the batch driver is stopped after a stage (tests): kernels also write the planes that nothing but a test reads */
	int split_chroma;     /* the V sequence works in planes of its own (B_*_V), so that its head -- everything up to the second dequantiser simulation -- runs right behind U's instead of behind U's quantiser, which waits for the luma tail's exception list; the stage checks keep the reference's one set of planes */
	int defer_verbatim;   /* the LL2 coder (Y16) runs beside the second dequantiser simulation: the samples it sent verbatim are put back by the synthesis behind both (k_dwt_syn) */
	int compat;   /* 0: canonical (out-of-bounds reads see zeros); 1: the heap neighbours of the stock one-image-per-process binary (nhw_hip.h) */
	template <typename T> __host__ __device__ T *buf(int b, int img) const { return (T *)(base + off[b] + (size_t)img * stride[b]); }
};


#endif
