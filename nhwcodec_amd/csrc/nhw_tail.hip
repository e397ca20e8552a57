/*
 * nhw_tail.hip -- kernels that run the order-dependent phases of the NHW encoder, one 256-thread workgroup
 * per image (see nhw_tail_par.h / nhw_tail_dev.h), plus the small block-copy kernel used between them.
 */
#include "nhw_tail_par.h"

using namespace nhw;

enum { PH_L1, PH_L2, PH_L3, PH_L4A, PH_C0, PH_C2, PH_C3, PH_C4, PH_C5, PH_FINAL, PH_L4B, PH_L4C, PH_L4D, PH_LLC, PH_L4C2 };

template <int PH>
__global__ __launch_bounds__(256) void k_phase(NhwWs ws, int comp, uint8_t *out, uint32_t *sizes, int32_t *status)
{
	__shared__ int sh_counts[2];
	__shared__ int sh_pos[2 * NT + 2];
	__shared__ uint32_t sh_z[4 * Q / 16 / 32 + 4];
	extern __shared__ __attribute__((aligned(16))) int16_t dyn_lds[];   /* LDS tiles of the row-serial passes (size chosen per phase at launch) */
	__shared__ PackShared sh_pack;
	const int img = blockIdx.x, tid = threadIdx.x;
	Ctx c;
	ctx_load(&c, ws, img);
	if (PH == PH_L1) luma_p1_par(&c, tid, sh_pos);
	else if (PH == PH_L2) luma_p2_par(&c, tid, dyn_lds);
	else if (PH == PH_L3) luma_p3_par(&c, tid, sh_pos, sh_counts, dyn_lds);
	else if (PH == PH_L4A) luma_p4a_par(&c, tid, dyn_lds);
	else if (PH == PH_L4B) luma_p4b_par(&c, tid, sh_pos, dyn_lds);
	else if (PH == PH_L4C) luma_p4c_par(&c, tid, sh_pos, dyn_lds);
	else if (PH == PH_L4D) luma_p4d_par(&c, tid, sh_counts, sh_z, dyn_lds);
	else if (PH == PH_L4C2) luma_p4c2_par(&c, tid, reinterpret_cast<unsigned *>(sh_z), sh_pos);
	else if (PH == PH_LLC) { PROF_BEGIN(); ll_code_chroma_par(&c, tid, reinterpret_cast<uint8_t *>(dyn_lds)); if (!tid) PROF(&c, 18); }
	else if (PH == PH_C0) chroma_p0_par(&c, comp, tid);
	else if (PH == PH_C2) { chroma_ll1_neighbour(&c, tid); dequant_sim_chroma_par(&c, 1, tid); }
	else if (PH == PH_C3) chroma_p3_par(&c, comp, tid);
	else if (PH == PH_C4) dequant_sim_chroma_par(&c, 0, tid);
	else if (PH == PH_C5) chroma_p5_par(&c, comp, tid, dyn_lds, sh_counts, ws.dbg != 0);
	else if (PH == PH_FINAL) {
		final_phase_par(&c, out + (size_t)img * (512u << 10), 512u << 10, &sizes[img], &status[img], &sh_pack, tid, reinterpret_cast<uint32_t *>(dyn_lds));
	}
}

/* passes that run one wavefront per image (nhw_tail_wave.h): four images per workgroup, no workgroup barriers */
enum { WV_DQ1, WV_DQ0, WV_EMIT, WV_QUANT };
template <int PH>
__global__ __launch_bounds__(256) void k_wave(NhwWs ws)
{
	const int img = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (img >= ws.n) return;
	Ctx c;
	ctx_load(&c, ws, img);
	if (PH == WV_DQ1) wave_dequant_sim_luma(&c, 1, lane);
	else if (PH == WV_DQ0) wave_dequant_sim_luma(&c, 0, lane);
	else if (PH == WV_QUANT) {
		__shared__ __attribute__((aligned(16))) uint8_t park[4][16 * QROW];
		PROF_BEGIN(); wave_quantise_luma(&c, lane, park[threadIdx.x >> 6], ws.q > 21 || ws.dbg); if (!lane) PROF(&c, 15);
	}
	else if (PH == WV_EMIT) { PROF_BEGIN(); wave_emit_ll2(&c, lane); if (!lane) PROF(&c, 4); }
}
void nhw_launch_wave(int ph, const NhwWs &ws, hipStream_t s)
{
	const dim3 g((ws.n + 3) / 4), b(256);
	switch (ph) {
	case WV_DQ1: k_wave<WV_DQ1><<<g, b, 0, s>>>(ws); break;
	case WV_DQ0: k_wave<WV_DQ0><<<g, b, 0, s>>>(ws); break;
	case WV_EMIT: k_wave<WV_EMIT><<<g, b, 0, s>>>(ws); break;
	case WV_QUANT: k_wave<WV_QUANT><<<g, b, 0, s>>>(ws); break;
	}
}

/* rows x cols block of shorts between two strided planes, every image of the batch */
__global__ __launch_bounds__(256) void k_copy_block(const int16_t *__restrict__ src, size_t src_plane, int src_row,
                                                    int16_t *__restrict__ dst, size_t dst_plane, int dst_row, int rows, int cols)
{
	const int img = blockIdx.z, r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
	if (c < cols) dst[(size_t)img * dst_plane + (size_t)r * dst_row + c] = src[(size_t)img * src_plane + (size_t)r * src_row + c];
}

/* dynamic LDS per phase: number of 256-row column tiles (TLS shorts per row) the phase stages at once */
static size_t phase_lds(int ph)
{
	const size_t tile = (size_t)NT * TLS * sizeof(int16_t);
	switch (ph) {
	case PH_L2: return 3 * 32 * 33 + 32;                          /* the three 32 x 32 blocks of steps of Y8 (Y9 works on the plane itself) */
	case PH_L3: return LL_LDS_BYTES;
	case PH_LLC: return LLC_LDS_BYTES;
	case PH_FINAL: return PK_LDS_BYTES;
	case PH_L4A: return CR_LDS_BYTES > (NT + 2) * TLS * sizeof(int16_t) ? CR_LDS_BYTES : (size_t)(NT + 2) * TLS * sizeof(int16_t);
	case PH_L4B: return (size_t)(NT + 2) * TLS * sizeof(int16_t);
	case PH_L4C: return 0;                                         /* Y26 is pointwise, Y27 a wavefront per row straight on the plane */
	case PH_L4D: return 4608;                                      /* the list of run starts (at most one per 15 groups of the stream); the stream itself is written by the quantiser kernel */
	case PH_C5: return CQ_LDS_BYTES > 32 * 130 * 2 + (32 * 128 + 258) * 2 ? CQ_LDS_BYTES : 32 * 130 * 2 + (32 * 128 + 258) * 2;   /* the quantiser's parked rows; the marks' and the emission's tables */
	default: return 0;
	}
}

/* per device, from nhw_enc_create (see nhw_front_set_attrs) */
int nhw_tail_set_attrs(const char **where)
{
#define SETATTR(fn) do { const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&fn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10); \
                         if (e_ != hipSuccess) { *where = "hipFuncSetAttribute(" #fn ", MaxDynamicSharedMemorySize)"; return (int)e_; } } while (0)
	SETATTR(k_phase<PH_L1>); SETATTR(k_phase<PH_L2>); SETATTR(k_phase<PH_L3>); SETATTR(k_phase<PH_C5>);
#undef SETATTR
	return 0;
}

void nhw_launch_phase(int ph, const NhwWs &ws, int comp, uint8_t *out, uint32_t *sizes, int32_t *status, hipStream_t s)
{
	const dim3 g(ws.n), b(256);
	const size_t lds = phase_lds(ph);
	switch (ph) {
	case PH_L1: k_phase<PH_L1><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L2: k_phase<PH_L2><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L3: k_phase<PH_L3><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4A: k_phase<PH_L4A><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4B: k_phase<PH_L4B><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4C: k_phase<PH_L4C><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4D: k_phase<PH_L4D><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_LLC: k_phase<PH_LLC><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4C2: k_phase<PH_L4C2><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C0: k_phase<PH_C0><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C2: k_phase<PH_C2><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C3: k_phase<PH_C3><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C4: k_phase<PH_C4><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C5: k_phase<PH_C5><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_FINAL: k_phase<PH_FINAL><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	}
}

void nhw_launch_copy_block(const int16_t *src, size_t src_plane, int src_row, int16_t *dst, size_t dst_plane, int dst_row,
                           int rows, int cols, int n, hipStream_t s)
{
	k_copy_block<<<dim3((cols + 255) / 256, rows, n), 256, 0, s>>>(src, src_plane, src_row, dst, dst_plane, dst_row, rows, cols);
}
