/*
 * nhw_tail.hip -- kernels that run the order-dependent phases of the NHW encoder, one 256-thread workgroup
 * per image (see nhw_tail_par.h / nhw_tail_dev.h), plus the small block-copy kernel used between them.
 */
#include "nhw_tail_par.h"
#include "nhw_dwt.h"

using namespace nhw;

#ifndef NHW_DENSE_STREAM
#define NHW_DENSE_STREAM 0          /* 1: the luma byte stream is written (and rewritten by the dense Y31) next to the list -- a developer switch for comparing the two forms */
#endif
enum { PH_L1, PH_L2, PH_L3, PH_L4A, PH_C0, PH_C2, PH_C3, PH_C4, PH_C5, PH_FINAL, PH_L4B, PH_L4C, PH_L4D, PH_LLC, PH_L4C2 };

template <int PH>
__global__ __launch_bounds__(256) void k_phase(NhwWs ws, int comp, uint8_t *out, uint32_t *sizes, int32_t *status)
{
	__shared__ int sh_counts[2];
	__shared__ int sh_pos[2 * NT + 2];
	__shared__ uint32_t sh_z[4 * Q / 16 / 32 + 4];
	extern __shared__ __attribute__((aligned(16))) int16_t dyn_lds[];   /* LDS tiles of the row-serial passes (size chosen per phase at launch) */
	const int img = blockIdx.x, tid = threadIdx.x;
	Ctx c;
	ctx_load(&c, ws, img, comp);
	if (PH == PH_L1) luma_p1_par(&c, tid, sh_pos);
	else if (PH == PH_L2) luma_p2_par(&c, tid, dyn_lds);
	else if (PH == PH_L3) luma_p3_par(&c, tid, sh_pos, sh_counts, dyn_lds, ws.q <= 12 || ws.dbg != 0);
	else if (PH == PH_L4A) luma_p4a_par(&c, tid, dyn_lds);
	else if (PH == PH_L4B) luma_p4b_par(&c, tid, sh_pos, dyn_lds);
	else if (PH == PH_L4C) luma_p4c_par(&c, tid, sh_pos, dyn_lds, ws.q > 21 || ws.dbg);
	else if (PH == PH_L4D) luma_p4d_par(&c, tid, sh_counts, sh_z, dyn_lds, NHW_DENSE_STREAM || ws.dbg);
	else if (PH == PH_L4C2) luma_p4c2_par(&c, tid, reinterpret_cast<unsigned *>(sh_z), sh_pos);
	else if (PH == PH_LLC) { PROF_BEGIN(); ll_code_chroma_par(&c, tid, reinterpret_cast<uint8_t *>(dyn_lds)); if (!tid) PROF(&c, 18); }
	else if (PH == PH_C0) chroma_p0_par(&c, comp, tid);
	else if (PH == PH_C2) { chroma_ll1_neighbour(&c, tid); dequant_sim_chroma_par(&c, 1, tid); }
	else if (PH == PH_C3) chroma_p3_par(&c, comp, tid);
	else if (PH == PH_C4) dequant_sim_chroma_par(&c, 0, tid);
	else if (PH == PH_C5) chroma_p5_par(&c, comp, tid, dyn_lds, sh_counts, ws.dbg != 0);
}

/* Z2 + the container as a kernel of its own: exactly four wavefronts a SIMD.  Its two parts are two inlined copies of the same walks since the
 * chroma part became a list too, and left alone the register allocator took 158 registers for them (three wavefronts a SIMD: +0.35 ms); held
 * to 128 it spills 21 dwords in the code-book construction, off the walks. */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_final(NhwWs ws, uint8_t *out, uint32_t *sizes, int32_t *status)
{
	extern __shared__ __attribute__((aligned(16))) int16_t dyn_lds[];
	__shared__ PackShared sh_pack;
	const int img = blockIdx.x, tid = threadIdx.x;
	Ctx c;
	ctx_load(&c, ws, img);
	final_phase_par(&c, out + (size_t)img * (512u << 10), 512u << 10, &sizes[img], &status[img], &sh_pack, tid, reinterpret_cast<uint32_t *>(dyn_lds));
}

/* Y31 on the symbol list as a kernel of 512 threads (production; the stage checks run k_phase<L4D>, which has the dense form behind it) */
__global__ __launch_bounds__(512) void k_y31(NhwWs ws)
{
	extern __shared__ __attribute__((aligned(16))) int16_t dyn_lds[];
	__shared__ int sh_counts[2];
	Ctx c;
	ctx_load(&c, ws, blockIdx.x);
	PROF_BEGIN();
	scan_rewrite_list_par<512>(&c, threadIdx.x, reinterpret_cast<uint8_t *>(dyn_lds), sh_counts);
	if (!threadIdx.x) PROF(&c, 17);
}

/* Y19-Y23 of quality 17 .. 23 as a kernel of its own, held to eight wavefronts a SIMD (64 registers): with its 20 KB of LDS that is eight
 * workgroups a CU -- the 16 images a CU gets of a 4096-image batch in two rounds (k_phase<PH_L4A> takes 67 registers: seven, 7 + 7 + 2).
 * The low qualities' Y20 (thin_l1_low_par) would spill at 64 and stays on k_phase. */
#ifndef NHW_L4A_WAVES
#define NHW_L4A_WAVES 8
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NHW_L4A_WAVES, NHW_L4A_WAVES))) void k_l4a(NhwWs ws)
{
	extern __shared__ __attribute__((aligned(16))) int16_t dyn_lds[];
	Ctx c;
	ctx_load(&c, ws, blockIdx.x);
	luma_p4a_par(&c, threadIdx.x, dyn_lds);
}

/* passes that run one wavefront per image (nhw_tail_wave.h): four images per workgroup, no workgroup barriers */
enum { WV_DQ1, WV_DQ0, WV_EMIT, WV_QUANT };
template <int PH>
__global__ __launch_bounds__(256) void k_wave(NhwWs ws)
{
	const int img = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	__shared__ uint32_t dq_lut[(PH == WV_DQ1 || PH == WV_DQ0) ? DQ_WORDS : 1];
	if (PH == WV_DQ1 || PH == WV_DQ0) {                            /* the one workgroup barrier of these kernels: the table of the dequantiser walk */
		for (int i = threadIdx.x; i < DQ_WORDS; i += 256) dq_lut[i] = dq_entry(i);
		__syncthreads();
	}
	if (img >= ws.n) return;
	Ctx c;
	ctx_load(&c, ws, img);
	if (PH == WV_DQ1) wave_dequant_sim_luma(&c, 1, lane, dq_lut, false, ws.dbg != 0);
	else if (PH == WV_DQ0) wave_dequant_sim_luma(&c, 0, lane, dq_lut, !ws.dbg, ws.dbg != 0);   /* production: the level-2 block straight from l2save (Y17's restore of the work plane is not made: luma_p3_par) */
	else if (PH == WV_QUANT) {
		__shared__ __attribute__((aligned(16))) uint8_t park[4][16 * QROW];
		__shared__ uint32_t lut[4][QLUT + 3];
		PROF_BEGIN(); wave_quantise_luma(&c, lane, park[threadIdx.x >> 6], lut[threadIdx.x >> 6], ws.q > 21 || ws.dbg, NHW_DENSE_STREAM || ws.dbg, !(ws.q > 21 || ws.dbg)); if (!lane) PROF(&c, 15);
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

/* The middle of the first closed loop on one LDS residency of the 256 x 256 block: level-2 synthesis (wavelet_filterbank.c:305-496), Y8 (the
 * tags of the level-2 details nudge the reconstruction, nhw_encoder.c:183-216) and Y9 (LL1 pre-compensation, :218-279).  The reconstruction
 * is only ever read by Y9, and Y9 only hands on `ll1 + step` (the input of the analysis that follows): as three kernels the block went out
 * twice in two orientations, came back through Y8's tiles and Y9's rows and went out again (6.3 GB per 4096 images); here it never leaves
 * the LDS (2.2 GB: coefficients and LL1 in, LL1 without its tags and the pre-compensated LL1 out).
 *   * the synthesis leaves column c of the block as what the plane holds in row c: proc[y][x] = A[x][y];
 *   * a tag at (r, j) of the LL1 plane nudges proc[2(j-128)+1][2r], proc[2j][2(r-128)+1] or proc[2(j-128)+1][2(r-128)+1] (:205-213), i.e. a
 *     cell of ROW 2r / 2(r-128)+1 of the block: a wavefront takes an LL1 row, its targets are cells of one row of LDS, no two tags share one;
 *   * Y9 walks a row of proc = a column of the block (odd dword stride: no bank conflicts), a lane four cells, the step handed from lane to
 *     lane until nothing moves (precompensate_ll1_par).  Its two outer neighbours are cells of the planes outside the block.
 * One 1024-thread workgroup per CU works through the batch, the next block on its way in registers (k_dwt_syn).  The tests' stage checks
 * (nhw_debug_stop_after) run the three kernels instead, which leave every intermediate plane. */
__global__ __launch_bounds__(1024) void k_l2_recon(int16_t *__restrict__ jpegb, const int16_t *__restrict__ procb, size_t plane_stride, int16_t *__restrict__ ll1b, size_t ll1_stride, int n)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	constexpr int S = H, LS = S + 2, HLF = S / 2, PPL = HLF / 64, NT_ = 1024, NPRE = S * (S / 8) / NT_;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	int16_t *A = smem;
	int8_t *ytab = reinterpret_cast<int8_t *>(A + S * LS);         /* Y9's step by (difference, what it sees of its neighbours) */
	if (t < PRECOMP_TAB) ytab[t] = (int8_t)precomp_pick(t / 11 - 12, t % 11 - 5);
	uint4 pre[NPRE];
	if ((int)blockIdx.x < n) {
		const int16_t *src = jpegb + (size_t)blockIdx.x * plane_stride;
#pragma unroll
		for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(v / (S / 8)) * W + 8 * (v % (S / 8))); }
	}
	for (int img = blockIdx.x; img < n; img += gridDim.x) {
		int16_t *jp = jpegb + (size_t)img * plane_stride, *o = ll1b + (size_t)img * ll1_stride;
		const int16_t *p = procb + (size_t)img * plane_stride;
#pragma unroll
		for (int u = 0; u < NPRE; u++) {
			const int v = t + u * NT_, row = v / (S / 8), c8 = v % (S / 8);
			uint32_t *d = reinterpret_cast<uint32_t *>(A + row * LS + 8 * c8);
			d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
		}
		lds_barrier();
		if (img + (int)gridDim.x < n) {
			const int16_t *src = jpegb + (size_t)(img + gridDim.x) * plane_stride;
#pragma unroll
			for (int u = 0; u < NPRE; u++) { const int v = t + u * NT_; pre[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(v / (S / 8)) * W + 8 * (v % (S / 8))); }
		}
		/* the cells of the planes around the block that Y9 looks at, requested now -- lane l < 16: the cell of proc before row l of mine, lanes
		 * 16 .. 31: the one behind it; lane 32 / 33: the LL1 cell before my first row / behind my last one (rows of my neighbours).  Taking a
		 * tag off an LL1 cell is a function of the cell, so nobody waits for the wavefront that writes the clean value back. */
		const int r0 = wv * 16, c0 = 4 * lane;
		int side = 0;
		if (lane < 16) side = p[(size_t)(r0 + lane) * W - 1];
		else if (lane < 32) side = p[(size_t)(r0 + lane - 16) * W + H];
		else if (lane == 32) { side = o[(size_t)r0 * H - 1]; if (wv > 0) side = side > 14000 ? side - 16000 : side > 10000 ? side - 12000 : side; }   /* (what lies outside the plane is left as it is) */
		else if (lane == 33) { side = o[(size_t)(r0 + 16) * H]; if (wv < 15) side = side > 14000 ? side - 16000 : side > 10000 ? side - 12000 : side; }
		for (int i = 0; i < 16; i++) {                             /* synthesis, first direction, un-normalised */
			int16_t *x = A + (wv * 16 + i) * LS;
			int e[PPL], od[PPL];
#pragma unroll
			for (int u = 0; u < PPL; u++) syn_pair<S>(x, 1, lane + 64 * u, false, &e[u], &od[u]);
#pragma unroll
			for (int u = 0; u < PPL; u++) reinterpret_cast<uint32_t *>(x)[lane + 64 * u] = (uint32_t)(uint16_t)e[u] | ((uint32_t)(uint16_t)od[u] << 16);
		}
		lds_barrier();
		for (int i = 0; i < 16; i++) {                             /* second direction along the columns, normalised, in place */
			int16_t *x = A + wv * 16 + i;
			int e[PPL], od[PPL];
#pragma unroll
			for (int u = 0; u < PPL; u++) syn_pair<S>(x, LS, lane + 64 * u, true, &e[u], &od[u]);
#pragma unroll
			for (int u = 0; u < PPL; u++) { const int k = lane + 64 * u; x[(2 * k) * LS] = (int16_t)e[u]; x[(2 * k + 1) * LS] = (int16_t)od[u]; }
		}
		lds_barrier();
		auto ll1_row = [&](int i) { return *reinterpret_cast<const uint2 *>(o + (size_t)(r0 + (i < 16 ? i : 15)) * H + c0); };   /* row i of mine */
		auto untag = [](int v) { return v > 14000 ? v - 16000 : v > 10000 ? v - 12000 : v; };
		/* both walks keep a window of four rows in registers: the row in hand and the three behind it, requested three rows ahead of their use
		 * (rolled loops: unrolled, the compiler interleaves the rows and spills) */
		uint2 w0 = ll1_row(0), w1 = ll1_row(1), w2 = ll1_row(2), w3 = ll1_row(3);
#pragma unroll 1
		for (int i = 0; i < 16; i++) {                             /* Y8: LL1 row r, a lane four cells */
			const int r = r0 + i;
			int v[4];
			unpack4(w0, v);
			w0 = w1; w1 = w2; w2 = w3; w3 = ll1_row(i + 4);
			bool any = false;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int step = v[k] > 14000 ? 1 : v[k] > 10000 ? -1 : 0;
				if (!step) continue;
				v[k] = untag(v[k]);
				any = true;
				const int j = c0 + k;
				if (r < HLF && j >= HLF) A[(2 * r) * LS + 2 * (j - HLF) + 1] += (int16_t)step;
				else if (r >= HLF && j < HLF) A[(2 * (r - HLF) + 1) * LS + 2 * j] += (int16_t)step;
				else if (r >= HLF && j >= HLF) A[(2 * (r - HLF) + 1) * LS + 2 * (j - HLF) + 1] += (int16_t)step;
			}
			if (any) {
				uint2 w;
				w.x = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16); w.y = (uint32_t)(uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
				*reinterpret_cast<uint2 *>(o + (size_t)r * H + c0) = w;
			}
		}
		w0 = ll1_row(0); w1 = ll1_row(1); w2 = ll1_row(2); w3 = ll1_row(3);   /* (with or without their tags: whichever the memory system hands out) */
		lds_barrier();                                             /* the nudged block */
		int o_left = __builtin_amdgcn_readlane(side, 32);          /* the LL1 cell before the row in memory, without its tag */
#pragma unroll 1
		for (int i = 0; i < 16; i++) {                             /* Y9 */
			const int r = r0 + i;
			int pv[4], ov[4], d[4], st[4];
			unpack4(w0, ov);
			/* left of column 0 / right of column 255: the cells before and behind the row in memory, never updated */
			const int o_right = i == 15 ? __builtin_amdgcn_readlane(side, 33) : untag((int)(int16_t)(__builtin_amdgcn_readlane((int)w1.x, 0) & 0xFFFF));
			w0 = w1; w1 = w2; w2 = w3; w3 = ll1_row(i + 4);
#pragma unroll
			for (int k = 0; k < 4; k++) { ov[k] = untag(ov[k]); pv[k] = A[(c0 + k) * LS + r]; d[k] = (int16_t)(pv[k] - ov[k]); }
			const int p_left = __shfl(side, i), p_right = __shfl(side, 16 + i);
			const int my_edge = lane ? p_right - o_right : p_left - o_left;
			o_left = __builtin_amdgcn_readlane(ov[3], 63);
			const int sd = __shfl_down(d[0], 1), su = __shfl_up(d[3], 1);
			const int dn4 = lane < 63 ? sd : my_edge;                 /* the difference on the right of my last cell, as it was */
			const int first = lane ? su : my_edge;
			const int nb[4] = { precomp_right(d[1]), precomp_right(d[2]), precomp_right(d[3]), precomp_right(dn4) };
			int prev_in = first;
			for (;;) {
				int prev = prev_in;
#pragma unroll
				for (int k = 0; k < 4; k++) { st[k] = ytab[precomp_index(d[k], nb[k] + prev)]; prev = d[k] + st[k]; }
				int np = __shfl_up(prev, 1);
				if (!lane) np = first;
				if (!__any(np != prev_in)) break;
				prev_in = np;
			}
			uint2 w;
			w.x = (uint32_t)(uint16_t)(ov[0] + st[0]) | ((uint32_t)(uint16_t)(ov[1] + st[1]) << 16); w.y = (uint32_t)(uint16_t)(ov[2] + st[2]) | ((uint32_t)(uint16_t)(ov[3] + st[3]) << 16);
			*reinterpret_cast<uint2 *>(jp + (size_t)r * W + c0) = w;
		}
		lds_barrier();                                             /* the block is done with before the next one moves in */
	}
}
void nhw_launch_l2_recon(int16_t *jpeg, const int16_t *proc, size_t plane_stride, int16_t *ll1, size_t ll1_stride, int n, hipStream_t s)
{
	k_l2_recon<<<n < 256 ? n : 256, 1024, H * (H + 2) * sizeof(int16_t) + 288, s>>>(jpeg, proc, plane_stride, ll1, ll1_stride, n);
}

/* rows x cols block of shorts between two strided planes, every image of the batch: a workgroup an image, 16 bytes a thread and turn, four
 * turns in flight (cols and both pitches are multiples of 8, every row 16-byte aligned: asserted by the launcher).  Until round 5 a thread
 * moved ONE short and a workgroup one row: 1.6 ms for the 128 KB block of 4096 images (q <= 12), 0.7 TB/s. */
__global__ __launch_bounds__(256) void k_copy_block(const int16_t *__restrict__ src, size_t src_plane, int src_row,
                                                    int16_t *__restrict__ dst, size_t dst_plane, int dst_row, int rows, int cols)
{
	const int16_t *sp = src + (size_t)blockIdx.x * src_plane;
	int16_t *dp = dst + (size_t)blockIdx.x * dst_plane;
	const int per = cols >> 3, n = rows * per;
	for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
		uint4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) { const int i = i0 + u * 256; if (i < n) v[u] = reinterpret_cast<const uint4 *>(sp + (size_t)(i / per) * src_row)[i % per]; }
#pragma unroll
		for (int u = 0; u < 4; u++) { const int i = i0 + u * 256; if (i < n) reinterpret_cast<uint4 *>(dp + (size_t)(i / per) * dst_row)[i % per] = v[u]; }
	}
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
	case PH_L4A: return RF_LDS_BYTES > (NT + 2) * TLS * sizeof(int16_t) ? RF_LDS_BYTES : (size_t)(NT + 2) * TLS * sizeof(int16_t);
	case PH_L4B: return (size_t)(NT + 2) * TLS * sizeof(int16_t);
	case PH_L4C: return 0;                                         /* Y26 is pointwise, Y27 a wavefront per row straight on the plane */
	case PH_L4D: return SL_LDS_BYTES;                              /* Y31 on the symbol list: the non-zero map and the slices' value offsets in stream order (the dense form of the stage checks: 4608 bytes of them for its list of run starts) */
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
	{ const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_l2_recon), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(H * (H + 2) * sizeof(int16_t) + 288));
	  if (e_ != hipSuccess) { *where = "hipFuncSetAttribute(k_l2_recon, MaxDynamicSharedMemorySize)"; return (int)e_; } }
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
	case PH_L4A: if (ws.q >= 17) k_l4a<<<g, b, lds, s>>>(ws); else k_phase<PH_L4A><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4B: k_phase<PH_L4B><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4C: k_phase<PH_L4C><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4D: if (NHW_DENSE_STREAM || ws.dbg) k_phase<PH_L4D><<<g, b, lds, s>>>(ws, comp, out, sizes, status); else k_y31<<<g, 512, lds, s>>>(ws); break;
	case PH_LLC: k_phase<PH_LLC><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_L4C2: k_phase<PH_L4C2><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C0: k_phase<PH_C0><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C2: k_phase<PH_C2><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C3: k_phase<PH_C3><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C4: k_phase<PH_C4><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_C5: k_phase<PH_C5><<<g, b, lds, s>>>(ws, comp, out, sizes, status); break;
	case PH_FINAL: k_final<<<g, b, lds, s>>>(ws, out, sizes, status); break;
	}
}

void nhw_launch_copy_block(const int16_t *src, size_t src_plane, int src_row, int16_t *dst, size_t dst_plane, int dst_row,
                           int rows, int cols, int n, hipStream_t s)
{
	if ((cols | src_row | dst_row) & 7 || (src_plane | dst_plane) & 7 || ((uintptr_t)src | (uintptr_t)dst) & 15) { fprintf(stderr, "nhw_launch_copy_block: block not 16-byte aligned\n"); abort(); }
	k_copy_block<<<n, 256, 0, s>>>(src, src_plane, src_row, dst, dst_plane, dst_row, rows, cols);
}
