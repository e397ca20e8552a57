// Developer probe (not product): what one instruction of a wavefront-per-image chain costs on gfx950, by kind and by how many wavefronts share a CU.
// Each kernel runs a loop body of 64 instructions of one kind n times; the host times it with 1, 4, 8 and 16 wavefronts a CU (256 CUs) and
// prints cycles per instruction and wavefront (at 2.4 GHz).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/dev/issue_probe tools/dev/issue_probe.hip      run on the GPU box: tools/dev/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
__global__ __launch_bounds__(64) void k_salu_dep(int n, int *out)           /* 64 dependent scalar adds */
{
	int a = n, b = 3;
	for (int i = 0; i < n; i++) asm volatile(REP64("s_add_i32 %0, %0, %1\n") : "+s"(a) : "s"(b) : "scc");
	if (a == 12345) out[0] = a;
}
__global__ __launch_bounds__(64) void k_salu_ind(int n, int *out)           /* 64 scalar adds on four independent chains */
{
	int a = n, b = 3, c = 5, d = 7, e = 9;
	for (int i = 0; i < n; i++) asm volatile(REP16("s_add_i32 %0, %0, %4\n s_add_i32 %1, %1, %4\n s_add_i32 %2, %2, %4\n s_add_i32 %3, %3, %4\n") : "+s"(a), "+s"(c), "+s"(d), "+s"(e) : "s"(b) : "scc");
	if (a + c + d + e == 12345) out[0] = a;
}
__global__ __launch_bounds__(64) void k_valu_dep(int n, int *out)           /* 64 dependent vector adds */
{
	int a = n + threadIdx.x, b = 3;
	for (int i = 0; i < n; i++) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
	if (a == 12345) out[0] = a;
}
__global__ __launch_bounds__(64) void k_valu_ind(int n, int *out)           /* 64 vector adds on four independent chains */
{
	int a = n + threadIdx.x, b = 3, c = 5, d = 7, e = 9;
	for (int i = 0; i < n; i++) asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
	if (a + c + d + e == 12345) out[0] = a;
}
__global__ __launch_bounds__(64) void k_mixed_dep(int n, int *out)          /* scalar add -> vector add of it -> readfirstlane -> scalar add: 16 x 4 instructions, each waiting for the other unit */
{
	int a = n, v = threadIdx.x;
	for (int i = 0; i < n; i++) asm volatile(REP16("s_add_i32 %0, %0, 1\n v_add_u32 %1, %0, %1\n v_readfirstlane_b32 %0, %1\n s_and_b32 %0, %0, 0xffff\n") : "+s"(a), "+v"(v) : : "scc");
	if (a == 12345) out[0] = a + v;
}
__global__ __launch_bounds__(64) void k_cmp_mask(int n, int *out)           /* v_cmp -> s_and_b64 of the mask -> v_cndmask: the lane-mask ping-pong, 16 x 4 */
{
	int v = threadIdx.x, w = 1;
	for (int i = 0; i < n; i++) asm volatile(REP16("v_cmp_lt_i32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, 1, %0\n") : "+v"(v) : "v"(w) : "vcc", "scc");
	if (v == 12345) out[0] = v;
}
__global__ __launch_bounds__(64) void k_branchy(int n, int *out)            /* 16 x (compare, taken branch over one instruction, add, add): what a chain of ifs costs */
{
	int a = n, b = 0;
	for (int i = 0; i < n; i++) asm volatile(REP16("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1\n s_add_i32 %1, %1, 7\n s_add_i32 %0, %0, 1\n") : "+s"(a), "+s"(b) : : "scc");
	if (a + b == 12345) out[0] = a;
}
__global__ __launch_bounds__(64) void k_lds_dep(int n, int *out)            /* a dependent LDS round trip by one address (v_mov, ds_read, wait, readfirstlane): 16 x 4 */
{
	__shared__ int tab[256];
	for (int i = threadIdx.x; i < 256; i += 64) tab[i] = (i * 4 + 4) & 1023;
	__syncthreads();
	int a = 0, v;
	for (int i = 0; i < n; i++) asm volatile(REP16("v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1\n") : "+s"(a), "=&v"(v) : : "memory");
	if (a == 12345) out[0] = a;
}
template <class K> static void run(const char *name, K kern, int *d_out)
{
	const int n = 4000;
	printf("%-12s", name); fflush(stdout);
	for (int w : { 1, 4, 8, 16 }) {
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		kern<<<256 * w, 64>>>(10, d_out);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		kern<<<256 * w, 64>>>(n, d_out);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		printf("  %2d waves/CU: %6.2f cyc/instr", w, ms * 1e-3 * 2.4e9 / (64.0 * n)); fflush(stdout);
	}
	printf("\n");
}
int main()
{
	int *d_out; hipMalloc(&d_out, 64);
	run("salu dep", k_salu_dep, d_out); run("salu ind", k_salu_ind, d_out); run("valu dep", k_valu_dep, d_out); run("valu ind", k_valu_ind, d_out);
	run("mixed dep", k_mixed_dep, d_out); run("cmp mask", k_cmp_mask, d_out); run("branchy", k_branchy, d_out); run("lds dep", k_lds_dep, d_out);
	return 0;
}
