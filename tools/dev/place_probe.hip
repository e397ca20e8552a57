// Developer probe (not product): where the dispatcher puts the workgroups of a launch that does not fill the chip.
// Every workgroup notes the XCD, shader engine and CU it runs on and stays busy for a while (so that all of them are resident together);
// the host prints how many workgroups each CU got.   build: hipcc --offload-arch=gfx950 -O3 -o tools/dev/place_probe tools/dev/place_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(128) void k_probe(uint32_t *out, int spin)
{
	extern __shared__ uint8_t lds[];
	uint32_t hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	int a = spin;
	for (int i = 0; i < spin; i++) asm volatile("s_add_i32 %0, %0, 1\n s_sleep 1" : "+s"(a) : : "scc");
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
	if (a == 123456789) lds[0] = 1;
}
int main(int argc, char **argv)
{
	const int lds = argc > 1 ? atoi(argv[1]) : 9616;
	uint32_t *d; hipMalloc(&d, 8 * 8192);
	hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
	for (int n : { 256, 512, 1024, 2048, 4096 }) {
		hipMemset(d, 0, 8 * 8192);
		k_probe<<<n, 128, lds>>>(d, 20000);
		hipDeviceSynchronize();
		std::vector<uint32_t> h(2 * n);
		hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
		std::map<uint32_t, int> per_cu;
		for (int i = 0; i < n; i++) { const uint32_t hw = h[2 * i], cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, x = h[2 * i + 1] & 15; per_cu[(x << 12) | (se << 8) | (sh << 4) | cu]++; }
		int hist[40] = { 0 }; for (auto &kv : per_cu) hist[kv.second < 39 ? kv.second : 39]++;
		printf("grid %4d, %d B of LDS: %zu CUs used; workgroups a CU -> CUs:", n, lds, per_cu.size());
		for (int k = 1; k < 40; k++) if (hist[k]) printf("  %d -> %d", k, hist[k]);
		printf("\n   first 24 workgroups (xcd.se.cu):");
		for (int i = 0; i < 24; i++) printf(" %u.%u.%u", h[2 * i + 1] & 15, (h[2 * i] >> 13) & 7, (h[2 * i] >> 8) & 15);
		printf("\n");
	}
	return 0;
}
