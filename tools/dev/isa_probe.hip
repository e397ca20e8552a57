// Developer probe (not product): semantics of a few gfx950 instructions the front kernel relies on, checked exhaustively on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__global__ void k_cvt(const float *in, uint32_t *out, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
__global__ void k_lerp(uint32_t *bad)
{
	// all byte pairs x both rounding bits
	int i = blockIdx.x * blockDim.x + threadIdx.x;      // 0..65535
	unsigned a = i & 255, b = i >> 8;
	unsigned A = a | (b << 8) | (a << 16) | (b << 24), B = b | (a << 8) | (b << 16) | (a << 24);
	unsigned r0 = __builtin_amdgcn_lerp(A, B, 0u), r1 = __builtin_amdgcn_lerp(A, B, 0x01010101u), r2 = __builtin_amdgcn_lerp(A, B, 0x00010001u);
	unsigned f = (a + b) >> 1, c = (a + b + 1) >> 1;
	if (r0 != (f * 0x01010101u)) atomicAdd(&bad[0], 1);
	if (r1 != (c * 0x01010101u)) atomicAdd(&bad[1], 1);
	if (r2 != (c | (f << 8) | (c << 16) | (f << 24))) atomicAdd(&bad[2], 1);
	// (x + 2y + z + 2) >> 2 == ceil_avg(floor_avg(x, z), y) for all bytes: third operand sweeps with blockIdx.y
	unsigned z = blockIdx.y;
	unsigned want = (a + 2 * b + z + 2) >> 2;
	unsigned got = __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(a, z, 0u), b, 0x01010101u) & 255;
	if (got != want) atomicAdd(&bad[3], 1);
}
__global__ void k_misc(uint32_t *out)
{
	out[0] = __builtin_amdgcn_sad_u8(0x01020304u, 0x04030201u, 100u);          // 3+1+1+3+100 = 108
	out[1] = __builtin_amdgcn_perm(0x11223344u, 0x55667788u, 0x07060100u);     // bytes: sel 0,1 from second operand (low), 6,7 from first
	out[2] = __builtin_amdgcn_alignbyte(0x11223344u, 0x55667788u, 1);           // ({hi,lo} >> 8) & 0xffffffff = 0x44556677
	out[3] = (uint32_t)__builtin_amdgcn_sdot2((short __attribute__((ext_vector_type(2)))){ 3, -4 }, (short __attribute__((ext_vector_type(2)))){ 1000, 2000 }, 7, false);
	out[4] = __builtin_amdgcn_udot4(0x01020304u, 0x05060708u, 9u, false);      // 4*8+3*7+2*6+1*5+9 = 79
	float f; asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(0x11C82233u)); out[5] = (uint32_t)f;   // 0xC8 = 200
}
int main()
{
	const int n = 1 << 20;
	float *hin = new float[n]; uint32_t *hout = new uint32_t[n];
	for (int i = 0; i < n; i++) hin[i] = -4.f + i * (264.f / n);   // -4 .. 260 in steps of 2.5e-4
	hin[0] = NAN; hin[1] = 1e9f; hin[2] = -1e9f; hin[3] = 254.5f; hin[4] = 255.5f; hin[5] = 0.5f; hin[6] = 1.5f; hin[7] = 2.5f; hin[8] = 0.99999f; hin[9] = 255.99f; hin[10] = 256.f;
	float *din; uint32_t *dout, *dbad;
	hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dbad, 64); hipMemset(dbad, 0, 64);
	hipMemcpy(din, hin, n * 4, hipMemcpyHostToDevice);
	k_cvt<<<n / 256, 256>>>(din, dout, n);
	hipMemcpy(hout, dout, n * 4, hipMemcpyDeviceToHost);
	int trunc_ok = 0, rne_ok = 0, other = 0, shell_bad = 0;
	for (int i = 11; i < n; i++) {
		const float v = hin[i]; const uint32_t r = (hout[i] >> 8) & 255;
		if ((hout[i] & 0xFFFF00FFu) != 0xAABB00DDu) shell_bad++;
		const float cl = v < 0 ? 0 : (v > 255 ? 255 : v);
		const uint32_t t = (uint32_t)cl, e = (uint32_t)nearbyintf(cl);
		if (r == t) trunc_ok++; if (r == e) rne_ok++; if (r != t && r != e) other++;
	}
	printf("cvt_pk_u8_f32: n=%d trunc-consistent %d rne-consistent %d neither %d shell_bad %d\n", n - 11, trunc_ok, rne_ok, other, shell_bad);
	for (int i = 0; i < 11; i++) printf("  in %g -> %u\n", hin[i], (hout[i] >> 8) & 255);
	k_lerp<<<dim3(256, 256), 256>>>(dbad);
	uint32_t bad[16]; hipMemcpy(bad, dbad, 64, hipMemcpyDeviceToHost);
	printf("lerp: floor-avg bad %u, ceil-avg bad %u, mixed bad %u, 121-filter identity bad %u\n", bad[0], bad[1], bad[2], bad[3]);
	k_misc<<<1, 1>>>(dout); hipMemcpy(hout, dout, 64, hipMemcpyDeviceToHost);
	printf("sad_u8 %u (108)  perm %08x  alignbyte %08x (44556677?)  sdot2 %d (-4993)  udot4 %u (79) ubyte2 %u (200)\n", hout[0], hout[1], hout[2], (int)hout[3], hout[4], hout[5]);
	return 0;
}
