// Developer microbenchmark: what bounds ONE sequential float sum of N values on a CU?  (the centre of the root's bounding sphere,
// pbdx_tetcontact_dev.h).  Variants: A registers only; B LDS reads, compiler-placed waits; C LDS reads by hand, counted waits;
// D = C on one lane; E = C + a workgroup barrier and restaging per 256 values; F = E + global loads;
// G every lane holds 32 consecutive values in its own registers, the running sum visits the lanes in turn (64 turns of 32 additions,
// v_mov_b32_dpp wave_ror:1 between turns): no operand instruction per value at all (the kernel's form).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o chain chain.hip && ./chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ void lds_read16(f4v &d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int B> __device__ __forceinline__ void rb(f4v (&v)[8], uint32_t a)
{
	lds_read16<128 * B + 0>(v[0], a); lds_read16<128 * B + 16>(v[1], a); lds_read16<128 * B + 32>(v[2], a); lds_read16<128 * B + 48>(v[3], a);
	lds_read16<128 * B + 64>(v[4], a); lds_read16<128 * B + 80>(v[5], a); lds_read16<128 * B + 96>(v[6], a); lds_read16<128 * B + 112>(v[7], a);
}
__device__ __forceinline__ void w8(f4v (&v)[8]) { asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])); }
__device__ __forceinline__ void w0(f4v (&v)[8]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])); }
__device__ __forceinline__ void c32(float &acc, const f4v (&v)[8]) {
#pragma unroll
	for (int q = 0; q < 8; q++) { acc += v[q].x; acc += v[q].y; acc += v[q].z; acc += v[q].w; } }
__device__ __forceinline__ void stage_hand(float &acc, uint32_t addr)
{
	f4v a[8], b[8];
	rb<0>(a, addr);
	rb<1>(b, addr); w8(a); c32(acc, a); rb<2>(a, addr); w8(b); c32(acc, b);
	rb<3>(b, addr); w8(a); c32(acc, a); rb<4>(a, addr); w8(b); c32(acc, b);
	rb<5>(b, addr); w8(a); c32(acc, a); rb<6>(a, addr); w8(b); c32(acc, b);
	rb<7>(b, addr); w8(a); c32(acc, a); w0(b); c32(acc, b);
}
__device__ __forceinline__ float turn_chunk(float acc, const float (&v)[32])
{
	for (int t = 0; t < 64; t++)
	{
#pragma unroll
		for (int k = 0; k < 32; k++) acc += v[k];
		acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x13C, 0xf, 0xf, false));
	}
	return acc;
}
// mode 0: A, 1: B, 2: C, 3: D, 4: E, 5: F, 6: G
__global__ __launch_bounds__(256) void k(const float4 *g, float *out, int stages, int mode)
{
	__shared__ __attribute__((aligned(16))) float comp[2][3][256];
	const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	for (int b = 0; b < 2; b++) for (int c = 0; c < 3; c++) comp[b][c][tid] = 1.0f + 1e-7f * tid;
	__syncthreads();
	float acc = 0.0f;
	float4 cur[6];
	for (int d = 0; d < 6; d++) cur[d] = g[d * 256 + tid];
	if (mode == 0)
	{
		if (wave < 3) { float v[8]; for (int q = 0; q < 8; q++) v[q] = comp[0][wave][q + lane];
			for (int st = 0; st < stages; st++) {
#pragma unroll
				for (int j = 0; j < 256; j++) acc += v[j & 7]; } }
	}
	else if (mode == 1)
	{
		if (wave < 3) for (int st = 0; st < stages; st++) { const float4 *src = (const float4 *)comp[st & 1][wave];
#pragma unroll 8
			for (int j = 0; j < 64; j++) { const float4 v = src[j]; acc += v.x; acc += v.y; acc += v.z; acc += v.w; } }
	}
	else if (mode == 2) { if (wave < 3) for (int st = 0; st < stages; st++) stage_hand(acc, (uint32_t)(uintptr_t)&comp[st & 1][wave][0]); }
	else if (mode == 3) { if (wave < 3 && lane == 0) for (int st = 0; st < stages; st++) stage_hand(acc, (uint32_t)(uintptr_t)&comp[st & 1][wave][0]); }
	else if (mode == 6)
	{
		if (wave < 3)
		{
			const float *gc = (const float *)g + (size_t)wave * stages * 256;      // component arrays
			float v[32], nx[32];
			for (int k = 0; k < 32; k++) v[k] = gc[lane * 32 + k];
			for (int c = 0; c < stages / 8; c++)                                   // 2048 values per chunk
			{
				for (int k = 0; k < 32; k++) nx[k] = gc[(size_t)(c + 1) * 2048 + lane * 32 + k];
				acc = turn_chunk(acc, v);
				for (int k = 0; k < 32; k++) v[k] = nx[k];
			}
		}
	}
	else
	{
		for (int st0 = 0; st0 < stages; st0 += 6)
#pragma unroll
			for (int d = 0; d < 6; d++)
			{
				const int st = st0 + d; if (st >= stages) break;
				const int buf = st & 1;
				comp[buf][0][tid] = cur[d].x; comp[buf][1][tid] = cur[d].y; comp[buf][2][tid] = cur[d].z;
				if (mode == 5) cur[d] = g[(size_t)(st + 6) * 256 + tid];
				__syncthreads();
				if (wave < 3) stage_hand(acc, (uint32_t)(uintptr_t)&comp[buf][wave][0]);
			}
	}
	if (lane == 0 && wave < 3) out[blockIdx.x * 3 + wave] = acc;
}
int main()
{
	const int stages = 1280;      // 327680 values per chain
	float4 *g; float *out;
	hipMalloc(&g, (size_t)(stages + 8) * 256 * sizeof(float4)); hipMemset(g, 0, (size_t)(stages + 8) * 256 * sizeof(float4)); hipMalloc(&out, 4096);
	const char *names[7] = { "A registers only", "B LDS, compiler waits", "C LDS by hand, counted waits", "D = C on one lane", "E = C + barrier + restage", "F = E + global loads", "G lanes in turn, own registers" };
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int blocks : { 1, 256 })
		for (int mode = 0; mode < 7; mode++)
		{
			hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, g, out, 16, mode);
			hipEventRecord(e0);
			hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, g, out, stages, mode);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			printf("%3d workgroup(s)  %-30s %.3f ms  = %.2f ns per addition of a chain\n", blocks, names[mode], ms, ms * 1e6 / (stages * 256.0));
		}
	return 0;
}
