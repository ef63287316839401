// Developer microbenchmark: issue rate of plain vs packed fp32 VALU instructions on gfx950 at the fused
// kernel's occupancy (1024-thread workgroup = 4 waves per SIMD).  Settles whether the colour sweep's
// VALU instruction count or something else bounds it.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(1024) void k(float *out, int iters, float seed)
{
	float a[8]; f2 p[8];
	for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i].x = a[i]; p[i].y = a[i] * 0.5f; }
	const float c = seed * 0.999f; const f2 cc = { c, c * 1.001f };
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int i = 0; i < 8; i++)
			{
				if (MODE == 0) a[i] = a[i] + c;                               // v_add_f32
				else if (MODE == 1) a[i] = __builtin_fmaf(a[i], c, c);        // v_fma_f32
				else if (MODE == 2) p[i] = p[i] + cc;                         // v_pk_add_f32
				else if (MODE == 3) p[i] = __builtin_elementwise_fma(p[i], cc, cc);   // v_pk_fma_f32
				else a[i] = a[i] * c;                                         // v_mul_f32
			}
	}
	float s = 0; for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
	if (s == 12345.678f) out[0] = s;
}

template <int MODE> void run(const char *name, int flops_per_lane_instr)
{
	float *d; hipMalloc(&d, 64);
	const int iters = 2000, blocks = 256 * 4;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, 10, 1.0f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, iters, 1.0f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double wave_instr = (double)blocks * 16 * iters * 64;        // 64 VALU instructions per iteration per wave
	const double per_simd_per_s = wave_instr / (256.0 * 4) / (ms * 1e-3);
	printf("%-14s %.3f ms  %.3f wave-instr/ns/SIMD -> %.2f cycles per wave64 instruction @2.4GHz, %.1f TFLOP/s\n", name, ms,
		per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s, wave_instr * 64 * flops_per_lane_instr / (ms * 1e-3) / 1e12);
	hipFree(d);
}
int main()
{
	run<0>("v_add_f32", 1); run<4>("v_mul_f32", 1); run<1>("v_fma_f32", 2); run<2>("v_pk_add_f32", 2); run<3>("v_pk_fma_f32", 4);
	return 0;
}
