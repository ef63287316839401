// Developer microbenchmark: VALU issue interval seen by ONE wave alone on its SIMD (the situation of a latency-bound
// colour step: a tile's step has fewer slots than lanes), as a function of the instruction-level parallelism in
// its stream: ILP = number of independent dependency chains interleaved in program order.  Compare with W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_single valu_single.hip && ./valu_single
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int ILP> __global__ __launch_bounds__(1024) void k(float *out, int iters, float seed)
{
	float a[ILP];
	for (int i = 0; i < ILP; i++) a[i] = seed + i + threadIdx.x;
	const float c = seed * 0.999f;
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int r = 0; r < 64 / ILP; r++)
#pragma unroll
			for (int i = 0; i < ILP; i++) a[i] = (r & 1) ? a[i] * c : a[i] + c;      // v_mul_f32 / v_add_f32, alternating
	}
	float s = 0; for (int i = 0; i < ILP; i++) s += a[i];
	if (s == 12345.678f) out[0] = s;
}

template <int ILP> void run(int threads)
{
	float *d; hipMalloc(&d, 64);
	const int iters = 20000, blocks = 256;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double instr_per_wave = (double)iters * 64;
	printf("ILP %d, %4d threads/WG (%d wave(s) per SIMD): %.3f ms -> %.2f ns per instruction of one wave = %.2f cycles @2.4GHz; per SIMD %.2f cycles/instr\n",
		ILP, threads, threads / 256 ? threads / 256 : 1, ms, ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave * 2.4,
		ms * 1e6 / instr_per_wave * 2.4 / (threads >= 256 ? threads / 256 : 1));
	hipFree(d);
}
int main()
{
	for (int t : { 64, 256, 512, 1024 })
	{
		run<1>(t); run<2>(t); run<4>(t); run<8>(t);
	}
	return 0;
}
