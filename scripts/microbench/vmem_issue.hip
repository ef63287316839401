// Developer microbenchmark: how long ONE wave (or W waves of a workgroup at once) needs to ISSUE a batch of independent vector-memory
// loads (cycle stamps before the first and after the last load instruction, not waiting for the data), for 11 x dword against
// 3 x dwordx4 per lane, and the LDS round trip of a uniform descriptor read (ds_read + readfirstlane) that precedes the fetch in the tile kernels.
//   hipcc --offload-arch=gfx950 -O3 -o vmem_issue vmem_issue.hip && ./vmem_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

template <int MODE> __global__ void k(const float *src, float *out, unsigned long long *stamps, int reps)
{
	__shared__ unsigned int desc[64];
	if (threadIdx.x < 64) desc[threadIdx.x] = threadIdx.x * 4096u;
	__syncthreads();
	float acc = 0.f;
	unsigned long long t_issue = 0, t_lds = 0;
	for (int r = 0; r < reps; r++)
	{
		__syncthreads();
		const unsigned long long t0 = now();
		const unsigned int base = __builtin_amdgcn_readfirstlane(desc[r & 63]);           // uniform LDS read -> SGPR
		asm volatile("" ::: "memory");
		const unsigned long long t1 = now();
		asm volatile("" ::: "memory");
		const float *p = src + base + (size_t)blockIdx.x * 262144 + threadIdx.x;
		float v[12];
		if (MODE == 0)
		{
#pragma unroll
			for (int i = 0; i < 11; i++) v[i] = p[i * 1024];       // 11 dword loads, 256 B per wave each
		}
		else
		{
			const float4 *q = reinterpret_cast<const float4 *>(src + base + (size_t)blockIdx.x * 262144) + threadIdx.x;
#pragma unroll
			for (int i = 0; i < 3; i++) { const float4 w = q[i * 1024]; v[4 * i] = w.x; v[4 * i + 1] = w.y; v[4 * i + 2] = w.z; v[4 * i + 3] = w.w; }
		}
		asm volatile("" ::: "memory");
		const unsigned long long t2 = now();
		t_lds += t1 - t0; t_issue += t2 - t1;
		if (MODE == 0) { for (int i = 0; i < 11; i++) acc += v[i]; } else { for (int i = 0; i < 12; i++) acc += v[i]; }
	}
	if (acc == 123.456f) out[0] = acc;
	if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t_lds; stamps[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = t_issue; }
}

template <int MODE> void run(int threads, const float *src, float *out)
{
	const int blocks = 256, reps = 200, waves = threads / 64;
	unsigned long long *d; (void)hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks * waves);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, src, out, d, reps);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, src, out, d, reps);
	(void)hipDeviceSynchronize();
	std::vector<unsigned long long> h(2 * blocks * waves);
	(void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
	std::vector<double> lds, iss;
	for (int i = 0; i < blocks * waves; i++) { lds.push_back((double)h[2 * i] / reps); iss.push_back((double)h[2 * i + 1] / reps); }
	std::sort(lds.begin(), lds.end()); std::sort(iss.begin(), iss.end());
	printf("%s, %d wave(s) per workgroup: uniform LDS read -> SGPR %.0f counts, issue of the loads %.0f counts (medians over waves; counts include one stamp ~ the E->A' figure)\n",
		MODE == 0 ? "11 x dword " : " 3 x dwordx4", waves, lds[lds.size() / 2], iss[iss.size() / 2]);
	(void)hipFree(d);
}
int main()
{
	float *src, *out; (void)hipMalloc(&src, (size_t)256 * 262144 * 4 + (1 << 24)); (void)hipMalloc(&out, 64);
	(void)hipMemset(src, 0, (size_t)256 * 262144 * 4 + (1 << 24));
	for (int t : { 64, 128, 256, 512 }) { run<0>(t, src, out); run<1>(t, src, out); }
	return 0;
}
