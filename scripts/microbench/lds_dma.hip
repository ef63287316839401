// Semantics check of global_load_lds_dwordx4 on gfx950: does lane l of a wave land at M0 base + l * 16 ?
// build: hipcc --offload-arch=gfx950 -O3 lds_dma.hip -o lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4 *src, const unsigned *gid, float4 *dst, unsigned n)
{
	extern __shared__ float4 lds[];
	for (unsigned base = threadIdx.x; base < n; base += 4 * blockDim.x)
	{
		unsigned g[4];
#pragma unroll
		for (unsigned q = 0; q < 4; q++) { const unsigned i = base + q * blockDim.x; g[q] = gid[i < n ? i : n - 1]; }
#pragma unroll
		for (unsigned q = 0; q < 4; q++)
		{
			const unsigned i = base + q * blockDim.x;
			if (i < n)
				__builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + g[q]),
					(void __attribute__((address_space(3))) *)(lds + (i & ~63u)), 16, 0, 0);
		}
	}
	__syncthreads();
	for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = lds[i];
}
int main()
{
	const unsigned n = 7001, m = 100000;
	std::vector<float4> src(m);
	std::vector<unsigned> gid(n);
	for (unsigned i = 0; i < m; i++) src[i] = make_float4((float)i, i + 0.25f, i + 0.5f, i + 0.75f);
	for (unsigned i = 0; i < n; i++) gid[i] = (i * 7919u + 13u) % m;
	float4 *dsrc, *ddst; unsigned *dgid;
	hipMalloc(&dsrc, m * 16); hipMalloc(&ddst, n * 16); hipMalloc(&dgid, n * 4);
	hipMemcpy(dsrc, src.data(), m * 16, hipMemcpyHostToDevice);
	hipMemcpy(dgid, gid.data(), n * 4, hipMemcpyHostToDevice);
	hipMemset(ddst, 0xff, n * 16);
	hipLaunchKernelGGL(k, dim3(1), dim3(1024), 8192 * 16, 0, dsrc, dgid, ddst, n);
	std::vector<float4> out(n);
	hipMemcpy(out.data(), ddst, n * 16, hipMemcpyDeviceToHost);
	unsigned bad = 0;
	for (unsigned i = 0; i < n; i++)
	{
		const float4 e = src[gid[i]];
		if (out[i].x != e.x || out[i].y != e.y || out[i].z != e.z || out[i].w != e.w) { if (bad < 5) printf("mismatch at %u: got %g %g %g %g expected %g %g %g %g\n", i, out[i].x, out[i].y, out[i].z, out[i].w, e.x, e.y, e.z, e.w); bad++; }
	}
	printf("lds_dma: %u mismatches of %u (%s)\n", bad, n, hipGetErrorString(hipGetLastError()));
	return bad != 0;
}
