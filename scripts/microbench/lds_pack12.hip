// Questions behind the 12-byte LDS position layout (x, y, z per particle, no inverse mass), gfx950:
//  (1) does global_load_lds_dwordx3 put lane l of a wave at M0 base + 12 l, reading 12 bytes at the lane's address?
//  (2) do ds_read_b96 / ds_write_b96 work at 4-byte aligned addresses (12 h), and what do they cost next to the 16-byte forms?
// build: hipcc --offload-arch=gfx950 -O3 lds_pack12.hip -o lds_pack12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct f3 { float x, y, z; };
__global__ void k_dma(const float4 *src, const unsigned *gid, f3 *dst, unsigned n)
{
	extern __shared__ float lds[];
	for (unsigned base = threadIdx.x; base < n; base += blockDim.x)
	{
		const unsigned i = base, g = gid[i];
		const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(lds + 3u * (i & ~63u)));
		const unsigned boff = g * 16u;
		unsigned saved;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_mov_b32 m0, %0"
			: "=&s"(saved) : "v"(boff), "s"(src), "s"(m0v) : "memory");
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (unsigned i = threadIdx.x; i < n; i += blockDim.x) { f3 v; v.x = lds[3 * i]; v.y = lds[3 * i + 1]; v.z = lds[3 * i + 2]; dst[i] = v; }
}
// gather / scatter through LDS: MODE 0 = float4 slots (ds_read_b128 / ds_write_b96 as the engine has them), 1 = 12-byte slots with b96 accesses,
// 2 = split planes: (x, y) as 8-byte slots (b64) + z as 4-byte slots (b32), every access naturally aligned
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k_gs(const unsigned *idx, float *out, unsigned n_local, unsigned iters, unsigned long long *cycles)
{
	extern __shared__ float lds[];
	const unsigned stride = MODE ? 3u : 4u;
	for (unsigned i = threadIdx.x; i < n_local * stride; i += blockDim.x) lds[i] = (float)i;
	__syncthreads();
	unsigned h0 = idx[threadIdx.x], h1 = idx[threadIdx.x + blockDim.x];
	const unsigned zbase = n_local * 8u;      // MODE 2: z plane behind the xy plane
	float acc = 0.f;
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (unsigned it = 0; it < iters; it++)
	{
		float ax, ay, az, bx, by, bz;
		if (MODE == 1)
		{
			v3f a, b;
			asm volatile("ds_read_b96 %0, %1" : "=v"(a) : "v"(h0 * 12u));
			asm volatile("ds_read_b96 %0, %1" : "=v"(b) : "v"(h1 * 12u));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			ax = a.x; ay = a.y; az = a.z; bx = b.x; by = b.y; bz = b.z;
		}
		else if (MODE == 2)
		{
			v2f a, b; float c, d;
			asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(h0 * 8u));
			asm volatile("ds_read_b32 %0, %1" : "=v"(c) : "v"(zbase + h0 * 4u));
			asm volatile("ds_read_b64 %0, %1" : "=v"(b) : "v"(h1 * 8u));
			asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(zbase + h1 * 4u));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			ax = a.x; ay = a.y; az = c; bx = b.x; by = b.y; bz = d;
		}
		else
		{
			v4f a, b;
			asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(h0 * 16u));
			asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(h1 * 16u));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			ax = a.x; ay = a.y; az = a.z; bx = b.x; by = b.y; bz = b.z;
		}
		const float dx = ax - bx, dy = ay - by, dz = az - bz;
		const float s = 1e-3f * (dx + dy + dz);
		acc += s;
		v3f wa, wb; wa.x = ax + s; wa.y = ay + s; wa.z = az + s; wb.x = bx - s; wb.y = by - s; wb.z = bz - s;
		if (MODE == 2)
		{
			v2f pa, pb; pa.x = wa.x; pa.y = wa.y; pb.x = wb.x; pb.y = wb.y;
			asm volatile("ds_write_b64 %0, %1" :: "v"(h0 * 8u), "v"(pa) : "memory");
			asm volatile("ds_write_b32 %0, %1" :: "v"(zbase + h0 * 4u), "v"(wa.z) : "memory");
			asm volatile("ds_write_b64 %0, %1" :: "v"(h1 * 8u), "v"(pb) : "memory");
			asm volatile("ds_write_b32 %0, %1" :: "v"(zbase + h1 * 4u), "v"(wb.z) : "memory");
		}
		else
		{
			asm volatile("ds_write_b96 %0, %1" :: "v"(h0 * (MODE ? 12u : 16u)), "v"(wa) : "memory");
			asm volatile("ds_write_b96 %0, %1" :: "v"(h1 * (MODE ? 12u : 16u)), "v"(wb) : "memory");
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__syncthreads();
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) cycles[MODE] = t1 - t0;
	out[threadIdx.x + MODE * blockDim.x] = acc + lds[(threadIdx.x * 7u) % (n_local * stride)];
}
int main()
{
	const unsigned n = 7001, m = 100000;
	std::vector<float4> src(m);
	std::vector<unsigned> gid(n);
	for (unsigned i = 0; i < m; i++) src[i] = make_float4((float)i, i + 0.25f, i + 0.5f, i + 0.75f);
	for (unsigned i = 0; i < n; i++) gid[i] = (i * 7919u + 13u) % m;
	float4 *dsrc; f3 *ddst; unsigned *dgid;
	hipMalloc(&dsrc, m * 16); hipMalloc(&ddst, n * 12); hipMalloc(&dgid, n * 4);
	hipMemcpy(dsrc, src.data(), m * 16, hipMemcpyHostToDevice);
	hipMemcpy(dgid, gid.data(), n * 4, hipMemcpyHostToDevice);
	hipMemset(ddst, 0xff, n * 12);
	hipLaunchKernelGGL(k_dma, dim3(1), dim3(1024), 8192 * 12, 0, dsrc, dgid, ddst, n);
	std::vector<f3> out(n);
	hipMemcpy(out.data(), ddst, n * 12, hipMemcpyDeviceToHost);
	unsigned bad = 0;
	for (unsigned i = 0; i < n; i++)
	{
		const float4 e = src[gid[i]];
		if (out[i].x != e.x || out[i].y != e.y || out[i].z != e.z) { if (bad < 5) printf("mismatch at %u: got %g %g %g expected %g %g %g\n", i, out[i].x, out[i].y, out[i].z, e.x, e.y, e.z); bad++; }
	}
	printf("global_load_lds_dwordx3 -> 12-byte slots: %u mismatches of %u (%s)\n", bad, n, hipGetErrorString(hipGetLastError()));
	// gather / scatter cost
	const unsigned n_local = 7000, threads = 1024, iters = 2000;
	std::vector<unsigned> idx(2 * threads);
	for (unsigned i = 0; i < threads; i++) { idx[i] = (i * 2654435761u) % n_local; idx[i + threads] = (idx[i] + 1 + (i % 5)) % n_local; }   // (disjoint per lane pair is not needed for timing)
	unsigned *didx; float *dout; unsigned long long *dcyc;
	hipMalloc(&didx, idx.size() * 4); hipMalloc(&dout, 3 * threads * 4); hipMalloc(&dcyc, 24);
	hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
	for (int rep = 0; rep < 2; rep++)
	{
		hipLaunchKernelGGL(k_gs<0>, dim3(1), dim3(threads), n_local * 16, 0, didx, dout, n_local, iters, dcyc);
		hipLaunchKernelGGL(k_gs<1>, dim3(1), dim3(threads), n_local * 12, 0, didx, dout, n_local, iters, dcyc);
		hipLaunchKernelGGL(k_gs<2>, dim3(1), dim3(threads), n_local * 12, 0, didx, dout, n_local, iters, dcyc);
		hipDeviceSynchronize();
	}
	unsigned long long cyc[3];
	hipMemcpy(cyc, dcyc, 24, hipMemcpyDeviceToHost);
	printf("gather 2 + scatter 2 per lane, 1024 threads, %u rounds: 16-byte slots %.1f cycles per round, 12-byte slots (b96) %.1f, split planes (b64 + b32) %.1f (%s)\n", iters, (double)cyc[0] / iters, (double)cyc[1] / iters, (double)cyc[2] / iters, hipGetErrorString(hipGetLastError()));
	return bad != 0;
}
