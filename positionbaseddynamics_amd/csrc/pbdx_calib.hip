// pbdx_calib.hip -- calibration kernels and their developer entry points (include/pbdx_debug.h): what this GPU delivers for plain streaming reads, writes and
// copies and how fast one SIMD issues dependent VALU instructions -- the ceilings bench.py prices the sweep kernels against (roofline.copy_ceiling,
// valu_issue).  Nothing here is on the product path.
#include <hip/hip_runtime.h>
#include "pbdx_internal.h"
#include "pbdx_device.h"
#include "../../include/pbdx_debug.h"
#include <chrono>
#include <algorithm>
#include <vector>
#include <string.h>

using namespace pbdx;

#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return PBDX_ERR_HIP; } } while (0)

namespace {

// ---- counter calibration kernels (pbdx_debug_stream): known byte counts in this engine's own access
// widths, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be turned into bytes (MI355X_MICROARCH.md, HBM)
__global__ __launch_bounds__(256) void calib_read_b32(const float *__restrict__ src, float *__restrict__ sink, size_t n)
{
	float acc = 0.0f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
	if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read_b128(const float4 *__restrict__ src, float *__restrict__ sink, size_t n)
{
	float acc = 0.0f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.w; }
	if (acc == 123.456f) sink[0] = acc;
}
// the same buffer read `passes` times in one launch: larger than the L2s, smaller than the 256 MiB Infinity Cache -- tells
// whether a memory-side counter sees Infinity-Cache hits (scripts/mall_counters.sh)
__global__ __launch_bounds__(256) void calib_reread_b128(const float4 *__restrict__ src, float *__restrict__ sink, size_t n, int passes)
{
	float acc = 0.0f;
	for (int p = 0; p < passes; p++)
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		{
			const float4 v = src[(i + (size_t)p * 977) % n];
			acc += v.x + v.w;
		}
	if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_write_b32(float *__restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 1.0f;
}
__global__ __launch_bounds__(256) void calib_write_b128(float4 *__restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}
// the practical HBM roof next to the 8 TB/s of the data sheet ("use the measured copy bandwidth as the practical roof and report both"):
// a float4 copy (pbdx_debug_copy_bandwidth)
__global__ __launch_bounds__(256) void calib_copy_b128(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
	// four independent 16-byte loads in flight per lane and round
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i + 3 * stride < n; i += 4 * stride)
	{
		const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
		dst[i] = v0; dst[i + stride] = v1; dst[i + 2 * stride] = v2; dst[i + 3 * stride] = v3;
	}
	for (; i < n; i += stride) dst[i] = src[i];
}
// the vector-ALU issue interval of a SIMD at the occupancy of the sweep kernels (pbdx_debug_valu_issue): every wave of a BLOCK-thread workgroup (one
// per CU: BLOCK / 256 waves per SIMD) runs `iters` x 64 independent-enough v_mul_f32 / v_add_f32 (eight chains) between two reads of the shader
// clock; cycles / (instructions of a wave x waves per SIMD) = cycles per wave64 instruction and SIMD -- what SQ_INSTS_VALU is to be priced with
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void calib_valu_kernel(float *sink, unsigned long long *cycles, int iters, float seed)
{
	float a[8];
#pragma unroll
	for (int i = 0; i < 8; i++) a[i] = seed + (float)i + (float)threadIdx.x;
	const float c = seed * 0.999f;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int i = 0; i < 8; i++) a[i] = (i & 1) ? a[i] + c : a[i] * c;
	}
	asm volatile("" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
	const unsigned long long t1 = __builtin_readcyclecounter();
	if ((threadIdx.x & 63u) == 0u) cycles[blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)] = t1 - t0;
	float sum = 0.0f;
#pragma unroll
	for (int i = 0; i < 8; i++) sum += a[i];
	if (sum == 12345.678f) sink[0] = sum;
}

} // namespace

extern "C" {


int pbdx_debug_stream(int device, uint64_t nbytes, int mode)
{
	ENTER_DEVICE(device);
	if (nbytes < 4096) { set_error("debug_stream: nbytes too small"); return PBDX_ERR_INVALID; }
	float *buf = nullptr, *sink = nullptr;
	HIPCHECK(hipMalloc(&buf, nbytes));
	HIPCHECK(hipMalloc(&sink, 64));
	HIPCHECK(hipMemset(buf, 0, nbytes));
	HIPCHECK(hipDeviceSynchronize());
	const dim3 grid(256 * 32), block(256);
	switch (mode)
	{
	case 0: hipLaunchKernelGGL(calib_read_b32, grid, block, 0, 0, buf, sink, (size_t)(nbytes / 4)); break;
	case 1: hipLaunchKernelGGL(calib_read_b128, grid, block, 0, 0, reinterpret_cast<const float4 *>(buf), sink, (size_t)(nbytes / 16)); break;
	case 2: hipLaunchKernelGGL(calib_write_b32, grid, block, 0, 0, buf, (size_t)(nbytes / 4)); break;
	case 3: hipLaunchKernelGGL(calib_write_b128, grid, block, 0, 0, reinterpret_cast<float4 *>(buf), (size_t)(nbytes / 16)); break;
	case 4: hipLaunchKernelGGL(calib_reread_b128, grid, block, 0, 0, reinterpret_cast<const float4 *>(buf), sink, (size_t)(nbytes / 16), 8); break;
	default: (void)hipFree(buf); (void)hipFree(sink); set_error("debug_stream: mode 0..4"); return PBDX_ERR_INVALID;
	}
	HIPCHECK(hipGetLastError());
	HIPCHECK(hipDeviceSynchronize());
	(void)hipFree(buf); (void)hipFree(sink);
	return PBDX_OK;
}

// Measured device-to-device float4 copy bandwidth (GB/s, read + written bytes) of `nbytes` per direction: the practical HBM roof.
int pbdx_debug_copy_bandwidth(int device, uint64_t nbytes, int reps, double *gbs)
{
	ENTER_DEVICE(device);
	if (!gbs || nbytes < 4096 || reps < 1) { set_error("debug_copy_bandwidth: invalid arguments"); return PBDX_ERR_INVALID; }
	float4 *a = nullptr, *b = nullptr;
	HIPCHECK(hipMalloc(&a, nbytes));
	if (hipMalloc(&b, nbytes) != hipSuccess) { (void)hipFree(a); set_error("debug_copy_bandwidth: allocation failed"); return PBDX_ERR_HIP; }
	HIPCHECK(hipMemset(a, 0, nbytes));
	hipEvent_t e0, e1;
	HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
	const dim3 grid(256 * 32), block(256);
	float best_ms = 1e30f;
	// the engine's own float4 copy kernel and the runtime's device-to-device copy, in turns; the faster of the two is the roof
	for (int r = 0; r < 2 * (reps + 1); r++)      // (the first repetition of each warms up)
	{
		HIPCHECK(hipEventRecord(e0, 0));
		if (r & 1) HIPCHECK(hipMemcpyAsync(b, a, nbytes, hipMemcpyDeviceToDevice, 0));
		else hipLaunchKernelGGL(calib_copy_b128, grid, block, 0, 0, a, b, (size_t)(nbytes / 16));
		HIPCHECK(hipEventRecord(e1, 0));
		HIPCHECK(hipEventSynchronize(e1));
		float ms = 0.0f;
		HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
		if (r >= 2 && ms < best_ms) best_ms = ms;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	(void)hipFree(a); (void)hipFree(b);
	*gbs = 2.0 * (double)nbytes / ((double)best_ms * 1e-3) / 1e9;
	return PBDX_OK;
}

// Measured vector-ALU issue interval: shader cycles per wave64 v_mul_f32 / v_add_f32 and SIMD with `threads` (256 / 512 / 1024) threads per workgroup,
// one workgroup per CU (threads / 256 waves per SIMD) -- median over all waves of the launch.
int pbdx_debug_valu_issue(int device, int threads, double *cycles_per_instruction)
{
	ENTER_DEVICE(device);
	if (!cycles_per_instruction || (threads != 256 && threads != 512 && threads != 1024)) { set_error("debug_valu_issue: threads must be 256, 512 or 1024"); return PBDX_ERR_INVALID; }
	hipDeviceProp_t prop;
	HIPCHECK(hipGetDeviceProperties(&prop, device));
	const int cus = std::max(1, prop.multiProcessorCount), waves = threads / 64, iters = 400;
	float *sink = nullptr; unsigned long long *d_cyc = nullptr;
	HIPCHECK(hipMalloc(&sink, 64));
	HIPCHECK(hipMalloc(&d_cyc, (size_t)cus * waves * sizeof(unsigned long long)));
	for (int r = 0; r < 2; r++)             // (the first launch warms up)
	{
		if (threads == 256) hipLaunchKernelGGL(calib_valu_kernel<256>, dim3(cus), dim3(256), 0, 0, sink, d_cyc, iters, 1.0f);
		else if (threads == 512) hipLaunchKernelGGL(calib_valu_kernel<512>, dim3(cus), dim3(512), 0, 0, sink, d_cyc, iters, 1.0f);
		else hipLaunchKernelGGL(calib_valu_kernel<1024>, dim3(cus), dim3(1024), 0, 0, sink, d_cyc, iters, 1.0f);
		HIPCHECK(hipGetLastError());
		HIPCHECK(hipDeviceSynchronize());
	}
	std::vector<unsigned long long> cyc((size_t)cus * waves);
	HIPCHECK(pbdx::copy_from_device(cyc.data(), d_cyc, cyc.size() * sizeof(unsigned long long)));
	(void)hipFree(sink); (void)hipFree(d_cyc);
	std::sort(cyc.begin(), cyc.end());
	const double per_wave = (double)cyc[cyc.size() / 2] / ((double)iters * 64.0);      // cycles per instruction of ONE wave
	*cycles_per_instruction = per_wave / (double)(threads / 256);                          // ... of the SIMD, which interleaves threads / 256 waves
	return PBDX_OK;
}

} // extern "C"
