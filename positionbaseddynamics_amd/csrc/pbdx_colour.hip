// pbdx_colour.hip -- the reference's constraint colouring on the device, EXACT.
//
// SimulationModel::initConstraintGroups (SimulationModel.cpp:1033-1094) walks the constraints in creation order and puts constraint i into
// the first group none of whose members shares a body with it:
//        group(i) = mex { group(j) : j < i, j shares a body with i }.
// The engine must consume that colouring verbatim (a different one changes the Gauss-Seidel order and hence the result), so a device form
// has to reproduce it bit for bit.  It does, because the recurrence only orders constraints that share a body:
//   * the constraints on one body are coloured in index order (k > j on the same body needs group(j)), so every body has ONE next
//     uncoloured constraint at any time, and a constraint is READY once it is the next one of all its bodies;
//   * two ready constraints never share a body, so all ready constraints are coloured at once from the per-body sets of used groups
//     (a bit mask per body), without any race;
//   * colouring a constraint advances its bodies to their next constraints; a constraint becomes ready when the last of its bodies
//     arrives (one atomic counter per constraint: exactly one arrival sees it reach zero).
// The number of rounds is the longest chain of index-ordered neighbours -- 12.45 N for an N x N cloth (12 450 for configs[1]), 2 220 for
// the 100 k-tet bar -- with a few hundred ready constraints each: the work is a latency chain, not a throughput problem, so ONE workgroup
// runs all rounds (a workgroup barrier per round, no grid-wide synchronisation), after full-width kernels have built the per-body
// successor links with a stable radix sort of the (body, constraint) pairs.
//
// Results are compared group for group with the host colouring (pbdx_model.cpp, itself integer-exact against the reference:
// tests/test_model_vs_reference.py) in tests/test_colouring.py.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pbdx_internal.h"
#include "pbdx_device.h"

namespace pbdx {
namespace {

constexpr uint32_t kNone = 0xffffffffu;
constexpr uint32_t kWords = 2;                 // 64-bit words of a body's used-group mask: up to 128 groups on the device
constexpr uint32_t kPropagateThreads = 1024;
enum { kStError = 0, kStRounds = 1, kStFront0 = 2, kStColoured = 3, kStWords = 4 };
enum { kErrDuplicateBody = 1, kErrBodyRange = 2, kErrTooManyGroups = 3, kErrStalled = 4 };

#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return PBDX_ERR_HIP; } } while (0)

// (body, constraint) pairs, four per constraint (unused ones sort to the end); bodies4 = the constraint's bodies padded with kNone
__global__ void pairs_kernel(uint32_t nc, uint32_t num_bodies, const uint32_t *body_off, const uint32_t *bodies, uint4 *bodies4,
	uint32_t *keys, uint32_t *vals, uint32_t *pending, uint32_t *status)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= nc) return;
	const uint32_t b0 = body_off[c], nb = body_off[c + 1] - b0;
	uint32_t b[4] = { kNone, kNone, kNone, kNone };
	for (uint32_t k = 0; k < nb && k < 4u; k++) b[k] = bodies[b0 + k];
	for (uint32_t k = 0; k < nb && k < 4u; k++)
	{
		if (b[k] >= num_bodies) atomicMax(status + kStError, (uint32_t)kErrBodyRange);
		for (uint32_t j = 0; j < k; j++) if (b[j] == b[k]) atomicMax(status + kStError, (uint32_t)kErrDuplicateBody);
	}
	bodies4[c] = make_uint4(b[0], b[1], b[2], b[3]);
	for (uint32_t k = 0; k < 4u; k++) { keys[4u * c + k] = b[k]; vals[4u * c + k] = 4u * c + k; }
	pending[c] = nb;
}

// sorted by body (stable: ascending constraint inside a body): the successor of every pair on its body; the first pair of a body arrives
__global__ void links_kernel(uint32_t npairs, const uint32_t *skeys, const uint32_t *svals, uint32_t *succ, uint32_t *pending, uint32_t *front,
	uint32_t *status)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= npairs) return;
	const uint32_t key = skeys[s];
	if (key == kNone) { succ[svals[s]] = kNone; return; }
	const uint32_t q = svals[s];
	succ[q] = (s + 1u < npairs && skeys[s + 1u] == key) ? (svals[s + 1u] >> 2) : kNone;
	if (s == 0u || skeys[s - 1u] != key)
	{
		const uint32_t c = q >> 2;
		if (atomicSub(pending + c, 1u) == 1u) front[atomicAdd(status + kStFront0, 1u)] = c;
	}
}

__device__ __forceinline__ uint32_t ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all rounds, one workgroup.  State that one round writes and the next reads goes through agent-scope accesses (past the CU's L1).
__global__ __launch_bounds__(kPropagateThreads) void propagate_kernel(uint32_t nc, const uint4 *bodies4, const uint4 *succ4, uint32_t *pending,
	uint64_t *used, uint32_t *colour, uint32_t *front_a, uint32_t *front_b, uint32_t *status)
{
	__shared__ uint32_t s_next, s_error;
	uint32_t *cur = front_a, *nxt = front_b;
	uint32_t n = status[kStFront0], rounds = 0, coloured = 0;
	if (threadIdx.x == 0) { s_next = 0u; s_error = 0u; }
	__syncthreads();
	while (n)
	{
		for (uint32_t t = threadIdx.x; t < n; t += kPropagateThreads)
		{
			const uint32_t c = ld32(cur + t);
			const uint4 b = bodies4[c], sc = succ4[c];
			const uint32_t body[4] = { b.x, b.y, b.z, b.w }, next[4] = { sc.x, sc.y, sc.z, sc.w };
			uint64_t occ[kWords];
			uint64_t mine[4][kWords];
			for (uint32_t w = 0; w < kWords; w++) occ[w] = 0;
			for (uint32_t k = 0; k < 4u; k++)
				if (body[k] != kNone)
					for (uint32_t w = 0; w < kWords; w++) { mine[k][w] = ld64(used + (size_t)body[k] * kWords + w); occ[w] |= mine[k][w]; }
			// the first group none of the bodies is in yet
			uint32_t g = kNone;
			for (uint32_t w = 0; w < kWords && g == kNone; w++) if (~occ[w]) g = w * 64u + (uint32_t)__builtin_ctzll(~occ[w]);
			if (g == kNone) { s_error = kErrTooManyGroups; continue; }
			colour[c] = g;
			for (uint32_t k = 0; k < 4u; k++)
				if (body[k] != kNone)
				{
					st64(used + (size_t)body[k] * kWords + (g >> 6), mine[k][g >> 6] | (1ull << (g & 63u)));
					// the body moves on to its next constraint; the last body to arrive there makes it ready
					if (next[k] != kNone && atomicSub(pending + next[k], 1u) == 1u) st32(nxt + atomicAdd(&s_next, 1u), next[k]);
				}
		}
		coloured += n;
		rounds++;
		__threadfence();
		__syncthreads();
		n = s_next;
		uint32_t *tmp = cur; cur = nxt; nxt = tmp;
		__syncthreads();
		if (threadIdx.x == 0) s_next = 0u;
		if (s_error) break;
		__syncthreads();
	}
	if (threadIdx.x == 0)
	{
		status[kStRounds] = rounds;
		status[kStColoured] = coloured;
		if (s_error) atomicMax(status + kStError, s_error);
		else if (coloured != nc) atomicMax(status + kStError, (uint32_t)kErrStalled);
	}
}

struct DeviceBuffers
{
	std::vector<void *> ptrs;
	~DeviceBuffers() { for (void *p : ptrs) if (p) (void)hipFree(p); }
	template <class T> hipError_t alloc(T **out, size_t count)
	{
		void *p = nullptr;
		const hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
		if (e == hipSuccess) ptrs.push_back(p);
		*out = static_cast<T *>(p);
		return e;
	}
};

} // namespace
} // namespace pbdx

using namespace pbdx;

extern "C" int pbdx_colour_constraints(int device, uint32_t num_bodies, uint32_t num_constraints, const uint32_t *body_off, const uint32_t *bodies,
	uint32_t *group_of, uint32_t *num_groups, uint32_t *rounds)
{
	if ((!body_off || !bodies || !group_of) && num_constraints) { set_error("pbdx_colour_constraints: null argument"); return PBDX_ERR_INVALID; }
	if (num_groups) *num_groups = 0;
	if (rounds) *rounds = 0;
	if (num_constraints == 0) return PBDX_OK;
	if (num_constraints >= (1u << 29)) { set_error("pbdx_colour_constraints: too many constraints (the pair sort counts 4 per constraint in an int)"); return PBDX_ERR_UNSUPPORTED; }
	for (uint32_t c = 0; c < num_constraints; c++)
		if (body_off[c + 1] < body_off[c] || body_off[c + 1] - body_off[c] > 4u || body_off[c + 1] == body_off[c])
		{ set_error("pbdx_colour_constraints: constraint %u has %u bodies (1..4 supported)", c, body_off[c + 1] - body_off[c]); return PBDX_ERR_UNSUPPORTED; }
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible: the engine has no CPU fallback"); return PBDX_ERR_NO_DEVICE; }
	if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(device);
	// developer aid: PBDX_COLOUR_VERBOSE=1 prints where the time goes
	const bool verbose = getenv("PBDX_COLOUR_VERBOSE") != nullptr;
	const auto t_start = std::chrono::steady_clock::now();
	auto lap = [&](const char *what)
	{
		if (!verbose) return;
		(void)hipDeviceSynchronize();
		fprintf(stderr, "[colour] %-28s %8.3f ms since start\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
	};
	const uint32_t nc = num_constraints, npairs = 4u * nc, nidx = body_off[nc];
	DeviceBuffers buf;
	uint32_t *d_off, *d_bodies, *d_keys, *d_vals, *d_skeys, *d_svals, *d_succ, *d_pending, *d_colour, *d_front_a, *d_front_b, *d_status;
	uint4 *d_bodies4;
	uint64_t *d_used;
	HIPCHECK(buf.alloc(&d_off, (size_t)nc + 1)); HIPCHECK(buf.alloc(&d_bodies, nidx)); HIPCHECK(buf.alloc(&d_bodies4, nc));
	HIPCHECK(buf.alloc(&d_keys, npairs)); HIPCHECK(buf.alloc(&d_vals, npairs)); HIPCHECK(buf.alloc(&d_skeys, npairs)); HIPCHECK(buf.alloc(&d_svals, npairs));
	HIPCHECK(buf.alloc(&d_succ, npairs)); HIPCHECK(buf.alloc(&d_pending, nc)); HIPCHECK(buf.alloc(&d_colour, nc));
	HIPCHECK(buf.alloc(&d_front_a, nc)); HIPCHECK(buf.alloc(&d_front_b, nc)); HIPCHECK(buf.alloc(&d_status, (size_t)kStWords));
	HIPCHECK(buf.alloc(&d_used, (size_t)num_bodies * kWords));
	HIPCHECK(pbdx::copy_to_device(d_off, body_off, ((size_t)nc + 1) * sizeof(uint32_t)));
	HIPCHECK(pbdx::copy_to_device(d_bodies, bodies, (size_t)nidx * sizeof(uint32_t)));
	HIPCHECK(hipMemset(d_status, 0, kStWords * sizeof(uint32_t)));
	HIPCHECK(hipMemset(d_used, 0, (size_t)num_bodies * kWords * sizeof(uint64_t)));
	lap("allocated, uploaded");
	const uint32_t tb = 256;
	hipLaunchKernelGGL(pairs_kernel, dim3((nc + tb - 1) / tb), dim3(tb), 0, 0, nc, num_bodies, d_off, d_bodies, d_bodies4, d_keys, d_vals, d_pending, d_status);
	HIPCHECK(hipGetLastError());
	// stable sort of the pairs by body: inside a body the pairs stay in constraint order
	{
		size_t temp_bytes = 0;
		HIPCHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_skeys, d_vals, d_svals, (int)npairs, 0, 32, (hipStream_t)0));
		unsigned char *d_temp;
		HIPCHECK(buf.alloc(&d_temp, temp_bytes));
		HIPCHECK(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_skeys, d_vals, d_svals, (int)npairs, 0, 32, (hipStream_t)0));
	}
	lap("pairs sorted by body");
	hipLaunchKernelGGL(links_kernel, dim3((npairs + tb - 1) / tb), dim3(tb), 0, 0, npairs, d_skeys, d_svals, d_succ, d_pending, d_front_a, d_status);
	HIPCHECK(hipGetLastError());
	lap("successor links");
	hipLaunchKernelGGL(propagate_kernel, dim3(1), dim3(kPropagateThreads), 0, 0, nc, d_bodies4, reinterpret_cast<const uint4 *>(d_succ), d_pending, d_used, d_colour,
		d_front_a, d_front_b, d_status);
	HIPCHECK(hipGetLastError());
	uint32_t status[kStWords];
	HIPCHECK(hipMemcpy(status, d_status, sizeof(status), hipMemcpyDeviceToHost));
	lap("propagated");
	if (verbose) fprintf(stderr, "[colour] %u constraints, %u rounds, first frontier %u\n", nc, status[kStRounds], status[kStFront0]);
	if (rounds) *rounds = status[kStRounds];
	if (status[kStError])
	{
		static const char *what[] = { "", "a constraint names the same body twice", "a body index is out of range", "more than 128 groups",
			"the propagation stalled before every constraint was coloured" };
		set_error("pbdx_colour_constraints: %s (colour on the host instead)", what[std::min<uint32_t>(status[kStError], 4u)]);
		return status[kStError] == kErrBodyRange ? PBDX_ERR_INVALID : PBDX_ERR_UNSUPPORTED;
	}
	HIPCHECK(pbdx::copy_from_device(group_of, d_colour, (size_t)nc * sizeof(uint32_t)));
	if (num_groups)
	{
		uint32_t g = 0;
		for (uint32_t c = 0; c < nc; c++) g = std::max(g, group_of[c] + 1u);
		*num_groups = g;
	}
	return PBDX_OK;
}

// Stable radix sort of (key, value) pairs of 32-bit words on a stream (hipcub lives in this translation unit only).  temp == nullptr: size query.
namespace pbdx {
int sort_pairs_u32(void *temp, size_t *temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, void *stream)
{
	const hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 32, (hipStream_t)stream);
	if (e != hipSuccess) { set_error("radix sort of %u pairs failed: %s", n, hipGetErrorString(e)); return PBDX_ERR_HIP; }
	return PBDX_OK;
}
}

