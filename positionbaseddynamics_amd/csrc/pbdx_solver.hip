// pbdx_solver.hip -- the device engine: particle state in HBM, colour-batched
// constraint schedule, HIP kernels for gfx950 and the substep loop.
//
// Replaces the inner loop of PBD::TimeStepController::step
// (Simulation/TimeStepController.cpp:75-176) and positionConstraintProjection
// (:251-295) of the reference for particle scenes.
//
// HBM layout
//   pos [n] float4  (x, y, z, invMass)   gathered/scattered by every projection: ONE 16-byte
//                                        access per endpoint instead of 3+1 scattered dwords
//   vel [n] float4  (vx, vy, vz, mass)   streamed by integrate / velocity update only
//   old [n] float4, last [n] float4      oldX / lastX (w unused)
//   per (colour group, constraint type) batch:
//     idx    [count] uint2 / uint4       particle indices of one constraint, one vector load per lane
//     params [planes][count] float       planar (SoA) parameter streams -> lane-contiguous loads;
//                                        parameters that are uniform over the batch (stiffness,
//                                        material constants) are folded into kernel arguments
//     lambda [count] float               XPBD multipliers (own stream; not read at iteration 0)
// One launch = one (group, type) batch, one constraint per lane (wave64, 256-thread workgroups);
// groups are launched in colour order on one stream (stream order = Gauss-Seidel order); one
// substep (integrate, iterations x batches, velocity update) is captured into a hipGraph.
// All kernels are HBM/L2-bound gather-scatter streams: no LDS reuse exists inside one colour
// (every particle is touched by at most one constraint per colour), no MFMA.
#include <hip/hip_runtime.h>
#include "pbdx_internal.h"
#include "pbdx_project.h"
#include <algorithm>
#include <string.h>

using namespace pbdx;

#define PBDX_MAX_PARAMS 24

namespace {

struct ParamView
{
	const float *base;        // planar parameter streams
	uint32_t stride;          // floats per plane (= count rounded up to 4)
	uint32_t umask;           // bit k set: parameter k is uniform over the batch -> u[k]
	float u[PBDX_MAX_PARAMS];
	uint8_t slot[PBDX_MAX_PARAMS];   // plane index of a non-uniform parameter
};

struct BatchArgs
{
	float4 *pos;
	const uint32_t *idx;
	float *lambda;
	ParamView pv;
	uint32_t count;
	float dt;                 // substep size (XPBD compliance)
	int first_iter;           // iteration 0 of a substep: lambda := 0 without reading it
	uint32_t num_blocks;      // grid size (for the XCD-aware remap)
	int xcd_remap;
};

__device__ __forceinline__ float pget(const ParamView &pv, int k, uint32_t i)
{
	return ((pv.umask >> k) & 1u) ? pv.u[k] : pv.base[(size_t)pv.slot[k] * pv.stride + i];
}

__device__ __forceinline__ void ldp(const float4 *pos, uint32_t i, V3 &p, float &w)
{
	const float4 v = pos[i];
	p = mk(v.x, v.y, v.z); w = v.w;
}
__device__ __forceinline__ void apply(float4 *pos, uint32_t i, V3 p, V3 c, float w)
{
	if (w != 0.0f)
		pos[i] = make_float4(p.x + c.x, p.y + c.y, p.z + c.z, w);
}

struct QFull
{
	float q[16];   // column-major Q(j,k) = q[k*4+j]
	__device__ __forceinline__ float operator()(int j, int k) const { return q[k * 4 + j]; }
};

template <int TYPE> __device__ __forceinline__ void project_one(const BatchArgs &a, uint32_t i);

template <> __device__ __forceinline__ void project_one<PBDX_DISTANCE>(const BatchArgs &a, uint32_t i)
{
	const uint2 id = reinterpret_cast<const uint2 *>(a.idx)[i];
	V3 p0, p1; float w0, w1;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1);
	V3 c0, c1;
	if (solve_distance(p0, w0, p1, w1, pget(a.pv, 0, i), pget(a.pv, 1, i), c0, c1))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_DISTANCE_XPBD>(const BatchArgs &a, uint32_t i)
{
	const uint2 id = reinterpret_cast<const uint2 *>(a.idx)[i];
	V3 p0, p1; float w0, w1;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1);
	float lambda = a.first_iter ? 0.0f : a.lambda[i];
	V3 c0, c1;
	if (solve_distance_xpbd(p0, w0, p1, w1, pget(a.pv, 0, i), pget(a.pv, 1, i), a.dt, lambda, c0, c1))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1);
	}
	a.lambda[i] = lambda;
}

template <> __device__ __forceinline__ void project_one<PBDX_DIHEDRAL>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	V3 c0, c1, c2, c3;
	if (solve_dihedral(p0, w0, p1, w1, p2, w2, p3, w3, pget(a.pv, 0, i), pget(a.pv, 1, i), c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_ISOMETRIC_BENDING>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	QFull q;
#pragma unroll
	for (int k = 0; k < 16; k++) q.q[k] = pget(a.pv, 1 + k, i);
	V3 c0, c1, c2, c3;
	if (solve_isometric_bending(p0, w0, p1, w1, p2, w2, p3, w3, q, pget(a.pv, 0, i), c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_ISOMETRIC_BENDING_XPBD>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	QFull q;
#pragma unroll
	for (int k = 0; k < 16; k++) q.q[k] = pget(a.pv, 1 + k, i);
	float lambda = a.first_iter ? 0.0f : a.lambda[i];
	V3 c0, c1, c2, c3;
	if (solve_isometric_bending_xpbd(p0, w0, p1, w1, p2, w2, p3, w3, q, pget(a.pv, 0, i), a.dt, lambda, c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
	a.lambda[i] = lambda;
}

template <> __device__ __forceinline__ void project_one<PBDX_FEM_TRIANGLE>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2; float w0, w1, w2;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2);
	float im[2][2];
	im[0][0] = pget(a.pv, 1, i); im[1][0] = pget(a.pv, 2, i); im[0][1] = pget(a.pv, 3, i); im[1][1] = pget(a.pv, 4, i);
	V3 c0, c1, c2;
	if (solve_fem_triangle(p0, w0, p1, w1, p2, w2, pget(a.pv, 0, i), im, pget(a.pv, 5, i), pget(a.pv, 6, i), pget(a.pv, 7, i),
		pget(a.pv, 8, i), pget(a.pv, 9, i), c0, c1, c2))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_STRAIN_TRIANGLE>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2; float w0, w1, w2;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2);
	float im[2][2];
	im[0][0] = pget(a.pv, 0, i); im[1][0] = pget(a.pv, 1, i); im[0][1] = pget(a.pv, 2, i); im[1][1] = pget(a.pv, 3, i);
	V3 c0, c1, c2;
	if (solve_strain_triangle(p0, w0, p1, w1, p2, w2, im, pget(a.pv, 4, i), pget(a.pv, 5, i), pget(a.pv, 6, i),
		pget(a.pv, 7, i) != 0.0f, pget(a.pv, 8, i) != 0.0f, c0, c1, c2))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_VOLUME>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	V3 c0, c1, c2, c3;
	if (solve_volume(p0, w0, p1, w1, p2, w2, p3, w3, pget(a.pv, 0, i), pget(a.pv, 1, i), c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_VOLUME_XPBD>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	float lambda = a.first_iter ? 0.0f : a.lambda[i];
	V3 c0, c1, c2, c3;
	if (solve_volume_xpbd(p0, w0, p1, w1, p2, w2, p3, w3, pget(a.pv, 0, i), pget(a.pv, 1, i), a.dt, lambda, c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
	a.lambda[i] = lambda;
}

__device__ __forceinline__ M3 load_m3(const ParamView &pv, int first, uint32_t i)
{
	M3 A;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int r = 0; r < 3; r++) A.m[r][c] = pget(pv, first + c * 3 + r, i);
	return A;
}

template <> __device__ __forceinline__ void project_one<PBDX_FEM_TET>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	const float vol = pget(a.pv, 0, i);
	const M3 im = load_m3(a.pv, 1, i);
	const bool hi = fem_tet_handle_inversion(p0, p1, p2, p3, vol);
	V3 c0, c1, c2, c3;
	if (solve_fem_tet(p0, w0, p1, w1, p2, w2, p3, w3, vol, im, pget(a.pv, 10, i), pget(a.pv, 11, i), hi, c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_FEM_TET_XPBD>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	const float vol = pget(a.pv, 0, i);
	const M3 im = load_m3(a.pv, 1, i);
	const bool hi = fem_tet_handle_inversion(p0, p1, p2, p3, vol);
	float lambda = a.first_iter ? 0.0f : a.lambda[i];
	V3 c0, c1, c2, c3;
	if (solve_fem_tet_xpbd(p0, w0, p1, w1, p2, w2, p3, w3, vol, im, pget(a.pv, 10, i), pget(a.pv, 11, i), hi, a.dt, lambda, c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
	a.lambda[i] = lambda;
}

template <> __device__ __forceinline__ void project_one<PBDX_STRAIN_TET>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a.pos, id.x, p0, w0); ldp(a.pos, id.y, p1, w1); ldp(a.pos, id.z, p2, w2); ldp(a.pos, id.w, p3, w3);
	const M3 im = load_m3(a.pv, 0, i);
	V3 c0, c1, c2, c3;
	if (solve_strain_tet(p0, w0, p1, w1, p2, w2, p3, w3, im, pget(a.pv, 9, i), pget(a.pv, 10, i),
		pget(a.pv, 11, i) != 0.0f, pget(a.pv, 12, i) != 0.0f, c0, c1, c2, c3))
	{
		apply(a.pos, id.x, p0, c0, w0); apply(a.pos, id.y, p1, c1, w1); apply(a.pos, id.z, p2, c2, w2); apply(a.pos, id.w, p3, c3, w3);
	}
}

template <> __device__ __forceinline__ void project_one<PBDX_SHAPE_MATCHING>(const BatchArgs &a, uint32_t i)
{
	const uint4 id = reinterpret_cast<const uint4 *>(a.idx)[i];
	const uint32_t ids[4] = { id.x, id.y, id.z, id.w };
	V3 x[4], x0[4], corr[4]; float wl[4], w[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		ldp(a.pos, ids[k], x[k], wl[k]);
		x0[k] = mk(pget(a.pv, 4 + 3 * k, i), pget(a.pv, 5 + 3 * k, i), pget(a.pv, 6 + 3 * k, i));
		w[k] = pget(a.pv, 16 + k, i);
	}
	const V3 restCm = mk(pget(a.pv, 1, i), pget(a.pv, 2, i), pget(a.pv, 3, i));
	if (solve_shape_matching4(x0, x, w, restCm, pget(a.pv, 0, i), corr))
	{
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			// (1.0 / m_numClusters[i]) * m_corr[i]   Constraints.cpp:2024 (double quotient narrowed to Real)
			const float f = (float)(1.0 / (double)(unsigned int)pget(a.pv, 20 + k, i));
			if (w[k] != 0.0f)
				a.pos[ids[k]] = make_float4(x[k].x + f * corr[k].x, x[k].y + f * corr[k].y, x[k].z + f * corr[k].z, wl[k]);
		}
	}
}

// blockIdx -> logical block: with xcd_remap the 8 XCDs (hardware places block b on XCD b%8) each
// walk one contiguous eighth of the batch, so the particles a chiplet touches stay the same
// from colour to colour (per-XCD L2 locality); speed only, never correctness.
__device__ __forceinline__ uint32_t logical_block(const BatchArgs &a)
{
	const uint32_t b = blockIdx.x;
	if (!a.xcd_remap || a.num_blocks < 16)
		return b;
	const uint32_t per = a.num_blocks >> 3;          // full blocks per XCD
	const uint32_t body = per << 3;
	if (b >= body)
		return b;                                    // remainder blocks keep their place at the end
	return (b & 7u) * per + (b >> 3);
}

template <int TYPE>
__global__ __launch_bounds__(256) void project_kernel(BatchArgs a)
{
	const uint32_t i = logical_block(a) * blockDim.x + threadIdx.x;
	if (i < a.count)
		project_one<TYPE>(a, i);
}

// lastX <- oldX; oldX <- x; semi-implicit Euler for dynamic particles
// TimeStepController.cpp:112-118 + TimeIntegration.cpp:7-19 (acceleration == gravity for every
// dynamic particle, TimeStep.cpp:28-62)
__global__ __launch_bounds__(256) void integrate_kernel(float4 *__restrict__ pos, float4 *__restrict__ vel,
	float4 *__restrict__ old, float4 *__restrict__ last, uint32_t n, float h, float gx, float gy, float gz)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 o = old[i];
	float4 p = pos[i];
	last[i] = o;
	old[i] = p;
	float4 v = vel[i];
	if (v.w != 0.0f)   // mass != 0
	{
		v.x = v.x + gx * h; v.y = v.y + gy * h; v.z = v.z + gz * h;
		p.x = p.x + v.x * h; p.y = p.y + v.y * h; p.z = p.z + v.z * h;
		vel[i] = v;
		pos[i] = p;
	}
}

// TimeIntegration::velocityUpdateFirstOrder / SecondOrder  TimeIntegration.cpp:42-51, 69-79
__global__ __launch_bounds__(256) void velocity_kernel(const float4 *__restrict__ pos, float4 *__restrict__ vel,
	const float4 *__restrict__ old, const float4 *__restrict__ last, uint32_t n, float inv_h, int second_order)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float4 v = vel[i];
	if (v.w == 0.0f) return;
	const float4 p = pos[i];
	const float4 o = old[i];
	if (!second_order)
	{
		v.x = inv_h * (p.x - o.x); v.y = inv_h * (p.y - o.y); v.z = inv_h * (p.z - o.z);
	}
	else
	{
		const float4 l = last[i];
		v.x = inv_h * (1.5f * p.x - 2.0f * o.x + 0.5f * l.x);
		v.y = inv_h * (1.5f * p.y - 2.0f * o.y + 0.5f * l.y);
		v.z = inv_h * (1.5f * p.z - 2.0f * o.z + 0.5f * l.z);
	}
	vel[i] = v;
}

typedef void (*project_fn)(BatchArgs);
project_fn kProjectKernels[PBDX_NUM_CONSTRAINT_TYPES] = {
	project_kernel<PBDX_DISTANCE>, project_kernel<PBDX_DISTANCE_XPBD>, project_kernel<PBDX_DIHEDRAL>,
	project_kernel<PBDX_ISOMETRIC_BENDING>, project_kernel<PBDX_ISOMETRIC_BENDING_XPBD>,
	project_kernel<PBDX_FEM_TRIANGLE>, project_kernel<PBDX_STRAIN_TRIANGLE>,
	project_kernel<PBDX_VOLUME>, project_kernel<PBDX_VOLUME_XPBD>,
	project_kernel<PBDX_FEM_TET>, project_kernel<PBDX_FEM_TET_XPBD>, project_kernel<PBDX_STRAIN_TET>,
	project_kernel<PBDX_SHAPE_MATCHING>,
};

struct Batch
{
	int type = 0;
	uint32_t group = 0;
	uint32_t count = 0;
	uint32_t seq = 0;               // add_batch order
	uint32_t *d_idx = nullptr;
	float *d_params = nullptr;
	float *d_lambda = nullptr;
	ParamView pv;                   // base filled in after upload
	std::vector<uint32_t> h_idx;    // kept for validate_schedule
};

#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return PBDX_ERR_HIP; } } while (0)

} // namespace

struct pbdx_solver
{
	int device = 0;
	hipStream_t stream = nullptr;
	hipDeviceProp_t prop;
	uint32_t n = 0;
	float4 *d_pos = nullptr, *d_vel = nullptr, *d_old = nullptr, *d_last = nullptr;
	std::vector<Batch> batches;          // in add order
	std::vector<uint32_t> order;         // batch indices sorted by (group, seq)
	bool schedule_open = false;
	uint64_t schedule_version = 0;

	// options
	int use_graph = 1;
	int block_size = 256;
	int xcd_remap = 0;
	int profile = 0;

	// cached graph of one substep
	hipGraph_t graph = nullptr;
	hipGraphExec_t graph_exec = nullptr;
	struct GraphKey { float h; uint32_t iters; int vel; float g[3]; uint64_t sched; int block; int remap; uint32_t n; } key = {};
	bool graph_valid = false;

	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	std::vector<hipEvent_t> prof_events;
	pbdx_step_stats stats = {};
	double type_ms[PBDX_NUM_CONSTRAINT_TYPES] = {};
	uint64_t type_launches[PBDX_NUM_CONSTRAINT_TYPES] = {};
	uint64_t type_projections[PBDX_NUM_CONSTRAINT_TYPES] = {};

	void free_batches()
	{
		for (Batch &b : batches)
		{
			if (b.d_idx) (void)hipFree(b.d_idx);
			if (b.d_params) (void)hipFree(b.d_params);
			if (b.d_lambda) (void)hipFree(b.d_lambda);
		}
		batches.clear();
		order.clear();
	}
	void drop_graph()
	{
		if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
		if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
		graph_valid = false;
	}
	void free_particles()
	{
		for (float4 **p : { &d_pos, &d_vel, &d_old, &d_last })
			if (*p) { (void)hipFree(*p); *p = nullptr; }
		n = 0;
	}
};

namespace {

int launch_batch(pbdx_solver *s, const Batch &b, float dt, int first_iter)
{
	BatchArgs a;
	a.pos = s->d_pos;
	a.idx = b.d_idx;
	a.lambda = b.d_lambda;
	a.pv = b.pv;
	a.count = b.count;
	a.dt = dt;
	a.first_iter = first_iter;
	const uint32_t bs = (uint32_t)s->block_size;
	a.num_blocks = (b.count + bs - 1) / bs;
	a.xcd_remap = s->xcd_remap;
	hipLaunchKernelGGL(kProjectKernels[b.type], dim3(a.num_blocks), dim3(bs), 0, s->stream, a);
	HIPCHECK(hipGetLastError());
	return PBDX_OK;
}

// event bookkeeping for the profiled (eager) mode: one event before every projection launch and
// one after the last launch of a sweep; elapsed(e[i], e[i+1]) is charged to launch i.
struct ProfCursor { pbdx_solver *s; size_t next = 0; std::vector<int> types; std::vector<uint32_t> counts; };

int prof_mark(ProfCursor *pc)
{
	pbdx_solver *s = pc->s;
	if (pc->next >= s->prof_events.size())
	{
		hipEvent_t e;
		HIPCHECK(hipEventCreate(&e));
		s->prof_events.push_back(e);
	}
	HIPCHECK(hipEventRecord(s->prof_events[pc->next++], s->stream));
	return PBDX_OK;
}

int projection_sweeps(pbdx_solver *s, float dt, uint32_t iterations, ProfCursor *pc)
{
	for (uint32_t it = 0; it < iterations; it++)
		for (uint32_t bi : s->order)
		{
			const Batch &b = s->batches[bi];
			if (pc) { int r = prof_mark(pc); if (r) return r; pc->types.push_back(b.type); pc->counts.push_back(b.count); }
			int r = launch_batch(s, b, dt, it == 0);
			if (r) return r;
		}
	if (pc) { int r = prof_mark(pc); if (r) return r; pc->types.push_back(-1); pc->counts.push_back(0); }
	return PBDX_OK;
}

int enqueue_substep(pbdx_solver *s, float hs, float inv_h, uint32_t iters, int vel, const float g[3], ProfCursor *pc)
{
	const uint32_t bs = 256;
	const uint32_t nb = (s->n + bs - 1) / bs;
	if (s->n)
	{
		hipLaunchKernelGGL(integrate_kernel, dim3(nb), dim3(bs), 0, s->stream, s->d_pos, s->d_vel, s->d_old, s->d_last, s->n, hs, g[0], g[1], g[2]);
		HIPCHECK(hipGetLastError());
	}
	int r = projection_sweeps(s, hs, iters, pc);
	if (r) return r;
	if (s->n)
	{
		hipLaunchKernelGGL(velocity_kernel, dim3(nb), dim3(bs), 0, s->stream, s->d_pos, s->d_vel, s->d_old, s->d_last, s->n, inv_h, vel != 0);
		HIPCHECK(hipGetLastError());
	}
	return PBDX_OK;
}

int collect_profile(pbdx_solver *s, ProfCursor *pc)
{
	HIPCHECK(hipStreamSynchronize(s->stream));
	for (size_t i = 0; i + 1 < pc->next; i++)
	{
		const int t = pc->types[i];
		if (t < 0) continue;
		float ms = 0.0f;
		HIPCHECK(hipEventElapsedTime(&ms, s->prof_events[i], s->prof_events[i + 1]));
		s->type_ms[t] += ms;
		s->type_launches[t]++;
		s->type_projections[t] += pc->counts[i];
		s->stats.projection_ms += ms;
		s->stats.projection_launches++;
	}
	return PBDX_OK;
}

} // namespace

extern "C" {

int pbdx_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int pbdx_solver_create(pbdx_solver **out, int device)
{
	if (!out) { set_error("pbdx_solver_create: null out"); return PBDX_ERR_INVALID; }
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
	{
		set_error("no HIP device visible: the engine has no CPU fallback");
		return PBDX_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return PBDX_ERR_INVALID; }
	pbdx_solver *s = new (std::nothrow) pbdx_solver();
	if (!s) { set_error("out of memory"); return PBDX_ERR_ALLOC; }
	s->device = device;
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess) e = hipGetDeviceProperties(&s->prop, device);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreate(&s->ev_start);
	if (e == hipSuccess) e = hipEventCreate(&s->ev_stop);
	if (e != hipSuccess)
	{
		set_error("engine initialisation failed: %s", hipGetErrorString(e));
		delete s;
		return PBDX_ERR_HIP;
	}
	*out = s;
	return PBDX_OK;
}

void pbdx_solver_destroy(pbdx_solver *s)
{
	if (!s) return;
	(void)hipSetDevice(s->device);
	if (s->stream) (void)hipStreamSynchronize(s->stream);
	s->drop_graph();
	s->free_batches();
	s->free_particles();
	for (hipEvent_t e : s->prof_events) (void)hipEventDestroy(e);
	if (s->ev_start) (void)hipEventDestroy(s->ev_start);
	if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
	if (s->stream) (void)hipStreamDestroy(s->stream);
	delete s;
}

int pbdx_solver_set_particles(pbdx_solver *s, uint32_t n, const float *x, const float *v, const float *old_x,
	const float *last_x, const float *mass, const float *inv_mass)
{
	if (!s || !x || !mass || !inv_mass) { set_error("set_particles: x, mass and inv_mass are required"); return PBDX_ERR_INVALID; }
	HIPCHECK(hipSetDevice(s->device));
	if (n != s->n)
	{
		HIPCHECK(hipStreamSynchronize(s->stream));
		s->free_particles();
		s->drop_graph();
		if (n)
		{
			HIPCHECK(hipMalloc(&s->d_pos, (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_vel, (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_old, (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_last, (size_t)n * sizeof(float4)));
		}
		s->n = n;
	}
	if (!n) return PBDX_OK;
	std::vector<float4> tmp(n);
	for (uint32_t i = 0; i < n; i++) tmp[i] = make_float4(x[3 * i], x[3 * i + 1], x[3 * i + 2], inv_mass[i]);
	HIPCHECK(hipMemcpyAsync(s->d_pos, tmp.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
	HIPCHECK(hipStreamSynchronize(s->stream));
	for (uint32_t i = 0; i < n; i++)
		tmp[i] = v ? make_float4(v[3 * i], v[3 * i + 1], v[3 * i + 2], mass[i]) : make_float4(0.0f, 0.0f, 0.0f, mass[i]);
	HIPCHECK(hipMemcpyAsync(s->d_vel, tmp.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
	HIPCHECK(hipStreamSynchronize(s->stream));
	const float *src = old_x ? old_x : x;
	for (uint32_t i = 0; i < n; i++) tmp[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.0f);
	HIPCHECK(hipMemcpyAsync(s->d_old, tmp.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
	HIPCHECK(hipStreamSynchronize(s->stream));
	src = last_x ? last_x : x;
	for (uint32_t i = 0; i < n; i++) tmp[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.0f);
	HIPCHECK(hipMemcpyAsync(s->d_last, tmp.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_set_positions(pbdx_solver *s, uint32_t n, const float *x)
{
	if (!s || !x || n != s->n) { set_error("set_positions: particle count mismatch"); return PBDX_ERR_INVALID; }
	if (!n) return PBDX_OK;
	HIPCHECK(hipSetDevice(s->device));
	std::vector<float4> tmp(n);
	HIPCHECK(hipMemcpyAsync(tmp.data(), s->d_pos, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
	HIPCHECK(hipStreamSynchronize(s->stream));
	for (uint32_t i = 0; i < n; i++) { tmp[i].x = x[3 * i]; tmp[i].y = x[3 * i + 1]; tmp[i].z = x[3 * i + 2]; }
	HIPCHECK(hipMemcpyAsync(s->d_pos, tmp.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_get_particles(pbdx_solver *s, uint32_t n, float *x, float *v, float *old_x, float *last_x)
{
	if (!s || n != s->n) { set_error("get_particles: particle count mismatch (%u vs %u)", n, s ? s->n : 0); return PBDX_ERR_INVALID; }
	if (!n) return PBDX_OK;
	HIPCHECK(hipSetDevice(s->device));
	std::vector<float4> tmp(n);
	struct { float *dst; const float4 *src; } jobs[4] = { { x, s->d_pos }, { v, s->d_vel }, { old_x, s->d_old }, { last_x, s->d_last } };
	for (auto &j : jobs)
	{
		if (!j.dst) continue;
		HIPCHECK(hipMemcpyAsync(tmp.data(), j.src, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
		HIPCHECK(hipStreamSynchronize(s->stream));
		for (uint32_t i = 0; i < n; i++) { j.dst[3 * i] = tmp[i].x; j.dst[3 * i + 1] = tmp[i].y; j.dst[3 * i + 2] = tmp[i].z; }
	}
	return PBDX_OK;
}

int pbdx_solver_begin_schedule(pbdx_solver *s)
{
	if (!s) return PBDX_ERR_INVALID;
	HIPCHECK(hipSetDevice(s->device));
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->drop_graph();
	s->free_batches();
	s->schedule_open = true;
	s->schedule_version++;
	return PBDX_OK;
}

int pbdx_solver_add_batch(pbdx_solver *s, uint32_t group, int type, uint32_t count,
	const uint32_t *indices, const float *params, uint32_t param_stride)
{
	if (!s || !s->schedule_open) { set_error("add_batch outside begin_schedule/end_schedule"); return PBDX_ERR_INVALID; }
	const TypeInfo *ti = type_info(type);
	if (!ti) { set_error("add_batch: constraint type %d is not handled by the engine", type); return PBDX_ERR_UNSUPPORTED; }
	if (param_stride != ti->param_stride) { set_error("add_batch: %s expects param_stride %u, got %u", ti->name, ti->param_stride, param_stride); return PBDX_ERR_INVALID; }
	if (count == 0) return PBDX_OK;
	if (!indices || !params) { set_error("add_batch: null indices/params"); return PBDX_ERR_INVALID; }
	const uint32_t nb = ti->num_bodies;
	for (size_t i = 0; i < (size_t)count * nb; i++)
		if (indices[i] >= s->n) { set_error("add_batch: particle index %u out of range (%u particles uploaded)", indices[i], s->n); return PBDX_ERR_INVALID; }
	HIPCHECK(hipSetDevice(s->device));

	Batch b;
	b.type = type; b.group = group; b.count = count; b.seq = (uint32_t)s->batches.size();
	b.h_idx.assign(indices, indices + (size_t)count * nb);

	// indices: uint2 for 2-body, uint4 (padded) for 3- and 4-body constraints
	const uint32_t iw = (nb == 2) ? 2 : 4;
	std::vector<uint32_t> idx((size_t)count * iw, 0);
	for (uint32_t i = 0; i < count; i++)
		for (uint32_t k = 0; k < nb; k++) idx[(size_t)i * iw + k] = indices[(size_t)i * nb + k];
	HIPCHECK(hipMalloc(&b.d_idx, idx.size() * sizeof(uint32_t)));
	HIPCHECK(hipMemcpy(b.d_idx, idx.data(), idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice));

	// parameters: detect batch-uniform ones, lay the rest out planar
	memset(&b.pv, 0, sizeof(b.pv));
	const uint32_t np = ti->param_stride;
	const uint32_t stride = (count + 3u) & ~3u;
	uint32_t planes = 0;
	for (uint32_t k = 0; k < np; k++)
	{
		bool uniform = true;
		uint32_t first; memcpy(&first, &params[k], 4);
		for (uint32_t i = 1; i < count && uniform; i++)
		{
			uint32_t cur; memcpy(&cur, &params[(size_t)i * np + k], 4);
			uniform = (cur == first);
		}
		if (uniform) { b.pv.umask |= 1u << k; b.pv.u[k] = params[k]; }
		else b.pv.slot[k] = (uint8_t)planes++;
	}
	b.pv.stride = stride;
	if (planes)
	{
		std::vector<float> planar((size_t)planes * stride, 0.0f);
		for (uint32_t k = 0; k < np; k++)
		{
			if ((b.pv.umask >> k) & 1u) continue;
			float *dst = &planar[(size_t)b.pv.slot[k] * stride];
			for (uint32_t i = 0; i < count; i++) dst[i] = params[(size_t)i * np + k];
		}
		HIPCHECK(hipMalloc(&b.d_params, planar.size() * sizeof(float)));
		HIPCHECK(hipMemcpy(b.d_params, planar.data(), planar.size() * sizeof(float), hipMemcpyHostToDevice));
	}
	b.pv.base = b.d_params;
	if (ti->xpbd)
	{
		HIPCHECK(hipMalloc(&b.d_lambda, (size_t)count * sizeof(float)));
		HIPCHECK(hipMemset(b.d_lambda, 0, (size_t)count * sizeof(float)));
	}
	s->batches.push_back(std::move(b));
	return PBDX_OK;
}

int pbdx_solver_end_schedule(pbdx_solver *s)
{
	if (!s || !s->schedule_open) { set_error("end_schedule without begin_schedule"); return PBDX_ERR_INVALID; }
	s->order.resize(s->batches.size());
	for (uint32_t i = 0; i < s->order.size(); i++) s->order[i] = i;
	std::stable_sort(s->order.begin(), s->order.end(), [s](uint32_t a, uint32_t b) { return s->batches[a].group < s->batches[b].group; });
	s->schedule_open = false;
	s->schedule_version++;
	return PBDX_OK;
}

int pbdx_solver_validate_schedule(pbdx_solver *s)
{
	if (!s) return PBDX_ERR_INVALID;
	std::vector<uint32_t> stamp(s->n, 0xffffffffu);
	uint32_t cur = 0;
	for (size_t oi = 0; oi < s->order.size(); oi++)
	{
		const Batch &b = s->batches[s->order[oi]];
		if (oi == 0 || b.group != s->batches[s->order[oi - 1]].group) cur++;
		for (uint32_t p : b.h_idx)
		{
			if (stamp[p] == cur) { set_error("colour group %u touches particle %u more than once", b.group, p); return PBDX_ERR_INVALID; }
			stamp[p] = cur;
		}
	}
	return PBDX_OK;
}

int pbdx_solver_set_option(pbdx_solver *s, int option, int64_t value)
{
	if (!s) return PBDX_ERR_INVALID;
	switch (option)
	{
	case PBDX_OPT_USE_GRAPH: s->use_graph = value != 0; break;
	case PBDX_OPT_BLOCK_SIZE:
		if (value != 64 && value != 128 && value != 256) { set_error("block size must be 64, 128 or 256"); return PBDX_ERR_INVALID; }
		s->block_size = (int)value; break;
	case PBDX_OPT_XCD_REMAP: s->xcd_remap = value != 0; break;
	default: set_error("unknown option %d", option); return PBDX_ERR_INVALID;
	}
	s->drop_graph();
	return PBDX_OK;
}

int pbdx_solver_set_profiling(pbdx_solver *s, int on)
{
	if (!s) return PBDX_ERR_INVALID;
	s->profile = on != 0;
	return PBDX_OK;
}

int pbdx_solver_step(pbdx_solver *s, float h, uint32_t sub_steps, uint32_t max_iterations,
	int vel, const float gravity[3], uint32_t num_steps)
{
	if (!s || !gravity || sub_steps == 0) { set_error("step: bad arguments"); return PBDX_ERR_INVALID; }
	if (s->schedule_open) { set_error("step: schedule still open"); return PBDX_ERR_INVALID; }
	HIPCHECK(hipSetDevice(s->device));
	const float hs = h / (float)sub_steps;                  // TimeStepController.cpp:91
	const float inv_h = (float)(1.0 / (double)hs);          // TimeIntegration.cpp:50 evaluates 1.0/h in double

	uint64_t proj_per_sweep = 0, bytes_per_sweep = 0;
	for (const Batch &b : s->batches) { proj_per_sweep += b.count; bytes_per_sweep += (uint64_t)b.count * type_info(b.type)->algorithmic_bytes; }
	const uint64_t substeps_total = (uint64_t)sub_steps * num_steps;
	s->stats = pbdx_step_stats();
	s->stats.projections = proj_per_sweep * max_iterations * substeps_total;
	s->stats.kernel_launches = ((uint64_t)s->order.size() * max_iterations + 2) * substeps_total;
	s->stats.algorithmic_bytes = (bytes_per_sweep * max_iterations + (uint64_t)s->n * 140) * substeps_total;
	memset(s->type_ms, 0, sizeof(s->type_ms));
	memset(s->type_launches, 0, sizeof(s->type_launches));
	memset(s->type_projections, 0, sizeof(s->type_projections));

	if (s->profile)
	{
		HIPCHECK(hipEventRecord(s->ev_start, s->stream));
		for (uint64_t k = 0; k < substeps_total; k++)
		{
			ProfCursor pc; pc.s = s;
			int r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, &pc);
			if (r) return r;
			r = collect_profile(s, &pc);
			if (r) return r;
		}
		HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
	}
	else if (s->use_graph)
	{
		pbdx_solver::GraphKey k = { hs, max_iterations, vel, { gravity[0], gravity[1], gravity[2] }, s->schedule_version, s->block_size, s->xcd_remap, s->n };
		if (!s->graph_valid || memcmp(&k, &s->key, sizeof(k)) != 0)
		{
			s->drop_graph();
			HIPCHECK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
			int r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr);
			hipGraph_t g = nullptr;
			hipError_t e = hipStreamEndCapture(s->stream, &g);
			if (r) { if (g) (void)hipGraphDestroy(g); return r; }
			if (e != hipSuccess) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return PBDX_ERR_HIP; }
			s->graph = g;
			HIPCHECK(hipGraphInstantiate(&s->graph_exec, s->graph, nullptr, nullptr, 0));
			s->key = k;
			s->graph_valid = true;
		}
		HIPCHECK(hipEventRecord(s->ev_start, s->stream));
		for (uint64_t k2 = 0; k2 < substeps_total; k2++)
			HIPCHECK(hipGraphLaunch(s->graph_exec, s->stream));
		HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
	}
	else
	{
		HIPCHECK(hipEventRecord(s->ev_start, s->stream));
		for (uint64_t k = 0; k < substeps_total; k++)
		{
			int r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr);
			if (r) return r;
		}
		HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
	}
	HIPCHECK(hipStreamSynchronize(s->stream));
	float ms = 0.0f;
	HIPCHECK(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
	s->stats.total_ms = ms;
	return PBDX_OK;
}

int pbdx_solver_project(pbdx_solver *s, float h_sub, uint32_t iterations)
{
	if (!s || s->schedule_open) { set_error("project: bad state"); return PBDX_ERR_INVALID; }
	HIPCHECK(hipSetDevice(s->device));
	int r = projection_sweeps(s, h_sub, iterations, nullptr);
	if (r) return r;
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_synchronize(pbdx_solver *s)
{
	if (!s) return PBDX_ERR_INVALID;
	HIPCHECK(hipSetDevice(s->device));
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_get_lambdas(pbdx_solver *s, uint32_t batch_index, uint32_t count, float *out)
{
	if (!s || !out || batch_index >= s->batches.size()) { set_error("get_lambdas: bad batch"); return PBDX_ERR_INVALID; }
	const Batch &b = s->batches[batch_index];
	if (!b.d_lambda || count != b.count) { set_error("get_lambdas: batch has no multipliers or count mismatch"); return PBDX_ERR_INVALID; }
	HIPCHECK(hipSetDevice(s->device));
	HIPCHECK(hipStreamSynchronize(s->stream));
	HIPCHECK(hipMemcpy(out, b.d_lambda, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
	return PBDX_OK;
}

int pbdx_solver_get_stats(pbdx_solver *s, pbdx_step_stats *out)
{
	if (!s || !out) return PBDX_ERR_INVALID;
	*out = s->stats;
	return PBDX_OK;
}

int pbdx_solver_get_type_stats(pbdx_solver *s, int type, double *ms, uint64_t *launches, uint64_t *projections)
{
	if (!s || type < 0 || type >= PBDX_NUM_CONSTRAINT_TYPES) return PBDX_ERR_INVALID;
	if (ms) *ms = s->type_ms[type];
	if (launches) *launches = s->type_launches[type];
	if (projections) *projections = s->type_projections[type];
	return PBDX_OK;
}

int pbdx_solver_describe(pbdx_solver *s, char *buf, size_t n)
{
	if (!s || !buf || !n) return PBDX_ERR_INVALID;
	uint64_t nc = 0; uint32_t ng = 0, last = 0xffffffffu;
	for (uint32_t bi : s->order) { nc += s->batches[bi].count; if (s->batches[bi].group != last) { ng++; last = s->batches[bi].group; } }
	snprintf(buf, n, "device=%d name=%s arch=%s CUs=%d particles=%u constraints=%llu groups=%u batches=%zu graph=%d block=%d xcd_remap=%d",
		s->device, s->prop.name, s->prop.gcnArchName, s->prop.multiProcessorCount, s->n, (unsigned long long)nc, ng, s->batches.size(),
		s->use_graph, s->block_size, s->xcd_remap);
	return PBDX_OK;
}

} // extern "C"
