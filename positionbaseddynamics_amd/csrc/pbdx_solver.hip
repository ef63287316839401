// pbdx_solver.hip -- the device engine: particle state in HBM, the colour-ordered constraint
// schedule, HIP kernels for gfx950 and the substep loop.
//
// Replaces the inner loop of PBD::TimeStepController::step
// (Simulation/TimeStepController.cpp:75-176) and positionConstraintProjection
// (:251-295) of the reference for particle scenes.
//
// HBM layout
//   pos[2] [n] float4  (x, y, z, invMass)  double buffered; gathered/scattered by the projections:
//                                          ONE 16-byte access per endpoint
//   vel    [n] float4  (vx, vy, vz, mass)  streamed by integrate / velocity update only
//   old [n] float4, last [n] float4        oldX / lastX (w unused)
//
// Two device schedules execute the same per-constraint code (pbdx_access.h / pbdx_project.h):
//
//  (A) colour-fused tiles (default; planner in pbdx_plan.cpp).  A run of consecutive colours is
//      ONE launch: a workgroup stages the dependency closure of its tile of particles in LDS
//      (float4 per particle, up to 160 KiB), sweeps its constraints colour by colour with a
//      workgroup barrier between colours and writes back the particles it owns.  Per slot it
//      streams 16-bit tile-local indices, the parameter record (planes, or vector segments for
//      workgroups up to 512 threads; wide records that repeat on regular meshes -- bending matrices,
//      FEM rest geometry -- as a 2-byte offset into the tile's table of distinct records, which sits
//      in LDS behind the particles: pbdx_plan.h) and the XPBD multiplier once from HBM; positions
//      never leave the CU during the segment.  A 27-colour cloth sweep becomes 2-4 launches instead of 29.
//  (A') the same tiles and passes as ONE persistent launch per substep: a tile starts its next pass as
//      soon as its neighbouring tiles have published theirs (persistent_kernel; default where measured
//      faster; needs every workgroup resident, checked by a handshake before anything is modified).
//  (B) one launch per (colour, type) batch straight on the HBM/L2-resident position array
//      (32-bit indices).  Used when a plan cannot be built, for per-type profiling, and as the
//      cross-check of (A) in the tests (the two are bit-identical by construction).
//
// One substep (integrate, iterations x schedule, velocity update) is captured into a hipGraph.
// No MFMA: there is no dense contraction on this path; everything is gather/scatter streaming.
//
// This file: the particle, contact and collision kernels, the host engine (plan image, schedules, substep loop, transfers) and the C ABI of include/pbdx.h.
// The constraint-sweep kernels of (A), (A') and (B) are pbdx_sweep.hip (shared declarations: pbdx_sweep.h), the calibration kernels pbdx_calib.hip.
#include <hip/hip_runtime.h>
#include "pbdx_internal.h"
#include "pbdx_device.h"
#include "pbdx_sweep.h"
#include "pbdx_plan.h"
#include "pbdx_contact.h"
#include "pbdx_tetcontact_dev.h"
#include <chrono>
#include <mutex>
#include <array>
#include <unordered_map>
#include <thread>
#include <vector>
#include <algorithm>
#include <type_traits>
#include <string.h>

using namespace pbdx;

namespace {
// ------------------------------------------------------------------------------------------------
// particle kernels
// ------------------------------------------------------------------------------------------------
// lastX <- oldX; oldX <- x; semi-implicit Euler for dynamic particles
// TimeStepController.cpp:112-118 + TimeIntegration.cpp:7-19 (acceleration == gravity for every
// dynamic particle, TimeStep.cpp:28-62).  Reads pos_in, writes pos_out (the same buffer, or the
// buffer the first fused segment of the substep reads).
// `ctl` (or null): control block of the persistent schedule; once a persistent launch has refused to start
// (ctl[kCtlAbort]) every later kernel of the same pbdx_solver_step call is a no-op, so that the host finds the state
// exactly as it was when the refusal happened and completes the step with the multi-launch schedule.
__global__ __launch_bounds__(256) void integrate_kernel(const float4 *pos_in, float4 *pos_out, float4 *__restrict__ vel,
	float4 *__restrict__ old, float4 *__restrict__ last, uint32_t n, float h, float gx, float gy, float gz, const uint32_t *ctl)
{
	if (ctl && ctl[kCtlAbort]) return;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 o = old[i];
	float4 p = pos_in[i];
	last[i] = o;
	old[i] = p;
	float4 v = vel[i];
	if (v.w != 0.0f)   // mass != 0
	{
		v.x = v.x + gx * h; v.y = v.y + gy * h; v.z = v.z + gz * h;
		p.x = p.x + v.x * h; p.y = p.y + v.y * h; p.z = p.z + v.z * h;
		vel[i] = v;
	}
	pos_out[i] = p;
}

// TimeIntegration::velocityUpdateFirstOrder / SecondOrder  TimeIntegration.cpp:42-51, 69-79
__global__ __launch_bounds__(256) void velocity_kernel(const float4 *__restrict__ pos, float4 *__restrict__ vel,
	const float4 *__restrict__ old, const float4 *__restrict__ last, uint32_t n, float inv_h, int second_order, uint32_t *ctl)
{
	if (ctl && ctl[kCtlAbort]) return;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (ctl && i == 0) ctl[kCtlSubstep] = ctl[kCtlSubstep] + 1u;      // substeps completed in this call (single writer)
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

// ---- host boundary: packed xyz (std::vector<Vector3r> of a float build, ParticleData.h:91-100) <-> float4
// `src`/`dst` are device staging copies of the caller's arrays; w = inv_mass / mass / 0
// (T = float: a float build of the reference; T = double: the default build -- the conversion to the device's fp32 happens
// here, on the device, as one rounding per value exactly like a host-side (float) cast)
template <class T>
__global__ __launch_bounds__(256) void pack_kernel(const T *__restrict__ xyz, const T *__restrict__ w, float4 *__restrict__ dst, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	dst[i] = make_float4((float)xyz[3 * (size_t)i], (float)xyz[3 * (size_t)i + 1], (float)xyz[3 * (size_t)i + 2], w ? (float)w[i] : 0.0f);
}
template <class T>
__global__ __launch_bounds__(256) void pack_zero_kernel(const T *__restrict__ w, float4 *__restrict__ dst, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	dst[i] = make_float4(0.0f, 0.0f, 0.0f, (float)w[i]);
}
template <class T>
__global__ __launch_bounds__(256) void unpack_kernel(const float4 *__restrict__ src, T *__restrict__ xyz, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 v = src[i];
	xyz[3 * (size_t)i] = (T)v.x; xyz[3 * (size_t)i + 1] = (T)v.y; xyz[3 * (size_t)i + 2] = (T)v.z;
}
// partial upload (pbdx_solver_update_particle_ranges): elements [first, first + count) of one staged array
template <class T>
__global__ __launch_bounds__(256) void update_xyz_kernel(const T *__restrict__ xyz, float4 *__restrict__ dst, uint32_t first, uint32_t count)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= count) return;
	const size_t i = (size_t)first + k;
	float4 v = dst[i];
	v.x = (float)xyz[3 * i]; v.y = (float)xyz[3 * i + 1]; v.z = (float)xyz[3 * i + 2];
	dst[i] = v;
}
template <class T>
__global__ __launch_bounds__(256) void update_w_kernel(const T *__restrict__ w, float4 *__restrict__ dst, uint32_t first, uint32_t count)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= count) return;
	const size_t i = (size_t)first + k;
	dst[i].w = (float)w[i];
}
// block hashes of a staged host array (include/pbdx.h: pbdx_hash_word / pbdx_hash_block): one workgroup per block of
// PBDX_HASH_BLOCK elements, XOR over the block's 32-bit words
__global__ __launch_bounds__(256) void hash_blocks_kernel(const uint32_t *__restrict__ words32, uint64_t total_bytes, uint32_t bytes_per_block, uint64_t *__restrict__ out)
{
	__shared__ uint64_t part[4];
	const uint64_t first = (uint64_t)blockIdx.x * bytes_per_block;
	uint64_t last = first + bytes_per_block;
	if (last > total_bytes) last = total_bytes;
	uint64_t h = 0;
	for (uint64_t o = first + 8ull * threadIdx.x; o < last; o += 8ull * 256)
	{
		// (assembled from two 32-bit loads: the staging slot of array q starts at 3 n q sizeof(T), 4-byte aligned only for odd n)
		const uint64_t w = o + 8 <= last ? ((uint64_t)words32[o >> 2] | ((uint64_t)words32[(o >> 2) + 1] << 32)) : (uint64_t)words32[o >> 2];      // an odd 32-bit word at the very end of the array
		h ^= pbdx_hash_word(w, (uint32_t)(o >> 3));
	}
	for (int o = 32; o > 0; o >>= 1) h ^= __shfl_xor(h, o, 64);
	if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = h;
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = part[0] ^ part[1] ^ part[2] ^ part[3];
}
__global__ __launch_bounds__(256) void set_xyz_kernel(const float *__restrict__ xyz, float4 *__restrict__ dst, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float4 v = dst[i];
	v.x = xyz[3 * (size_t)i]; v.y = xyz[3 * (size_t)i + 1]; v.z = xyz[3 * (size_t)i + 2];
	dst[i] = v;
}

// ---- contacts with static colliders: one particle per lane (pbdx_contact.h) -----------------------------
struct DynContact { uint32_t particle, collider, range, pad; float cp_w[3], n_w[3]; };      // an entry of the sequential list (dynamic bodies)
struct ContactArgs
{
	const float4 *pos;
	float4 *vel;
	const pbdx_collider *colliders;
	uint32_t num_colliders;
	uint32_t first, count;          // particle range of one collision model
	float tolerance, stiffness, restitution, friction;
	uint32_t iterations;
	unsigned int *counters;         // [0] contacts, [1] overflow flag
	const uint32_t *ctl;            // see integrate_kernel
	const uint8_t *imp_mark;        // or null: particles whose velocity chain also holds particle-tet contact impulses (run by tet_impulse_kernel)
	// dynamic bodies (include/pbdx.h: dynamic rigid bodies as impulse sinks) or dyn == null: a particle that touches one hands ALL its contacts to the
	// sequential list instead of solving them here
	const pbdx_collider_dynamics *dyn;
	const uint32_t *rank;           // per particle: position in its collision object's point hierarchy
	uint32_t pair_base;             // (position of the range among the ranges ordered by object index) * num_colliders
	const uint8_t *collider_pos;    // per collider: its position among the colliders ordered by object index
	uint32_t range_index;
	DynContact *list; uint32_t *list_keys, *list_vals; uint32_t list_cap;
	unsigned int *list_counters;    // [0] entries, [1] overflow
};
__global__ __launch_bounds__(256) void contact_kernel(ContactArgs a)
{
	if (a.ctl && a.ctl[kCtlAbort]) return;
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= a.count) return;
	const uint32_t i = a.first + k;
	if (a.imp_mark && a.imp_mark[i]) return;
	const float4 p = a.pos[i];
	float4 vv = a.vel[i];
	V3 v = mk(vv.x, vv.y, vv.z);
	if (a.dyn)
	{
		RawContact raw[PBDX_MAX_CONTACTS_PER_PARTICLE];
		const int n = detect_particle_contacts(mk(p.x, p.y, p.z), a.colliders, a.num_colliders, a.tolerance, raw);
		if (n < 0) { atomicExch(&a.counters[1], 1u); return; }
		bool dynamic = false;
		for (int q = 0; q < n; q++) if (a.dyn[raw[q].collider].inv_mass != 0.0f) dynamic = true;
		if (dynamic)
		{
			// the reference's place of every contact: pair (range object, collider object) lexicographically, then the particle's place in the
			// range's point hierarchy (DistanceFieldCollisionDetection.cpp:34-47, kdTree.inl:84-105)
			atomicAdd(&a.counters[0], (unsigned int)n);
			const uint32_t at = atomicAdd(&a.list_counters[0], (unsigned int)n);
			if (at + (uint32_t)n > a.list_cap) { atomicExch(&a.list_counters[1], 1u); return; }
			for (int q = 0; q < n; q++)
			{
				DynContact c;
				c.particle = i; c.collider = raw[q].collider; c.range = a.range_index; c.pad = 0u;
				c.cp_w[0] = raw[q].cp_w.x; c.cp_w[1] = raw[q].cp_w.y; c.cp_w[2] = raw[q].cp_w.z;
				c.n_w[0] = raw[q].n_w.x; c.n_w[1] = raw[q].n_w.y; c.n_w[2] = raw[q].n_w.z;
				a.list[at + q] = c;
				a.list_keys[at + q] = ((a.pair_base + a.collider_pos[raw[q].collider]) << 24) | (a.rank[i] & 0xffffffu);
				a.list_vals[at + q] = at + q;
			}
			return;
		}
		// static bodies only: the independent chain, on the contacts just found
		if (n > 0)
		{
			solve_particle_contacts(mk(p.x, p.y, p.z), v, p.w, vv.w, a.colliders, raw, n, a.stiffness, a.restitution, a.friction, a.iterations, NoExtraImpulses());
			atomicAdd(&a.counters[0], (unsigned int)n);
			a.vel[i] = make_float4(v.x, v.y, v.z, vv.w);
		}
		return;
	}
	const int nc = particle_contacts(mk(p.x, p.y, p.z), v, p.w, vv.w, a.colliders, a.num_colliders, a.tolerance, a.stiffness,
		a.restitution, a.friction, a.iterations);
	if (nc < 0) { atomicExch(&a.counters[1], 1u); return; }
	if (nc > 0)
	{
		atomicAdd(&a.counters[0], (unsigned int)nc);
		a.vel[i] = make_float4(v.x, v.y, v.z, vv.w);
	}
}

// The sequential part of the contact velocity solve with dynamic bodies (include/pbdx.h): the list in the reference's order (sorted keys), its contacts
// initialised in parallel from the pre-solve velocities (ParticleRigidBodyContactConstraint::initConstraint, Constraints.cpp:2115-2146), then ONE lane walks
// the list `iterations` times (TimeStepController.cpp:342-350) with the bodies' velocities in registers / LDS; the loads of contact j + 1 are issued before
// contact j is solved (the chain is through the body, not through memory).
struct DynSolveArgs
{
	const float4 *pos; float4 *vel;
	const pbdx_collider *colliders; const pbdx_collider_dynamics *dyn; uint32_t num_colliders;
	const pbdx_collision_range *ranges;
	const DynContact *list; const uint32_t *order;      // order[j] = list entry at place j of the reference's order
	DynContactInfo *info; float *sum_impulses;
	const unsigned int *list_counters;
	float *body;                    // per collider: v (3), omega (3), pad (2): in = the colliders' velocities, out = after the solve
	float stiffness; uint32_t iterations;
	const uint32_t *ctl;
};
constexpr uint32_t kMaxDynColliders = 32;
__global__ __launch_bounds__(256) void dyn_contact_solve_kernel(DynSolveArgs a)
{
	if (a.ctl && a.ctl[kCtlAbort]) return;
	__shared__ float s_body[kMaxDynColliders][8];
	const uint32_t count = a.list_counters[1] ? 0u : a.list_counters[0];
	for (uint32_t k = threadIdx.x; k < a.num_colliders; k += blockDim.x)
	{
		const pbdx_collider &c = a.colliders[k];
		s_body[k][0] = c.body_v[0]; s_body[k][1] = c.body_v[1]; s_body[k][2] = c.body_v[2];
		s_body[k][3] = c.body_omega[0]; s_body[k][4] = c.body_omega[1]; s_body[k][5] = c.body_omega[2];
	}
	for (uint32_t j = threadIdx.x; j < count; j += blockDim.x)
	{
		const DynContact c = a.list[a.order[j]];
		const pbdx_collider &col = a.colliders[c.collider];
		const pbdx_collider_dynamics &d = a.dyn[c.collider];
		const float4 x = a.pos[c.particle], vv = a.vel[c.particle];
		DynContactInfo ci;
		dyn_contact_init(x.w, mk(vv.x, vv.y, vv.z), d.inv_mass, mk(col.com[0], col.com[1], col.com[2]), mk(col.body_v[0], col.body_v[1], col.body_v[2]), d.inertia_inv_w,
			mk(col.body_omega[0], col.body_omega[1], col.body_omega[2]), mk(x.x, x.y, x.z), mk(c.cp_w[0], c.cp_w[1], c.cp_w[2]), mk(c.n_w[0], c.n_w[1], c.n_w[2]),
			a.ranges[c.range].restitution * col.restitution, ci);
		a.info[j] = ci;
		a.sum_impulses[j] = 0.0f;
	}
	__syncthreads();
	if (threadIdx.x == 0 && count)
	{
		for (uint32_t it = 0; it < a.iterations; it++)
		{
			DynContact c = a.list[a.order[0]];
			float4 vv = a.vel[c.particle], x = a.pos[c.particle];
			DynContactInfo ci = a.info[0];
			float sum = a.sum_impulses[0];
			for (uint32_t j = 0; j < count; j++)
			{
				// what contact j + 1 needs from memory, requested now (its particle's velocity only if it is another particle: otherwise the value computed below)
				const uint32_t jn = j + 1u < count ? j + 1u : j;
				const DynContact cn = a.list[a.order[jn]];
				const bool same = cn.particle == c.particle;
				float4 vn = make_float4(0.f, 0.f, 0.f, 0.f), xn = x;
				if (!same) { vn = a.vel[cn.particle]; xn = a.pos[cn.particle]; }
				const DynContactInfo cin = a.info[jn];
				const float sumn = a.sum_impulses[jn];
				const pbdx_collider &col = a.colliders[c.collider];
				const pbdx_collider_dynamics &d = a.dyn[c.collider];
				float *b = s_body[c.collider];
				V3 v0 = mk(vv.x, vv.y, vv.z), v1 = mk(b[0], b[1], b[2]), w1 = mk(b[3], b[4], b[5]);
				dyn_contact_velocity_solve(x.w, vv.w, v0, d.inv_mass, mk(col.com[0], col.com[1], col.com[2]), v1, d.inertia_inv_w, w1, a.stiffness,
					a.ranges[c.range].friction + col.friction, sum, ci);
				b[0] = v1.x; b[1] = v1.y; b[2] = v1.z; b[3] = w1.x; b[4] = w1.y; b[5] = w1.z;
				vv = make_float4(v0.x, v0.y, v0.z, vv.w);
				a.vel[c.particle] = vv;
				a.sum_impulses[j] = sum;
				if (!same) vv = vn;
				x = xn; c = cn; ci = cin; sum = sumn;
			}
		}
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < a.num_colliders; k += blockDim.x)
		for (int q = 0; q < 6; q++) a.body[k * 8u + q] = s_body[k][q];
}

// ---- contacts between deformable solids (pbdx_tetcontact.h, pbdx_tetcontact_dev.h) -------------------------------------------
// Per step, after the substeps: refresh the bounding spheres of every collider's point and tet hierarchy, the colliders' boxes, then
// the detection.  Its parallel form is in pbdx_tetcontact_dev.h; the two kernels below are the reference's control flow in ONE thread
// (PBDX_OPT_TET_CONTACTS_SERIAL), kept as the in-engine cross-check of the parallel form: both must produce the same list and state.
__global__ __launch_bounds__(256) void tet_hull_kernel(const TetColliderView *views, uint32_t collider, const P4 *pos, int which)
{
	const TetColliderView &v = views[collider];
	const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
	if (which == 0) { if (node < v.points.num_nodes) hull_points(v.points, node, pos + v.first); }
	else if (node < v.tet_bvh.num_nodes) hull_tets(v.tet_bvh, node, pos + v.first, v.tets, v.tolerance);
}
// CollisionDetection::updateAABB: min / max over the model's particles (order-independent)
__global__ __launch_bounds__(256) void tet_aabb_kernel(const TetColliderView *views, const P4 *pos, float *aabb)
{
	__shared__ float lo[3][256], hi[3][256];
	const TetColliderView &v = views[blockIdx.x];
	float mn[3], mx[3];
	const P4 p0 = pos[v.first];
	mn[0] = mx[0] = p0.x; mn[1] = mx[1] = p0.y; mn[2] = mx[2] = p0.z;
	for (uint32_t i = threadIdx.x; i < v.num_vertices; i += blockDim.x)
	{
		const P4 p = pos[v.first + i];
		const float q[3] = { p.x, p.y, p.z };
		for (int k = 0; k < 3; k++) { if (mn[k] > q[k]) mn[k] = q[k]; if (mx[k] < q[k]) mx[k] = q[k]; }
	}
	for (int k = 0; k < 3; k++) { lo[k][threadIdx.x] = mn[k]; hi[k][threadIdx.x] = mx[k]; }
	__syncthreads();
	for (uint32_t stride = 128; stride > 0; stride >>= 1)
	{
		if (threadIdx.x < stride)
			for (int k = 0; k < 3; k++)
			{
				if (lo[k][threadIdx.x] > lo[k][threadIdx.x + stride]) lo[k][threadIdx.x] = lo[k][threadIdx.x + stride];
				if (hi[k][threadIdx.x] < hi[k][threadIdx.x + stride]) hi[k][threadIdx.x] = hi[k][threadIdx.x + stride];
			}
		__syncthreads();
	}
	if (threadIdx.x < 3) { aabb[6 * blockIdx.x + threadIdx.x] = lo[threadIdx.x][0]; aabb[6 * blockIdx.x + 3 + threadIdx.x] = hi[threadIdx.x][0]; }
}
__global__ void tet_detect_kernel(const TetColliderView *views, uint32_t n, const P4 *pos, const P4 *rest, const P4 *vel, const float *aabb, TetContact *contacts, uint32_t *counters, uint32_t max_contacts)
{
	if (blockIdx.x || threadIdx.x) return;
	uint32_t found = 0;
	bool ok = true;
	for (uint32_t i = 0; i < n; i++)
		for (uint32_t k = 0; k < n; k++)
		{
			if (i == k || !views[i].test_mesh || !aabb_intersect(aabb + 6 * i, aabb + 6 * k)) continue;
			ok = tet_pair_contacts(views[i], views[k], pos, rest, vel, [&](const TetContact &c) {
				if (found < max_contacts) contacts[found] = c;
				found++;
			}) && ok;
		}
	counters[kTcCount] = found < max_contacts ? found : max_contacts;
	if (found > max_contacts) counters[kTcOverflow] = 1u;
	if (!ok) counters[kTcStack] = 1u;
}
// TimeStepController.cpp:288-291: after the colour groups of an iteration, the contact list sequentially
__global__ void tet_contact_solve_kernel(float4 *pos, const TetContact *contacts, const uint32_t *counters)
{
	if (blockIdx.x || threadIdx.x) return;
	TetPosAccess acc = { pos };
	const uint32_t n = counters[kTcCount];
	for (uint32_t i = 0; i < n; i++) tet_contact_position_solve(contacts[i], acc);
}
struct Batch
{
	int type = 0;
	uint32_t group = 0;
	uint32_t count = 0;
	uint32_t seq = 0;               // add_batch order
	uint32_t *d_idx = nullptr;
	float *d_params = nullptr;
	float *d_lambda = nullptr;
	uint32_t par_stride = 0;
	TypeView view;
	std::vector<uint32_t> h_idx;    // kept for validate_schedule and the planner
	std::vector<float> h_params;    // host image of the parameter records (planner, lazy device arrays)
};

struct DeviceSegment
{
	TileDev *d_tiles = nullptr;
	FusedChunk *d_chunks = nullptr;
	uint16_t *d_idx = nullptr;
	float *d_params = nullptr;
	float *d_lambda = nullptr;
	uint32_t *d_gid = nullptr;
	unsigned long long *d_trace = nullptr;
	uint32_t *d_dep_off = nullptr, *d_dep_tile = nullptr;   // persistent schedule: per tile, the tiles it waits for
	uint32_t num_tiles = 0;
	uint32_t lds_bytes = 0;
	uint32_t idx_bytes = 0, params_bytes = 0, lambda_bytes = 0;
	uint32_t gid_count = 0, chunk_count = 0, n_particles = 0, dep_count = 0;      // sizes the debug build (PBDX_BOUNDS) checks raw indices against
	uint32_t type_mask = 0;
	uint64_t algorithmic_bytes = 0;  // SURVEY 8d bytes of the DISTINCT constraints of the segment
	uint64_t constraints = 0;
	fused_fn kernel = nullptr;
	int block = 0;
	// profiling
	double ms = 0.0;
	uint64_t launches = 0;
};

#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return PBDX_ERR_HIP; } } while (0)


} // namespace

// Host transfers at the PCIe rate (PBDX_OPT_PIN_HOST) go through a page-locked MIRROR the engine owns (HostMirror below), not through
// hipHostRegister of the caller's arrays as they did until round 5.  Registering maps the caller's heap pages into the GPU's address space at
// their HOST addresses, the runtime finds a registration by address, and the whole arrangement depends on what the host's allocator does with
// the neighbouring heap afterwards; a full GPU test run of round 5 died with "Memory access fault by GPU" at a host heap address in the middle
// of a plug-in test, one run in four (profiles/HISTORY.md [9]).  With the mirror the GPU is never given an address of memory the engine does
// not own: the device copies to / from the mirror, host threads copy between the mirror and the caller's arrays, array by array, so that the
// copy of one array runs while the next one is on the bus.
using pbdx::host_copy;

struct pbdx_solver
{
	int device = 0;
	hipStream_t stream = nullptr;
	hipDeviceProp_t prop;
	uint32_t n = 0;
	float4 *d_pos[2] = { nullptr, nullptr };
	float4 *d_vel = nullptr, *d_old = nullptr, *d_last = nullptr;
	uint64_t *d_hash = nullptr; uint32_t hash_blocks = 0;   // block hashes of the staged arrays (pbdx_solver_get_particles_hashed)
	float *d_stage = nullptr;            // 4 x 3n + 2n values: device staging of the caller's packed arrays (floats; doubles once a double host called)
	size_t stage_elem = sizeof(float);
	std::vector<float> h_x;              // positions at upload time (tile partition only)
	std::vector<Batch> batches;          // in add order
	std::vector<uint32_t> order;         // batch indices sorted by (group, seq)
	bool schedule_open = false;
	uint64_t schedule_version = 0;
	bool params_dirty = false;           // pbdx_solver_update_batch_params since the last commit
	uint32_t inst_particles = 0, inst_count = 1;   // pbdx_solver_set_instancing: the schedule is inst_count congruent instances (hint, verified)

	// options
	int use_graph = 1;
	int block_size = 256;
	int xcd_remap = 1;
	int profile = 0;
	int fuse = 2;                        // 0 per-colour, 1 fused, 2 auto (fused; measured choice when compute-heavy types are present)
	int fuse_choice = 1;                 // outcome of the auto mode for the current schedule
	int autotune_rounds = 0; float autotune_spent_ms = 0.0f;      // measured rounds after the warm-up round, and the device time they took
	float autotune_ms[3] = { 0.0f, 0.0f, 0.0f }; // measured time of 12 sweeps: per-colour / fused / fused persistent (0 = not measured)
	bool persist_choice = false;         // outcome of the measurement for the one-launch form
	bool autotuned_for_tet = false;      // ... taken with / without deformable colliders installed (they select the per-iteration form): re-measured when that changes
	uint32_t tile_particles = 0;
	int fuse_block = 0;                  // 0 = auto
	uint32_t max_segment_colours = 16;
	uint32_t lds_particles = 10240;
	int trace = 0;
	int pin_host = 0;                    // PBDX_OPT_PIN_HOST: particle transfers through the page-locked mirror below
	char *h_mirror = nullptr;            // page-locked host image of d_stage (same layout) + the block hashes behind it
	size_t mirror_bytes = 0;
	hipEvent_t mirror_ev[4] = { nullptr, nullptr, nullptr, nullptr };
	int persistent = 1;                  // PBDX_OPT_PERSISTENT: the sweeps of a substep as one launch (A'): 0 never, 1 where measured faster, 2 always, 3 self-test
	bool persist_ok = false;             // the plan is eligible (and no launch has been refused or has timed out)
	uint32_t persist_refusals = 0;
	uint32_t persist_timeouts = 0;       // calls in which a tile gave up waiting for a neighbour (state restored, call repeated with schedule (A))
	uint32_t persist_timeout_ms = 250;   // PBDX_OPT_PERSISTENT_TIMEOUT_MS
	uint32_t persist_wgs_per_cu = 1;     // PBDX_OPT_PERSISTENT_WGS_PER_CU: tiles resident per CU in the one-launch schedule
	float4 *d_ckpt[4] = { nullptr, nullptr, nullptr, nullptr };   // pbdx_solver_save_state / restore_state: the caller's checkpoint of pos / vel / old / last
	uint32_t ckpt_n = 0;                                          // particles in the checkpoint (0 = none)
	float4 *d_snap[4] = { nullptr, nullptr, nullptr, nullptr };   // pos / vel / old / last as they were when the current call started (persistent schedule only)
	bool last_folded = false;            // the substeps enqueued last ran integrate / velocity update inside the persistent launch
	bool last_flips = false;             // ... and ended in the other position buffer (odd number of passes): swap_state() after each
	double persist_ms = 0.0;             // last profiled step: summed duration / number of persistent launches
	uint64_t persist_launches = 0;
	persist_fn persist_kernel = nullptr;
	int persist_block = 0;
	uint32_t persist_lds = 0, persist_grid = 0;
	uint32_t ids_halo_off16 = 0, ids_bnd_off16 = 0, ids_halo_cap = 0, ids_bnd_cap = 0;      // particle ids resident in LDS (LdsIds)
	bool pack_plain = getenv("PBDX_NO_PACK") == nullptr;     // packed records for compact one-plane types (build_idx_image; the switch is a developer A/B aid)
	uint32_t *d_epoch = nullptr;
	uint32_t *h_error = nullptr, *d_error = nullptr;   // page-locked host words the persistent kernel raises: [0] timeout, [1] launch refused, [2] at substep
	uint32_t *d_ctl = nullptr;           // kCtl* device words

	// contacts with static colliders
	std::vector<pbdx_collider> colliders;
	std::vector<pbdx_collision_range> ranges;
	pbdx_collider *d_colliders = nullptr;
	unsigned int *d_contact_counters = nullptr;
	float contact_tolerance = 0.01f, contact_stiffness = 100.0f;
	uint32_t max_iterations_v = 5;
	uint64_t contact_version = 0;
	// ... with dynamic bodies (include/pbdx.h): the sequential list and what orders it
	std::vector<pbdx_collider_dynamics> dynamics;      // per collider, or empty: every body static
	std::vector<uint32_t> range_object;                // per range: its object index
	pbdx_collider_dynamics *d_dynamics = nullptr;
	uint32_t *d_rank = nullptr; uint32_t rank_count = 0;
	uint8_t *d_collider_pos = nullptr;
	DynContact *d_dyn_list = nullptr; uint32_t *d_dyn_keys = nullptr, *d_dyn_keys_sorted = nullptr, *d_dyn_vals = nullptr, *d_dyn_vals_sorted = nullptr;
	DynContactInfo *d_dyn_info = nullptr; float *d_dyn_sum = nullptr, *d_dyn_body = nullptr;
	unsigned int *d_dyn_counters = nullptr;
	void *d_dyn_sort_temp = nullptr; size_t dyn_sort_temp_bytes = 0;
	uint32_t dyn_cap = 0;
	bool any_dynamic() const { for (const pbdx_collider_dynamics &d : dynamics) if (d.inv_mass != 0.0f) return true; return false; }
	void free_dynamic_work()
	{
		void *ptrs[] = { d_dyn_list, d_dyn_keys, d_dyn_keys_sorted, d_dyn_vals, d_dyn_vals_sorted, d_dyn_info, d_dyn_sum, d_dyn_sort_temp };
		for (void *q : ptrs) if (q) (void)hipFree(q);
		d_dyn_list = nullptr; d_dyn_keys = d_dyn_keys_sorted = d_dyn_vals = d_dyn_vals_sorted = nullptr; d_dyn_info = nullptr; d_dyn_sum = nullptr; d_dyn_sort_temp = nullptr;
		dyn_cap = 0; dyn_sort_temp_bytes = 0;
	}

	// contacts between deformable solids (pbdx_tetcontact.h)
	struct DevBvh { uint32_t *lst = nullptr; int32_t *nodes = nullptr; P4 *hulls = nullptr; uint32_t num_nodes = 0; uint32_t *flat = nullptr; P4 *gathered = nullptr; float *soa = nullptr; };
	struct DevTetCollider { uint32_t *tets = nullptr; DevBvh points, tet_bvh, tet_bvh0; uint32_t max_nodes = 0; };
	std::vector<DevTetCollider> tet_dev;
	std::vector<TetColliderView> tet_views;       // host copy of the device views
	TetColliderView *d_tet_views = nullptr;
	float *d_tet_aabb = nullptr;
	TetContact *d_tet_contacts = nullptr;
	uint32_t *d_tet_counters = nullptr;
	uint32_t *d_tet_big = nullptr; uint32_t tet_big_count = 0;   // nodes with long chains (tet_hull_kernel2)
	uint32_t *d_tet_big_slices = nullptr, *d_tet_big_r2 = nullptr; uint32_t tet_big_slices = 0;
	TetWork tet_work = {};
	void *tet_work_alloc[24] = {};
	size_t tet_sort_temp_bytes = 0;      // scratch of the (particle, slot) pair sort of the contact velocity impulses (tet_work_alloc[21])
	pbdx_collision_range *d_ranges = nullptr;     // device copy of `ranges` (read by the tet-contact velocity chains)
	uint32_t autotune_cache_hits = 0;     // schedule decisions taken from the process-wide cache (autotune_schedule)
	bool measuring_iter_form = false;    // autotune_schedule: time the one-launch-per-iteration form of the sweeps
	int tet_force_impulses = 0;          // PBDX_OPT_TET_FORCE_IMPULSES
	uint32_t tet_impulses_last = 0; uint64_t tet_impulses_total = 0;     // contacts with a non-zero velocity impulse: last detection / since the colliders were set
	int tet_serial = 0;                            // PBDX_OPT_TET_CONTACTS_SERIAL
	uint32_t tet_grown = 0;                        // times the detection's scratch was enlarged
	uint32_t tet_num_colliders = 0;
	// developer aid (PBDX_TET_PROFILE=1, hipGraph off): wall time per kernel of the contact path, printed when the solver is destroyed
	bool tet_profile = getenv("PBDX_TET_PROFILE") != nullptr;
	double tet_ms[8] = {}; uint64_t tet_launches[8] = {};
	template <class F> void tet_timed(int slot, F &&launch)
	{
		if (!tet_profile || use_graph) { launch(); return; }
		(void)hipStreamSynchronize(stream);
		const auto t0 = std::chrono::steady_clock::now();
		launch();
		(void)hipStreamSynchronize(stream);
		tet_ms[slot] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		tet_launches[slot]++;
	}
	float4 *d_rest = nullptr;                      // ParticleData::m_x0 (needed by the contacts between solids only)
	bool rest_set = false;
	bool tet_active() const { return !tet_views.empty(); }
	void free_tet_colliders()
	{
		tet_impulses_last = 0; tet_impulses_total = 0;
		for (DevTetCollider &d : tet_dev)
		{
			if (d.tets) (void)hipFree(d.tets);
			for (DevBvh *b : { &d.points, &d.tet_bvh, &d.tet_bvh0 })
			{
				if (b->lst) (void)hipFree(b->lst);
				if (b->nodes) (void)hipFree(b->nodes);
				if (b->hulls) (void)hipFree(b->hulls);
				if (b->flat) (void)hipFree(b->flat);
				if (b->gathered) (void)hipFree(b->gathered);
				if (b->soa) (void)hipFree(b->soa);
			}
		}
		tet_dev.clear(); tet_views.clear();
		if (d_tet_views) { (void)hipFree(d_tet_views); d_tet_views = nullptr; }
		if (d_tet_aabb) { (void)hipFree(d_tet_aabb); d_tet_aabb = nullptr; }
		for (void *&p : tet_work_alloc) if (p) { (void)hipFree(p); p = nullptr; }
		for (uint32_t **p : { &d_tet_big, &d_tet_big_slices, &d_tet_big_r2 }) if (*p) { (void)hipFree(*p); *p = nullptr; }
		tet_big_count = 0; tet_big_slices = 0;
		tet_work = TetWork{};
	}

	// fused plan
	FusedPlan plan;
	std::vector<DeviceSegment> dsegs;
	bool plan_instanced = false;         // the plan is one instance's, replicated
	bool plan_built = false;             // an attempt was made for the current schedule
	bool plan_ok = false;
	std::string plan_why;

	// cached graph of one substep
	// (two: with an odd number of passes per substep the folded launch ends in the OTHER position buffer, so the state
	// alternates between the two physical buffers from substep to substep and the captured launch exists in both orientations)
	hipGraph_t graph[2] = { nullptr, nullptr };
	hipGraphExec_t graph_exec[2] = { nullptr, nullptr };
	int phys = 0;                        // which physical buffer d_pos[0] (= the state, by name) currently is
	struct GraphKey { float h; uint32_t iters; int vel; float g[3]; uint64_t sched; int block; int remap; uint32_t n; int fused; int persist; } key = {};
	bool graph_valid[2] = { false, false };

	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	// PBDX_OPT_SUBSTEP_EVENTS: one event after every substep of a pbdx_solver_step call (device time per substep: median / spread)
	int substep_events = 0; std::vector<hipEvent_t> sub_events; std::vector<float> substep_ms;
	hipStream_t stream_side = nullptr;                 // forked from / joined to `stream` (also under capture): the short nodes' spheres next to the long chains
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	std::vector<hipEvent_t> prof_events;
	pbdx_step_stats stats = {};
	double type_ms[PBDX_NUM_CONSTRAINT_TYPES] = {};
	uint64_t type_launches[PBDX_NUM_CONSTRAINT_TYPES] = {};
	uint64_t type_projections[PBDX_NUM_CONSTRAINT_TYPES] = {};

	void free_plan()
	{
		for (DeviceSegment &d : dsegs)
		{
			if (d.d_tiles) (void)hipFree(d.d_tiles);
			if (d.d_chunks) (void)hipFree(d.d_chunks);
			if (d.d_idx) (void)hipFree(d.d_idx);
			if (d.d_params) (void)hipFree(d.d_params);
			if (d.d_lambda) (void)hipFree(d.d_lambda);
			if (d.d_gid) (void)hipFree(d.d_gid);
			if (d.d_trace) (void)hipFree(d.d_trace);
			if (d.d_dep_off) (void)hipFree(d.d_dep_off);
			if (d.d_dep_tile) (void)hipFree(d.d_dep_tile);
		}
		if (d_epoch) { (void)hipFree(d_epoch); d_epoch = nullptr; }
		persist_ok = false;
		dsegs.clear();
		plan = FusedPlan();
		plan_built = false;
		plan_ok = false;
		plan_instanced = false;
		plan_why.clear();
		fuse_choice = 1;
		autotune_ms[0] = autotune_ms[1] = autotune_ms[2] = 0.0f;
		persist_choice = false;
	}
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
		free_plan();
	}
	void drop_graph()
	{
		for (int i = 0; i < 2; i++)
		{
			if (graph_exec[i]) { (void)hipGraphExecDestroy(graph_exec[i]); graph_exec[i] = nullptr; }
			if (graph[i]) { (void)hipGraphDestroy(graph[i]); graph[i] = nullptr; }
			graph_valid[i] = false;
		}
	}
	void free_particles()
	{
		for (float4 **p : { &d_pos[0], &d_pos[1], &d_vel, &d_old, &d_last })
			if (*p) { (void)hipFree(*p); *p = nullptr; }
		if (d_stage) { (void)hipFree(d_stage); d_stage = nullptr; }
		if (d_hash) { (void)hipFree(d_hash); d_hash = nullptr; hash_blocks = 0; }
		for (float4 *&p : d_snap) if (p) { (void)hipFree(p); p = nullptr; }
		for (float4 *&p : d_ckpt) if (p) { (void)hipFree(p); p = nullptr; }
		ckpt_n = 0;
		if (d_rest) { (void)hipFree(d_rest); d_rest = nullptr; }
		rest_set = false;
		n = 0;
	}
	// the position buffers change roles: d_pos[0] is always the state by NAME (everything enqueued later sees the new roles;
	// kernels already enqueued carry their own pointer values)
	void swap_state() { std::swap(d_pos[0], d_pos[1]); phys ^= 1; }
	bool fused_active() const { return fuse && plan_ok && !dsegs.empty() && (fuse == 1 || fuse_choice); }
	// (contacts between solids are solved between the iterations of a substep: the sweeps of a substep cannot be one launch then)
	bool persistent_active() const { return persistent && persist_ok && persist_choice && fused_active() && !tet_active(); }
	// contacts between deformable solids are solved BETWEEN the iterations (TimeStepController.cpp:288-291): the one-launch schedule then runs one
	// launch per ITERATION (all segments of a sweep; tile-to-tile hand-offs instead of kernel boundaries), the contact solve in between
	bool persistent_iter_active() const { return persistent && persist_ok && persist_choice && fused_active() && tet_active(); }
	void free_mirror()
	{
		if (h_mirror) { (void)hipHostFree(h_mirror); h_mirror = nullptr; }
		mirror_bytes = 0;
		for (hipEvent_t &e : mirror_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
	}
	// the mirror for `bytes` of staging (+ hashes), or nullptr: option off, or no page-locked memory to be had (the copy then takes the
	// driver's path for pageable memory, as with the option off)
	char *mirror(size_t bytes)
	{
		if (!pin_host) return nullptr;
		if (bytes > mirror_bytes)
		{
			(void)hipStreamSynchronize(stream);
			if (h_mirror) { (void)hipHostFree(h_mirror); h_mirror = nullptr; }
			mirror_bytes = 0;
			if (hipHostMalloc(reinterpret_cast<void **>(&h_mirror), bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h_mirror = nullptr; return nullptr; }
			mirror_bytes = bytes;
		}
		for (hipEvent_t &e : mirror_ev)
			if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e = nullptr; return nullptr; }
		return h_mirror;
	}
};

namespace {

template <class T> int upload(T **dst, const std::vector<T> &src)
{
	*dst = nullptr;
	if (src.empty()) return PBDX_OK;
	HIPCHECK(hipMalloc(dst, src.size() * sizeof(T)));
	HIPCHECK(pbdx::copy_to_device(*dst, src.data(), src.size() * sizeof(T)));
	return PBDX_OK;
}

// Build the colour-fused plan for the current schedule + particle positions and upload it.
// A failure is not an error: the engine keeps schedule (B).
// Dependency lists and launch geometry of the persistent schedule (A').  Not being eligible is not an error.
int prepare_persistent(pbdx_solver *s)
{
	s->persist_ok = false;
	const size_t nseg = s->dsegs.size();
	const uint32_t k = s->plan.num_tiles;
	if (!nseg || nseg > kMaxPersistSegs || !k) return PBDX_OK;
	if (s->dsegs[0].block != 256 && s->dsegs[0].block != 512 && s->dsegs[0].block != 1024) return PBDX_OK;
	uint32_t mask = 0, lds = 0;
	for (const DeviceSegment &d : s->dsegs)
	{
		if (d.block != s->dsegs[0].block || d.num_tiles != k) return PBDX_OK;      // one workgroup shape for all passes
		mask |= d.type_mask;
		lds = std::max(lds, d.lds_bytes);
	}
	if ((mask & ~kMaskLight) && s->dsegs[0].block > 512) return PBDX_OK;
	{
		// the kernel keeps two flag words in static LDS next to the dynamic tile image
		const size_t cap = s->prop.maxSharedMemoryPerMultiProcessor ? s->prop.maxSharedMemoryPerMultiProcessor : s->prop.sharedMemPerBlock;
		if ((size_t)lds + 64 > cap) return PBDX_OK;
	}
	// per (segment, tile) the tiles it waits for (pbdx_plan.cpp; checked by an asynchronous-execution simulation in
	// pbdx_model_plan_check / tests/test_plan.py)
	PersistentDeps deps;
	build_persistent_deps(s->plan, deps);
	for (size_t si = 0; si < nseg; si++)
	{
		std::vector<uint32_t> lst = deps.tile[si];
		if (lst.empty()) lst.push_back(0);
		s->dsegs[si].dep_count = (uint32_t)deps.tile[si].size();
		int r = upload(&s->dsegs[si].d_dep_off, deps.off[si]);
		if (!r) r = upload(&s->dsegs[si].d_dep_tile, lst);
		if (r) return r;
	}
	HIPCHECK(hipMalloc(&s->d_epoch, ((size_t)k + 2) * sizeof(uint32_t)));      // + arrivals, decision
	s->persist_block = s->dsegs[0].block;
	s->persist_lds = lds;
	// particle ids resident in LDS (LdsIds): room for the largest halo id list and the largest boundary id list behind the largest tile image
	s->ids_halo_cap = s->ids_bnd_cap = 0;
	if (!getenv("PBDX_NO_LDS_IDS"))
	{
		uint32_t hmax = 0, bmax = 0;
		for (const FusedSegment &seg : s->plan.segs)
			for (const FusedTile &t : seg.tiles)
			{
				hmax = std::max(hmax, t.n_local - (t.n_owned & ~63u));
				bmax = std::max(bmax, t.n_owned - (t.wb_begin & ~3u));
			}
		const uint32_t hcap = (hmax + 255u) & ~255u, bcap = (bmax + 255u) & ~255u;      // (whole 1 KiB wave copies)
		const size_t cap = s->prop.maxSharedMemoryPerMultiProcessor ? s->prop.maxSharedMemoryPerMultiProcessor : s->prop.sharedMemPerBlock;
		const uint32_t base16 = (lds + 15u) / 16u;
		if ((size_t)base16 * 16u + (size_t)(hcap + bcap) * 4u + 64u <= cap / s->persist_wgs_per_cu)
		{
			s->ids_halo_off16 = base16; s->ids_bnd_off16 = base16 + hcap / 4u;
			s->ids_halo_cap = hcap; s->ids_bnd_cap = bcap;
			s->persist_lds = (base16 + (hcap + bcap) / 4u) * 16u;
			lds = s->persist_lds;
		}
	}
	s->persist_kernel = pick_persistent_kernel(mask, s->persist_block);
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(s->persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	// every workgroup must be resident: at most as many as the occupancy query admits (one per tile otherwise)
	int per_cu = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(s->persist_kernel), s->persist_block, lds) != hipSuccess || per_cu < 1)
	{
		(void)hipGetLastError();
		return PBDX_OK;
	}
	if ((uint32_t)per_cu < s->persist_wgs_per_cu) return PBDX_OK;
	// `wgs_per_cu` workgroups per CU at most (default one: leaves the margin the occupancy API lacks; the handshake of the
	// launch is what actually guards residency)
	s->persist_grid = std::min<uint32_t>(k, (uint32_t)std::max(1, s->prop.multiProcessorCount) * s->persist_wgs_per_cu);
	// Fewer workgroups than tiles: the walk (persistent_kernel).  The asynchronous-execution model of the lists is run for THIS grid before the schedule is
	// used (the plan check covers synthetic grids of two and three tiles per workgroup only; here the real one, uneven tile counts per workgroup
	// included).  A fraction of a second on the host for the 512-tile block, once per plan; a failure keeps the schedule off (one launch per segment) and says why.
	if (s->persist_grid < k && !getenv("PBDX_NO_WALK_CHECK"))
	{
		std::string why;
		if (!check_persistent_deps(s->plan, deps, 2u * (uint32_t)nseg + 1u, why, true, s->persist_grid))
		{
			s->plan_why = "persistent schedule off: the asynchronous-execution check of the walk failed for " + std::to_string(s->persist_grid) + " workgroups: " + why;
			return PBDX_OK;
		}
	}
	s->persist_ok = true;
	return PBDX_OK;
}

// Device image of a segment's index stream (run_typed PACKED).  The plan (pbdx_plan.cpp) keeps indices and parameters in separate streams; the image
// interleaves them where a slot streams exactly one dword besides its indices: steps of a compact one-plane type (packed_plain_type) and dictionary-form
// steps (the table offset, widened to a dword) become arrays of 8-byte (two bodies) or 12-byte (four bodies) records in slot order; every other step's
// indices are copied as they are.  steps[si] = byte offset of step si in the image, bytes per slot there, packed or not.
struct StepImage { uint32_t boff, rec_bytes; bool packed; };
void build_idx_image(const FusedSegment &seg, const TypeView *views, int block, bool pack_plain, std::vector<uint8_t> &img, std::vector<StepImage> &steps)
{
	img.clear();
	steps.assign(seg.steps.size(), StepImage{ 0u, 0u, false });
	for (size_t si = 0; si < seg.steps.size(); si++)
	{
		const FusedStep &st = seg.steps[si];
		const TypeInfo *ti = type_info((int)st.type);
		const bool compact = views[st.type].compact != 0;
		const uint32_t iw = ti->num_bodies == 2 ? 2u : 4u, ib = iw * 2u;
		const uint32_t np = (uint32_t)num_planes((int)st.type, compact);
		const bool quad = quad_strain_step((int)st.type, st.count, (uint32_t)block) || is_quad_type((int)st.type);
		const bool packed = !quad && ((st.dict && PBDX_PACK_DICT) || (!st.dict && pack_plain && packed_plain_type((int)st.type) && compact && np == 1u));
		const uint32_t rb = packed ? ib + 4u : ib;
		const size_t off = img.size();
		img.resize(off + (((size_t)st.count * rb + 15u) & ~(size_t)15u), 0);
		steps[si] = { (uint32_t)off, rb, packed };
		const uint16_t *src = &seg.idx[st.idx_off];
		if (!packed) { memcpy(&img[off], src, (size_t)st.count * ib); continue; }
		const uint16_t *entry = st.dict ? reinterpret_cast<const uint16_t *>(&seg.params[st.par_off]) : nullptr;
		for (uint32_t q = 0; q < st.count; q++)
		{
			uint8_t *rec = &img[off + (size_t)q * rb];
			memcpy(rec, src + (size_t)q * iw, ib);
			uint32_t word;
			if (entry) word = entry[q];
			else memcpy(&word, &seg.params[st.par_off + param_float_index(seg.vector_params, 1u, 0u, q)], 4);
			memcpy(rec + ib, &word, 4);
		}
	}
}

int ensure_plan(pbdx_solver *s)
{
	if (!s->fuse || s->plan_built) return PBDX_OK;
	s->plan_built = true;
	s->plan_ok = false;
	if (s->order.empty() || s->n == 0 || s->h_x.size() != (size_t)3 * s->n) { s->plan_why = "no schedule / particles"; return PBDX_OK; }
	const auto t_plan0 = std::chrono::steady_clock::now();
	const bool plan_verbose = getenv("PBDX_PLAN_VERBOSE") != nullptr;
	auto lap_plan = [&](const char *what) { if (plan_verbose) fprintf(stderr, "[engine] %-36s %7.3f s since ensure_plan\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_plan0).count()); };
	std::vector<PlanBatch> pbs;
	uint32_t colour = 0;
	for (size_t oi = 0; oi < s->order.size(); oi++)
	{
		const Batch &b = s->batches[s->order[oi]];
		if (oi && b.group != s->batches[s->order[oi - 1]].group) colour++;
		pbs.push_back({ b.type, colour, b.count, b.h_idx.data(), b.h_params.data() });
	}
	PlanOptions opt;
	opt.tile_particles = s->tile_particles;
	{
		const size_t lds = s->prop.maxSharedMemoryPerMultiProcessor ? s->prop.maxSharedMemoryPerMultiProcessor : s->prop.sharedMemPerBlock;
		const uint32_t wgs = s->persistent ? s->persist_wgs_per_cu : 1u;
		opt.max_local = std::min<uint32_t>(s->lds_particles, (uint32_t)(lds / 16 / wgs) - kMaxTileChunks - (wgs > 1 ? 8u : 0u));
		opt.max_tile_steps = kMaxTileSteps;
	}
	opt.num_cus = (uint32_t)std::max(1, s->prop.multiProcessorCount) * (s->persistent ? s->persist_wgs_per_cu : 1u);
	opt.max_segment_colours = s->max_segment_colours;
	if (s->persistent && (uint64_t)s->n > (uint64_t)opt.num_cus * 1024u)
	{
		// (large tiles only: with the 512-particle tiles of small scenes a pass is a few short colour steps and the
		// hand-off is what an extra pass costs -- measured 7 % slower on a 300x300 cloth and on the FEM bar)
		// planned for the one-launch schedule: a pass boundary costs a tile-to-tile hand-off (publish, poll, halo fill),
		// not a kernel boundary plus a full fill -- shorter segments with less halo redundancy pay off
		// (1 M cloth: 3 passes per sweep at redundancy 1.23 instead of 2 at 1.40, measured 2.6 % faster)
		opt.launch_cost_ns = 1500.0;
		opt.owned_stay_in_lds = true;
	}
	{
		// form of the parameter streams (pbdx_plan.h): follows the workgroup size, which is only final once the plan exists -- predicted here by
		// the rules applied below (forced size; heavy types and small scenes run 512 threads at most), converted afterwards where the prediction missed
		uint32_t mask = 0;
		for (const PlanBatch &pb : pbs) mask |= 1u << pb.type;
		const bool small_scene = (uint64_t)s->n <= (uint64_t)std::max(1, s->prop.multiProcessorCount) * 512u;
		opt.vector_params = (s->fuse_block == 256 || s->fuse_block == 512 || s->fuse_block == 1024) ? vector_params_for_block(s->fuse_block)
			: ((mask & ~kMaskLight) != 0 || small_scene);
		// dictionary form of the bending records (pbdx_plan.h dict_type; PBDX_NO_DICT: developer A/B switch).  Worth most where the sweep is
		// bandwidth-bound (1 M cloth -15 %, configs[3] block -17 %), still 4-8 % on 200x200 ... 360x360 cloths, nothing at 100x100 (profiles/r03z_*)
		opt.dict_params = !getenv("PBDX_NO_DICT") && opt.max_local > 2u * kDictTableF4;
		if (opt.dict_params) { opt.sizing_local = opt.max_local; opt.max_local -= kDictTableF4; }
		// bank-aware order of the slots inside a colour step (pbdx_plan.h lds_bank_model; PBDX_NO_BANK_ORDER: developer A/B switch)
		opt.bank_aware = !getenv("PBDX_NO_BANK_ORDER");
	}
	bool planned = false;
	if (s->inst_count > 1 && (uint64_t)s->inst_particles * s->inst_count == s->n)
	{
		// K congruent instances: plan one, replicate (falls through to the plan of the whole if the hint does not hold)
		std::string why_inst;
		planned = build_instanced_plan(s->inst_particles, s->inst_count, s->h_x.data(), pbs, opt, s->plan, why_inst);
		s->plan_instanced = planned;
	}
	if (!planned && !build_fused_plan(s->n, s->h_x.data(), pbs, opt, s->plan, s->plan_why))
		return PBDX_OK;
	lap_plan("plan built");
	// workgroup size per segment: enough threads to cover the largest colour step of a tile once, at most 1024
	std::vector<int> blocks;
	uint32_t all_mask = 0;
	for (const FusedSegment &seg : s->plan.segs) all_mask |= seg.type_mask;
	for (const FusedSegment &seg : s->plan.segs)
	{
		int block = s->fuse_block;
		if (block != 256 && block != 512 && block != 1024)
		{
			uint32_t widest = 0;
			for (const FusedStep &st : seg.steps) widest = std::max(widest, st.count * (is_quad_type((int)st.type) ? 4u : 1u));      // in lanes
			block = widest > 512 ? 1024 : widest > 256 ? 512 : 256;
			// small scenes (fewer than 512 particles per CU: every colour step is latency-bound, pbdx_plan.cpp): 8 wavefronts meet at the colour
			// barriers sooner than 16 and a step wider than 512 slots simply takes a second chunk -- measured 3 - 14 % faster than 1 024 threads
			// (profiles/r03b_c3_tile_sweep.log, r03c_small_scene_tile_sweep.log: bar XPBD distance + volume 0.84 -> 0.74 ms, 100x100 / 200x200 cloth -5 %)
			if (block > 512 && (uint64_t)s->n <= (uint64_t)std::max(1, s->prop.multiProcessorCount) * 512u) block = 512;
		}
		if ((seg.type_mask & ~kMaskLight) && block > 512) block = 512;   // heavy types need > 128 VGPRs
		if (block == 768) block = 512;
		blocks.push_back(block);
	}
	if (s->persistent)
	{
		// the persistent schedule runs every segment in one launch: one workgroup shape for all of them
		int widest = *std::max_element(blocks.begin(), blocks.end());
		if ((all_mask & ~kMaskLight) && widest > 512) widest = 512;
		if (widest == 768) widest = 1024;
		std::fill(blocks.begin(), blocks.end(), widest);
	}
	// per-segment device image (pushed first, so that free_plan() releases a partially built one)
	for (size_t segi = 0; segi < s->plan.segs.size(); segi++)
	{
		relayout_params(s->plan.segs[segi], s->plan.views, vector_params_for_block(blocks[segi]));      // (a no-op unless the prediction above missed)
		const FusedSegment &seg = s->plan.segs[segi];
		s->dsegs.emplace_back();
		DeviceSegment &d = s->dsegs.back();
		const int block = blocks[segi];
		d.block = block;
		// the index stream as the kernels read it: packed records where a slot streams one dword besides its indices (build_idx_image)
		std::vector<uint8_t> idx_img;
		std::vector<StepImage> step_img;
		build_idx_image(seg, s->plan.views, block, s->pack_plain, idx_img, step_img);
		// expand every tile's steps into workgroup-wide chunks (FusedChunk) for this workgroup size
		std::vector<FusedTile> tiles = seg.tiles;
		std::vector<FusedChunk> chunks;
		bool too_many = false;
		for (FusedTile &t : tiles)
		{
			t.chunk_begin = (uint32_t)chunks.size();
			for (uint32_t si = t.step_begin; si < t.step_end; si++)
			{
				const FusedStep &st = seg.steps[si];
				const TypeInfo *ti = type_info((int)st.type);
				const uint32_t nplanes = (uint32_t)num_planes((int)st.type, s->plan.views[st.type].compact != 0);
				const uint32_t slot_idx_bytes = ti->num_bodies == 2 ? 4u : 8u;
				// slots per workgroup-wide chunk: one per lane, or one per QUAD of lanes (pbdx_quad.h; a multiple of 64: the parameter planes are wave-tiled)
				const bool quad_step = quad_strain_step((int)st.type, st.count, (uint32_t)block);      // this step in quad form (one chunk)
				const uint32_t cap = (quad_step || is_quad_type((int)st.type)) ? (uint32_t)block / 4u : (uint32_t)block;
				const uint32_t nchunks = (st.count + cap - 1) / cap;
				for (uint32_t k = 0; k < nchunks; k++)
				{
					const uint32_t first = k * cap;
					const uint32_t valid = std::min<uint32_t>(cap, st.count - first);
					const bool last = (k + 1 == nchunks);
					FusedChunk c = {};
					const StepImage &im = step_img[si];
					c.info = (st.dict ? kDictChunkType + st.type : quad_step ? kQuadStrainChunk : im.packed ? kPackedChunkType + st.type : st.type) | ((last && st.barrier) ? 0x40u : 0u) | (last ? 0x80u : 0u) | (valid << 8);
					c.idx_boff = im.boff + first * im.rec_bytes;
					(void)slot_idx_bytes;
					c.par_boff = st.dict ? st.par_off * 4u + first * 2u : (st.par_off + (first / 64u) * nplanes * 64u) * 4u;       // (dictionary form: one uint16 per slot)
					c.lam_boff = (st.lam_off + first) * 4u;
					chunks.push_back(c);
				}
			}
			t.chunk_end = (uint32_t)chunks.size();
			if (t.chunk_end - t.chunk_begin > kMaxTileChunks) too_many = true;
			// chunks left in each run of equal type (a tile never has more than kMaxTileChunks < 8192)
			uint32_t left = 0;
			for (uint32_t ci = t.chunk_end; ci-- > t.chunk_begin;)
			{
				const bool same = (ci + 1 < t.chunk_end) && ((chunks[ci + 1].info & 0x3fu) == (chunks[ci].info & 0x3fu));
				left = same ? left + 1 : 1;
				chunks[ci].info |= std::min(left, 8191u) << 19;
			}
			// ... and where the chunk fetched while chunk ci is projected sits in the streams: `ring depth` positions ahead, the run's last chunk beyond its end
			for (uint32_t ci = t.chunk_begin; ci < t.chunk_end; ci++)
			{
				const uint32_t ctype = chunks[ci].info & 0x3fu, run_left = chunks[ci].info >> 19;
				const int real_type = ctype >= kPackedChunkType ? (int)(ctype - kPackedChunkType) : ctype >= kDictChunkType ? (int)(ctype - kDictChunkType) : ctype == kQuadStrainChunk ? (int)PBDX_STRAIN_TET : (int)ctype;
				const uint32_t depth = (uint32_t)ring_depth(real_type);
				const FusedChunk &f = chunks[ci + std::min(depth, run_left - 1u)];
				chunks[ci].f_idx_boff = f.idx_boff; chunks[ci].f_par_boff = f.par_boff; chunks[ci].f_lam_boff = f.lam_boff; chunks[ci].pad = 0u;
			}
		}
		if (too_many)
		{
			s->free_plan();
			s->plan_built = true;
			s->plan_why = "a tile needs more chunk descriptors than fit the LDS header: use smaller segments or larger workgroups";
			return PBDX_OK;
		}
		std::vector<TileDev> tiles_dev(tiles.size());
		for (size_t ti = 0; ti < tiles.size(); ti++) { memset(&tiles_dev[ti], 0, sizeof(TileDev)); tiles_dev[ti].t = tiles[ti]; }
		lap_plan("segment image (records, chunks)");
		int r = upload(&d.d_tiles, tiles_dev);
		if (!r) r = upload(&d.d_chunks, chunks);
		if (!r) r = upload(reinterpret_cast<uint8_t **>(&d.d_idx), idx_img);
		if (!r) r = upload(&d.d_params, seg.params);
		if (!r) r = upload(&d.d_gid, seg.gid);
		if (!r && seg.lam_count)
		{
			hipError_t e = hipMalloc(&d.d_lambda, (size_t)seg.lam_count * sizeof(float));
			if (e == hipSuccess) e = hipMemset(d.d_lambda, 0, (size_t)seg.lam_count * sizeof(float));
			if (e != hipSuccess) { set_error("lambda stream allocation failed: %s", hipGetErrorString(e)); r = PBDX_ERR_HIP; }
		}
		if (r)
		{
			s->free_plan();
			s->plan_built = true;
			return r;
		}
		d.idx_bytes = (uint32_t)idx_img.size();
		d.params_bytes = (uint32_t)(seg.params.size() * sizeof(float));
		d.lambda_bytes = (uint32_t)((size_t)seg.lam_count * sizeof(float));
		d.gid_count = (uint32_t)seg.gid.size(); d.chunk_count = (uint32_t)chunks.size(); d.n_particles = s->n;
		d.num_tiles = (uint32_t)seg.tiles.size();
		d.lds_bytes = (std::max(seg.max_local, 1u) + seg.max_tab_f4) * 16u + kMaxTileChunks * 16u;       // (+ the largest dictionary table of a tile)
		d.type_mask = seg.type_mask;
		d.constraints = seg.constraints;
		for (const PlanBatch &pb : pbs)
			if (pb.colour >= seg.colour_begin && pb.colour < seg.colour_end)
				d.algorithmic_bytes += (uint64_t)pb.count * type_info(pb.type)->algorithmic_bytes;
		d.kernel = pick_fused_kernel(seg.type_mask, d.block);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(d.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)d.lds_bytes);
	}
	s->plan_ok = true;
	lap_plan("segments uploaded");
	const int rp_ = prepare_persistent(s);
	lap_plan("persistent schedule prepared");
	return rp_;
}

// Device arrays of the per-colour schedule (B), created the first time that schedule actually runs.
int ensure_device_batches(pbdx_solver *s)
{
	for (Batch &b : s->batches)
	{
		if (b.d_idx) continue;
		const TypeInfo *ti = type_info(b.type);
		const uint32_t nb = ti->num_bodies, np = ti->param_stride, count = b.count;
		// indices: uint2 for 2-body, uint4 (padded) for 3- and 4-body constraints
		const uint32_t iw = (nb == 2) ? 2 : 4;
		std::vector<uint32_t> idx((size_t)count * iw, 0);
		for (uint32_t i = 0; i < count; i++)
			for (uint32_t k = 0; k < nb; k++) idx[(size_t)i * iw + k] = b.h_idx[(size_t)i * nb + k];
		HIPCHECK(hipMalloc(&b.d_idx, idx.size() * sizeof(uint32_t)));
		HIPCHECK(pbdx::copy_to_device(b.d_idx, idx.data(), idx.size() * sizeof(uint32_t)));
		// parameters: planar streams of the layout chosen for this batch (compact / full)
		const bool compact = b.view.compact != 0;
		if (num_planes(b.type, compact) && !b.d_params)
		{
			std::vector<float> planar((size_t)num_planes(b.type, compact) * b.par_stride, 0.0f);
			for (uint32_t k = 0; k < np; k++)
			{
				if (!param_streams(b.type, compact, (int)k)) continue;
				float *dst = &planar[(size_t)param_plane(b.type, compact, (int)k) * b.par_stride];
				for (uint32_t i = 0; i < count; i++) dst[i] = b.h_params[(size_t)i * np + k];
			}
			HIPCHECK(hipMalloc(&b.d_params, planar.size() * sizeof(float)));
			HIPCHECK(pbdx::copy_to_device(b.d_params, planar.data(), planar.size() * sizeof(float)));
		}
		if (ti->xpbd && !b.d_lambda)
		{
			HIPCHECK(hipMalloc(&b.d_lambda, (size_t)count * sizeof(float)));
			HIPCHECK(hipMemset(b.d_lambda, 0, (size_t)count * sizeof(float)));
		}
	}
	return PBDX_OK;
}

// trace buffers are allocated before anything is enqueued (no hipMalloc inside a stream capture)
int ensure_trace(pbdx_solver *s)
{
	if (!s->trace) return PBDX_OK;
	for (DeviceSegment &d : s->dsegs)
		if (!d.d_trace)
		{
			HIPCHECK(hipMalloc(&d.d_trace, (size_t)d.num_tiles * kTraceStride * sizeof(unsigned long long)));
			HIPCHECK(hipMemset(d.d_trace, 0, (size_t)d.num_tiles * kTraceStride * sizeof(unsigned long long)));
		}
	return PBDX_OK;
}

int launch_batch(pbdx_solver *s, const Batch &b, float dt, int first_iter)
{
	BatchArgs a;
	a.pos = s->d_pos[0];
	a.idx = b.d_idx;
	a.lambda = b.d_lambda;
	a.par = b.d_params;
	a.par_stride = b.par_stride;
	a.view = b.view;
	a.count = b.count;
	a.dt = dt;
	a.first_iter = first_iter;
	const uint32_t bs = (uint32_t)s->block_size;
	a.num_blocks = (b.count + bs - 1) / bs;
	a.xcd_remap = s->xcd_remap;
	hipLaunchKernelGGL(project_kernel_for(b.type, b.view.compact != 0), dim3(a.num_blocks), dim3(bs), 0, s->stream, a);
	HIPCHECK(hipGetLastError());
	return PBDX_OK;
}

SegArgs seg_args(const DeviceSegment &d)
{
	SegArgs g;
	g.tiles = d.d_tiles; g.chunks = d.d_chunks; g.idx = d.d_idx; g.params = d.d_params; g.lambda = d.d_lambda; g.gid = d.d_gid;
	g.idx_bytes = d.idx_bytes; g.params_bytes = d.params_bytes; g.lambda_bytes = d.lambda_bytes;
	g.num_tiles = d.num_tiles;
#if PBDX_BOUNDS
	g.gid_count = d.gid_count; g.chunk_count = d.chunk_count; g.n_particles = d.n_particles; g.lds_f4 = d.lds_bytes / 16u - kMaxTileChunks;
#endif
	return g;
}

// (A') all `iterations` sweeps as one launch; with `fold` also the integration before and the velocity update after them
int launch_persistent(pbdx_solver *s, int src, float dt, uint32_t iterations, const FoldArgs *fold = nullptr, bool first_iteration = true)
{
	PersistArgs a;
	memset(&a, 0, sizeof(a));
	a.pos[0] = s->d_pos[0]; a.pos[1] = s->d_pos[1];
	for (size_t si = 0; si < s->dsegs.size(); si++)
	{
		a.seg[si] = seg_args(s->dsegs[si]);
		a.dep_off[si] = s->dsegs[si].d_dep_off;
		a.dep_tile[si] = s->dsegs[si].d_dep_tile;
		a.trace[si] = s->trace ? s->dsegs[si].d_trace : nullptr;
#if PBDX_BOUNDS
		a.dep_count[si] = s->dsegs[si].dep_count;
#endif
	}
	a.epoch = s->d_epoch;
	a.ctl = s->d_ctl;
	a.error = s->d_error;
	a.num_segs = (uint32_t)s->dsegs.size();
	a.passes = iterations * a.num_segs;
	a.first_iter_passes = first_iteration ? a.num_segs : 0u;
	a.num_tiles = s->plan.num_tiles;
	a.expect = s->persist_grid + (s->persistent == 3 ? 1u : 0u);     // 3 = self-test: the handshake cannot complete
	a.spin_limit = (unsigned long long)s->persist_timeout_ms * kTicksPerMs;
	a.mute_tile0 = s->persistent == 4 ? 1 : 0;                       // 4 = self-test of the timeout path
	a.start = src;
	a.dt = dt;
	if (fold) { a.folded = 1; a.fold = *fold; }
	// (one tile per workgroup: decided again in the kernel from its own grid)
	if (s->ids_halo_cap && s->persist_grid == s->plan.num_tiles)
	{
		a.ids_halo_off16 = s->ids_halo_off16; a.ids_bnd_off16 = s->ids_bnd_off16; a.ids_halo_cap = s->ids_halo_cap; a.ids_bnd_cap = s->ids_bnd_cap;
	}
	memcpy(a.views, s->plan.views, sizeof(a.views));
	HIPCHECK(hipMemsetAsync(s->d_epoch, 0, ((size_t)s->plan.num_tiles + 2) * sizeof(uint32_t), s->stream));
	hipLaunchKernelGGL(s->persist_kernel, dim3(s->persist_grid), dim3(s->persist_block), s->persist_lds, s->stream, a);
	HIPCHECK(hipGetLastError());
	return PBDX_OK;
}

// The persistent schedule can fail in one way that touches the state: a tile gives up waiting for a neighbour
// (PersistArgs::spin_limit; a GPU shared with another process, a profiler, pre-emption).  The positions are then
// garbage.  While the schedule is active every call therefore starts by saving pos / vel / old / last on the device
// (four stream-ordered copies, 64 B per particle per CALL, not per substep); a timed-out call restores them, stops
// using the schedule and is repeated with one launch per segment -- the caller sees the result of an undisturbed
// run (self-test: PBDX_OPT_PERSISTENT = 4).
int snapshot_state(pbdx_solver *s, bool positions_only)
{
	if (!s->n) return PBDX_OK;
	float4 *src[4] = { s->d_pos[0], s->d_vel, s->d_old, s->d_last };
	for (int k = 0; k < (positions_only ? 1 : 4); k++)
	{
		if (!s->d_snap[k]) HIPCHECK(hipMalloc(&s->d_snap[k], (size_t)s->n * sizeof(float4)));
		HIPCHECK(hipMemcpyAsync(s->d_snap[k], src[k], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
	}
	return PBDX_OK;
}
// after a stream synchronisation: did a persistent launch time out?  Then: state := snapshot, schedule off.
// Returns 1 if the caller has to repeat its work, 0 if nothing happened, < 0 on error (-status).
int recover_persistent(pbdx_solver *s, bool positions_only)
{
	if (!s->h_error || s->h_error[0] == 0u) return 0;
	s->h_error[0] = s->h_error[1] = s->h_error[2] = 0u;
	s->persist_ok = false;
	s->persist_timeouts++;
	s->drop_graph();
	float4 *dst[4] = { s->d_pos[0], s->d_vel, s->d_old, s->d_last };
	for (int k = 0; k < (positions_only ? 1 : 4); k++)
	{
		if (!s->d_snap[k]) { set_error("persistent schedule: a tile timed out and no snapshot exists"); return -PBDX_ERR_HIP; }
		if (hipMemcpyAsync(dst[k], s->d_snap[k], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream) != hipSuccess) return -PBDX_ERR_HIP;
	}
	if (hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream) != hipSuccess) return -PBDX_ERR_HIP;
	return 1;
}

int launch_segment(pbdx_solver *s, size_t si, int src, float dt, int first_iter)
{
	DeviceSegment &d = s->dsegs[si];
	FusedArgs a;
	a.pos_in = s->d_pos[src];
	a.pos_out = s->d_pos[src ^ 1];
	a.seg = seg_args(d);
	a.dt = dt;
	a.first_iter = first_iter;
	a.xcd_remap = s->xcd_remap;
	a.trace = s->trace ? d.d_trace : nullptr;
	memcpy(a.views, s->plan.views, sizeof(a.views));
	hipLaunchKernelGGL(d.kernel, dim3(d.num_tiles), dim3(d.block), d.lds_bytes, s->stream, a);
	HIPCHECK(hipGetLastError());
	return PBDX_OK;
}

// event bookkeeping for the profiled (eager) mode: one event right before and one right after every
// projection launch; elapsed(before, after) is charged to the launch (kernel duration on the
// engine's stream, without the host-side gap to the next launch).
// kind >= 0: constraint type of a per-colour launch; kind <= -2: fused segment (-2 - kind); -1: persistent launch
struct ProfCursor { pbdx_solver *s; size_t next = 0; std::vector<int> kinds; std::vector<uint32_t> counts; };

int prof_event(ProfCursor *pc)
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
int prof_begin(ProfCursor *pc, int kind, uint32_t count)
{
	pc->kinds.push_back(kind);
	pc->counts.push_back(count);
	return prof_event(pc);
}
int prof_end(ProfCursor *pc) { return prof_event(pc); }

// number of position-buffer flips of `iterations` sweeps
inline uint32_t sweep_flips(const pbdx_solver *s, uint32_t iterations)
{
	return s->fused_active() ? (uint32_t)(((uint64_t)iterations * s->dsegs.size()) & 1u) : 0u;
}

// `iterations` Gauss-Seidel sweeps over the schedule.  Fused: reads buffer `src`, ends in buffer
// src ^ sweep_flips().  Per-colour: in place on buffer 0 (src must be 0).
int projection_sweeps(pbdx_solver *s, float dt, uint32_t iterations, int src, ProfCursor *pc, bool with_tet_contacts = false)
{
	if (s->persistent_active() && iterations)
	{
		if (pc) { int r = prof_begin(pc, -1, 0); if (r) return r; }
		int r = launch_persistent(s, src, dt, iterations);
		if (r) return r;
		return pc ? prof_end(pc) : PBDX_OK;
	}
	// contacts between solids: TimeStepController.cpp:288-291, after the colour groups of EVERY iteration, on the buffer that holds
	// the iteration's result
	auto tet_solve = [&](int buf) -> int
	{
		if (!with_tet_contacts || !s->tet_active()) return PBDX_OK;
		if (s->tet_serial)
			hipLaunchKernelGGL(tet_contact_solve_kernel, dim3(1), dim3(64), 0, s->stream, s->d_pos[buf], (const TetContact *)s->d_tet_contacts, (const uint32_t *)s->d_tet_counters);
		else
			s->tet_timed(7, [&] { hipLaunchKernelGGL(tet_contact_solve_levels_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_pos[buf], (const TetContact *)s->d_tet_contacts, s->tet_work); });
		HIPCHECK(hipGetLastError());
		return PBDX_OK;
	};
	if (s->persistent_iter_active() && (with_tet_contacts || s->measuring_iter_form))
	{
		for (uint32_t it = 0; it < iterations; it++)
		{
			if (pc) { int r = prof_begin(pc, -1, 0); if (r) return r; }
			int r = launch_persistent(s, src, dt, 1, nullptr, it == 0);
			if (r) return r;
			if (pc) { r = prof_end(pc); if (r) return r; }
			src ^= (int)(s->dsegs.size() & 1u);
			r = tet_solve(src);
			if (r) return r;
		}
		return PBDX_OK;
	}
	if (s->fused_active())
	{
		for (uint32_t it = 0; it < iterations; it++)
		{
			for (size_t si = 0; si < s->dsegs.size(); si++)
			{
				if (pc) { int r = prof_begin(pc, -2 - (int)si, (uint32_t)s->dsegs[si].constraints); if (r) return r; }
				int r = launch_segment(s, si, src, dt, it == 0);
				if (r) return r;
				if (pc) { r = prof_end(pc); if (r) return r; }
				src ^= 1;
			}
			int r = tet_solve(src);
			if (r) return r;
		}
	}
	else
	{
		for (uint32_t it = 0; it < iterations; it++)
		{
			for (uint32_t bi : s->order)
			{
				const Batch &b = s->batches[bi];
				if (pc) { int r = prof_begin(pc, b.type, b.count); if (r) return r; }
				int r = launch_batch(s, b, dt, it == 0);
				if (r) return r;
				if (pc) { r = prof_end(pc); if (r) return r; }
			}
			int r = tet_solve(0);
			if (r) return r;
		}
	}
	return PBDX_OK;
}

// Schedule selection by measurement.  (1) PBDX_OPT_FUSE = 2: the fused schedule executes halo constraints
// redundantly, which pays for streaming-bound types (distance, bending, volume: always faster, no measurement) but
// not always for compute-heavy ones (FEM, strain, shape matching: ~1000 VALU instructions per projection); for those
// the per-colour schedule is a candidate.  (2) PBDX_OPT_PERSISTENT = 1: the one-launch form of the fused schedule
// wins on large scenes (no kernel boundaries) and loses where a pass is shorter than a tile-to-tile hand-off.
// Every candidate is timed once on scratch copies of the positions and the fastest is kept.  All schedules are
// bit-identical, so the choice can never change a result.
int autotune_schedule(pbdx_solver *s)
{
	s->autotuned_for_tet = s->tet_active();      // the measurement below times the form this decides (per substep / per iteration)
	s->fuse_choice = 1;
	s->persist_choice = s->persistent >= 2;      // 2 = always, 3 = self-test (always try, and be refused)
	s->autotune_ms[0] = s->autotune_ms[1] = s->autotune_ms[2] = 0.0f;
	if (!s->plan_ok || s->dsegs.empty()) return PBDX_OK;
	uint32_t mask = 0;
	for (const DeviceSegment &d : s->dsegs) mask |= d.type_mask;
	const bool try_percolour = s->fuse == 2 && (mask & ~kMaskLight) != 0;
	const bool try_persistent = s->persistent == 1 && s->persist_ok;
	if (!try_percolour && !try_persistent) return PBDX_OK;
	// The decision depends on the shape of the plan, not on its contents: a plan of the same shape on the same device (a topology edit that leaves the
	// sizes alone, the engines of an ensemble with congruent blocks, a second solver on the same scene) takes the decision measured before instead of
	// 60-80 ms of measurement (PBDX_NO_AUTOTUNE_CACHE: always measure).  The cache is process-wide.
	uint64_t sig = 0xcbf29ce484222325ull;
	{
		auto mixin = [&](uint64_t v) { sig ^= v + 0x9e3779b97f4a7c15ull + (sig << 6) + (sig >> 2); sig *= 0x100000001b3ull; };
		mixin((uint64_t)s->device); mixin(s->n); mixin(s->plan.num_constraints); mixin(s->plan.num_tiles); mixin(s->dsegs.size()); mixin(mask);
		mixin(s->persist_ok ? s->persist_grid : 0u); mixin((uint64_t)s->persist_block); mixin(s->tet_active() ? 1u : 0u); mixin((uint64_t)s->fuse); mixin((uint64_t)s->persistent);
		for (const DeviceSegment &d : s->dsegs) { mixin(d.constraints); mixin(d.idx_bytes); mixin(d.params_bytes); mixin(d.lds_bytes); mixin((uint64_t)d.block); }
	}
	static std::mutex cache_mu;
	static std::unordered_map<uint64_t, std::array<float, 5>> cache;      // signature -> ms[3], fuse_choice, persist_choice
	if (!getenv("PBDX_NO_AUTOTUNE_CACHE"))
	{
		std::lock_guard<std::mutex> lk(cache_mu);
		auto it = cache.find(sig);
		if (it != cache.end())
		{
			for (int i = 0; i < 3; i++) s->autotune_ms[i] = it->second[i];
			s->fuse_choice = (int)it->second[3]; s->persist_choice = it->second[4] != 0.0f;
			s->autotune_rounds = 0; s->autotune_spent_ms = 0.0f; s->autotune_cache_hits++;
			return PBDX_OK;
		}
	}
	int r = try_percolour ? ensure_device_batches(s) : PBDX_OK;
	if (r) return r;
	float4 *scratch[2] = { nullptr, nullptr }, *keep[2] = { s->d_pos[0], s->d_pos[1] };
	HIPCHECK(hipMalloc(&scratch[0], (size_t)s->n * sizeof(float4)));
	if (hipMalloc(&scratch[1], (size_t)s->n * sizeof(float4)) != hipSuccess) { (void)hipFree(scratch[0]); set_error("autotune: out of memory"); return PBDX_ERR_HIP; }
	float ms[3] = { 0.0f, 0.0f, 0.0f };      // per-colour, fused (one launch per segment), fused persistent
	const float dt = 0.005f;
	const uint32_t sweeps = 12;
	bool use[3] = { try_percolour, true, try_persistent };
	s->d_pos[0] = scratch[0]; s->d_pos[1] = scratch[1];
	// The candidates are measured in turns (round 0 = warm-up, then the minimum over the following rounds): the clock of the
	// first milliseconds after an idle period would otherwise favour whichever candidate runs first.  One measured
	// run is long (12 sweeps), so that the fixed cost of an eager launch sequence (copies, memsets) does not decide
	// between one launch and twenty-four.
	// How many rounds: the device comes out of the planning phase (0.04-0.4 s of host work) at a low clock and needs tens of milliseconds of
	// work to reach the one it holds afterwards -- the substeps of the 1 M cloth take 0.669 ms right after two measured rounds, 0.644 twenty
	// steps later, 0.622 from about the fiftieth on (profiles/r04l_*) -- and a 2 % decision taken on the ramp is taken on noise.  Rounds are
	// repeated until the fastest run of a round stops improving on the previous round's (three rounds in a row within 0.2 %), within a budget of
	// kAutotuneBudgetMs of measured time and kAutotuneMaxRounds rounds.
	constexpr int kAutotuneMaxRounds = 64;
	constexpr float kAutotuneBudgetMs = 80.0f;
	float spent_ms = 0.0f, prev_round_best = 0.0f;
	int settled = 0, rounds_done = 0;
	for (int round = 0; round < kAutotuneMaxRounds && !r; round++)
	{
		float round_best = 0.0f;
		for (int cand = 0; cand < 3 && !r; cand++)
		{
			if (!use[cand]) continue;
			s->fuse_choice = cand == 0 ? 0 : 1;
			s->persist_choice = cand == 2;
			if (hipMemcpyAsync(scratch[0], keep[0], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream) != hipSuccess) { r = PBDX_ERR_HIP; break; }
			if (cand == 2 && hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream) != hipSuccess) { r = PBDX_ERR_HIP; break; }
			(void)hipEventRecord(s->ev_start, s->stream);
			s->measuring_iter_form = cand == 2 && s->tet_active();      // with contacts between the iterations the one-launch form is one launch per ITERATION
			r = projection_sweeps(s, dt, round == 0 ? 2u : sweeps, 0, nullptr);
			s->measuring_iter_form = false;
			(void)hipEventRecord(s->ev_stop, s->stream);
			if (hipStreamSynchronize(s->stream) != hipSuccess) r = PBDX_ERR_HIP;
			float t = 0.0f;
			if (!r) (void)hipEventElapsedTime(&t, s->ev_start, s->ev_stop);
			if (round > 0) ms[cand] = ms[cand] == 0.0f ? t : std::min(ms[cand], t);
			if (round > 0 && t > 0.0f) { spent_ms += t; round_best = round_best == 0.0f ? t : std::min(round_best, t); }
			if (cand == 2 && (s->h_error[0] || s->h_error[1]))
			{
				// refused or timed out on the scratch copy: not a candidate (nothing of the real state was touched)
				s->h_error[0] = s->h_error[1] = s->h_error[2] = 0u;
				(void)hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream);
				s->persist_ok = false;
				s->persist_refusals++;
				use[2] = false;
				ms[2] = 0.0f;
			}
		}
		if (round > 0) rounds_done++;
		if (round >= 2)
		{
			settled = (prev_round_best > 0.0f && round_best > 0.998f * prev_round_best) ? settled + 1 : 0;
			if (settled >= 3 || spent_ms > kAutotuneBudgetMs) break;
		}
		if (round > 0) prev_round_best = round_best;
	}
	s->autotune_rounds = rounds_done; s->autotune_spent_ms = spent_ms;
	s->d_pos[0] = keep[0]; s->d_pos[1] = keep[1];
	(void)hipFree(scratch[0]); (void)hipFree(scratch[1]);
	s->fuse_choice = 1;
	s->persist_choice = s->persistent >= 2;
	if (r) return r;
	for (int i = 0; i < 3; i++) s->autotune_ms[i] = ms[i];
	// The short run on scratch positions understates the one-launch form (measured 1-3 % ahead here where whole
	// substeps are 5-8 % faster: its launch-time costs are spread over 12 sweeps, not over a substep loop), so it
	// is only rejected where it is clearly behind.
	// (one launch per ITERATION, scenes with contacts between deformable solids: nothing is amortised over a substep there -- every launch pays its
	// residency handshake and the reset of the pass counters, about what the kernel boundaries it replaces cost; it has to win the measurement clearly)
	if (ms[2] > 0.0f && ms[2] < (s->tet_active() ? 0.97f : 1.02f) * ms[1]) s->persist_choice = true;
	const float fused_best = s->persist_choice ? ms[2] : ms[1];
	if (ms[0] > 0.0f && ms[0] < fused_best) { s->fuse_choice = 0; s->persist_choice = false; }
	if (s->persistent >= 2 && s->fuse_choice) s->persist_choice = true;
	{
		std::lock_guard<std::mutex> lk(cache_mu);
		cache[sig] = { ms[0], ms[1], ms[2], (float)s->fuse_choice, s->persist_choice ? 1.0f : 0.0f };
	}
	return PBDX_OK;
}

// the control block is handed to the particle kernels only while the persistent schedule runs (captured graphs are
// rebuilt when that changes: GraphKey::persist)
inline uint32_t *ctl_of(pbdx_solver *s, ProfCursor *) { return (s->persistent_active() || s->persistent_iter_active()) ? s->d_ctl : nullptr; }

// second half of a substep: the sweeps and the velocity update
int enqueue_substep_tail(pbdx_solver *s, float hs, float inv_h, uint32_t iters, int vel, ProfCursor *pc)
{
	const uint32_t bs = 256;
	const uint32_t nb = (s->n + bs - 1) / bs;
	const int start = (int)sweep_flips(s, iters);
	uint32_t *ctl = ctl_of(s, pc);
	int r = projection_sweeps(s, hs, iters, start, pc, true);
	if (r) return r;
	if (s->n)
	{
		hipLaunchKernelGGL(velocity_kernel, dim3(nb), dim3(bs), 0, s->stream, s->d_pos[0], s->d_vel, s->d_old, s->d_last, s->n, inv_h, vel != 0, ctl);
		HIPCHECK(hipGetLastError());
	}
	return PBDX_OK;
}

int enqueue_substep(pbdx_solver *s, float hs, float inv_h, uint32_t iters, int vel, const float g[3], ProfCursor *pc)
{
	const uint32_t bs = 256;
	const uint32_t nb = (s->n + bs - 1) / bs;
	// the state lives in buffer 0 between substeps; integrate writes the buffer from which an odd
	// number of fused launches ends in buffer 0 again
	const int start = (int)sweep_flips(s, iters);
	s->last_folded = false;
	s->last_flips = false;
	if (s->persistent_active() && iters && s->n)
	{
		// The whole substep as ONE launch: pass 0 integrates while it stages, the last pass updates the velocities.
		// Pass 0 reads the state buffer and writes the other one, so no tile can overwrite state a neighbour has not
		// integrated yet.  With an even number of passes the last pass ends in the state buffer again; with an odd number it
		// ends in the other one, and the two buffers change roles after the substep (the CALLER swaps them once the launch is
		// enqueued: swap_state; a captured launch exists in both orientations).
		s->last_flips = start != 0;
		FoldArgs f;
		f.vel = s->d_vel; f.old = s->d_old; f.last = s->d_last;
		f.state_bytes = s->n * 16u;
		f.h = hs; f.gx = g[0]; f.gy = g[1]; f.gz = g[2]; f.inv_h = inv_h; f.second_order = vel != 0;
		f.ghx = g[0] * hs; f.ghy = g[1] * hs; f.ghz = g[2] * hs;      // (host code is compiled without contraction, like the device code)
		s->last_folded = true;
		if (pc) { int r = prof_begin(pc, -1, 0); if (r) return r; }
		int r = launch_persistent(s, 0, hs, iters, &f);
		if (r) return r;
		return pc ? prof_end(pc) : PBDX_OK;
	}
	if (s->n)
	{
		hipLaunchKernelGGL(integrate_kernel, dim3(nb), dim3(bs), 0, s->stream, s->d_pos[0], s->d_pos[start], s->d_vel, s->d_old, s->d_last, s->n, hs, g[0], g[1], g[2],
			(const uint32_t *)ctl_of(s, pc));
		HIPCHECK(hipGetLastError());
	}
	return enqueue_substep_tail(s, hs, inv_h, iters, vel, pc);
}

// collision detection + velocity constraint projection of the contacts, once per step after the
// substeps (TimeStepController.cpp:216-223)
// DistanceFieldCollisionDetection::collisionDetection for the solid-solid pairs: bounding spheres, boxes, detection (pbdx_tetcontact.h).
// The new contact list is used by the position solves of the NEXT step.
// (re)allocates the scratch of the detection; the contact list is preserved across a growth of anything else
int alloc_tet_work_impl(pbdx_solver *s, uint64_t nodes, uint32_t contacts);
// a failed (re)allocation leaves nothing half-built behind: the colliders are dropped, the step that needed them reports the error
int alloc_tet_work(pbdx_solver *s, uint64_t nodes, uint32_t contacts)
{
	const int r = alloc_tet_work_impl(s, nodes, contacts);
	if (r) s->free_tet_colliders();
	return r;
}
int alloc_tet_work_impl(pbdx_solver *s, uint64_t nodes, uint32_t contacts)
{
	TetWork &w = s->tet_work;
	const uint32_t n = s->tet_num_colliders;
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->drop_graph();               // the captured substep carries the solve's launch with the old buffers
	if (contacts != w.max_contacts || !s->d_tet_contacts)
	{
		TetContact *fresh = nullptr;
		HIPCHECK(hipMalloc(&fresh, (size_t)contacts * sizeof(TetContact)));
		if (s->d_tet_contacts)
		{
			HIPCHECK(hipMemcpy(fresh, s->d_tet_contacts, (size_t)std::min(contacts, w.max_contacts) * sizeof(TetContact), hipMemcpyDeviceToDevice));
			(void)hipFree(s->d_tet_contacts);
		}
		s->d_tet_contacts = fresh;
	}
	for (void *&p : s->tet_work_alloc) if (p) { (void)hipFree(p); p = nullptr; }
	w.max_contacts = contacts;
	w.node_cap = (uint32_t)nodes;
	w.front_cap = (uint32_t)std::max<uint64_t>(nodes / 4, (uint64_t)n * n);
	w.chunk_cap = (uint32_t)(nodes / 2);
	HIPCHECK(hipMalloc(&s->tet_work_alloc[15], kTrWords * sizeof(uint32_t)));
	w.trav = (uint32_t *)s->tet_work_alloc[15];
	const size_t bytes[15] = { (size_t)3 * w.front_cap * 4, (size_t)contacts * 4, (size_t)2 * n * n * 4, ((size_t)w.front_cap + 1) * 4, (size_t)w.chunk_cap * 4,
		(size_t)w.chunk_cap * 8, (size_t)w.chunk_cap * 4, (size_t)contacts * 4, ((size_t)contacts + 1) * 4, (size_t)contacts * 4, (size_t)s->n * 4,
		(size_t)w.node_cap * 16, (size_t)w.node_cap * 8, (size_t)w.node_cap * 4, (size_t)w.node_cap * 8 };
	for (int q = 0; q < 15; q++) if (bytes[q]) HIPCHECK(hipMalloc(&s->tet_work_alloc[q], bytes[q]));
	w.leaf_pairs = (uint32_t *)s->tet_work_alloc[0]; w.pair_ik = (uint32_t *)s->tet_work_alloc[2];
	w.chunk_off = (uint32_t *)s->tet_work_alloc[3]; w.chunk_pair = (uint32_t *)s->tet_work_alloc[4]; w.chunk_mask = (unsigned long long *)s->tet_work_alloc[5];
	w.chunk_base = (uint32_t *)s->tet_work_alloc[6]; w.order = (uint32_t *)s->tet_work_alloc[7]; w.level_start = (uint32_t *)s->tet_work_alloc[8];
	w.level_of = (uint32_t *)s->tet_work_alloc[9]; w.owner = (uint32_t *)s->tet_work_alloc[10];
	w.node_rec = (unsigned long long *)s->tet_work_alloc[11]; w.node_cnt = (unsigned long long *)s->tet_work_alloc[12]; w.node_child = (uint32_t *)s->tet_work_alloc[13]; w.node_off = (unsigned long long *)s->tet_work_alloc[14];
	w.counters = s->d_tet_counters;
	w.imp_list = (uint32_t *)s->tet_work_alloc[1];
	HIPCHECK(hipMalloc(&s->tet_work_alloc[16], (size_t)std::max(1u, s->n)));
	HIPCHECK(hipMemset(s->tet_work_alloc[16], 0, (size_t)std::max(1u, s->n)));
	w.imp_mark = (uint8_t *)s->tet_work_alloc[16];
	// (particle, slot) pairs of the impulse-carrying contacts and the scratch of their sort (pbdx_tetcontact_dev.h: tet_impulse_kernel)
	for (int q = 17; q <= 20; q++) HIPCHECK(hipMalloc(&s->tet_work_alloc[q], (size_t)5 * contacts * 4));
	w.imp_keys = (uint32_t *)s->tet_work_alloc[17]; w.imp_slots = (uint32_t *)s->tet_work_alloc[18];
	w.imp_keys_sorted = (uint32_t *)s->tet_work_alloc[19]; w.imp_slots_sorted = (uint32_t *)s->tet_work_alloc[20];
	s->tet_sort_temp_bytes = 0;
	{ int rs = sort_pairs_u32(nullptr, &s->tet_sort_temp_bytes, w.imp_keys, w.imp_keys_sorted, w.imp_slots, w.imp_slots_sorted, 5u * contacts, s->stream); if (rs) return rs; }
	HIPCHECK(hipMalloc(&s->tet_work_alloc[21], std::max<size_t>(s->tet_sort_temp_bytes, 16)));
	w.force_impulses = s->tet_force_impulses;
	return PBDX_OK;
}


int launch_tet_detection(pbdx_solver *s)
{
	if (!s->tet_active() || !s->n) return PBDX_OK;
	const P4 *pos = reinterpret_cast<const P4 *>(s->d_pos[0]);
	const P4 *rest = reinterpret_cast<const P4 *>(s->d_rest);
	const P4 *vel = reinterpret_cast<const P4 *>(s->d_vel);
	const TetColliderView *views = s->d_tet_views;
	const uint32_t nc = (uint32_t)s->tet_views.size();
	if (s->tet_serial)
	{
		for (uint32_t c = 0; c < nc; c++)
		{
			const TetColliderView &v = s->tet_views[c];
			hipLaunchKernelGGL(tet_hull_kernel, dim3((v.points.num_nodes + 255) / 256), dim3(256), 0, s->stream, views, c, pos, 0);
			hipLaunchKernelGGL(tet_hull_kernel, dim3((v.tet_bvh.num_nodes + 255) / 256), dim3(256), 0, s->stream, views, c, pos, 1);
		}
		hipLaunchKernelGGL(tet_aabb_kernel, dim3(nc), dim3(256), 0, s->stream, views, pos, s->d_tet_aabb);
		hipLaunchKernelGGL(tet_detect_kernel, dim3(1), dim3(64), 0, s->stream, views, nc, pos, rest, vel, (const float *)s->d_tet_aabb, s->d_tet_contacts, s->d_tet_counters, s->tet_work.max_contacts);
	}
	else
	{
		uint32_t max_nodes = 1, max_elems = 1;
		for (const TetColliderView &v : s->tet_views)
		{
			max_nodes = std::max(max_nodes, std::max(v.points.num_nodes, v.tet_bvh.num_nodes));
			max_elems = std::max(max_elems, std::max(v.num_vertices, 4u * v.num_tets));
		}
		s->tet_timed(0, [&] {
			hipLaunchKernelGGL(tet_gather_kernel, dim3((max_elems + 255) / 256, 2 * nc), dim3(256), 0, s->stream, views, pos);
			// the few long chains (one CU each, ~1 ms for 330k vertices) and the tens of thousands of short ones do not depend on each other
			const bool fork = s->tet_big_count != 0;
			hipStream_t side = fork ? s->stream_side : s->stream;
			if (fork) { (void)hipEventRecord(s->ev_fork, s->stream); (void)hipStreamWaitEvent(side, s->ev_fork, 0); }
			hipLaunchKernelGGL(tet_hull_kernel2, dim3(max_nodes, 2 * nc), dim3(192), 0, side, views, (const uint32_t *)nullptr, (uint32_t *)nullptr);
			hipLaunchKernelGGL(tet_aabb_kernel, dim3(nc), dim3(256), 0, side, views, pos, s->d_tet_aabb);
			if (fork)
			{
				(void)hipEventRecord(s->ev_join, side);
				hipLaunchKernelGGL(tet_hull_kernel2, dim3(s->tet_big_count), dim3(192), kTcBigNodeLds, s->stream, views, (const uint32_t *)s->d_tet_big, s->d_tet_big_r2);
				hipLaunchKernelGGL(tet_big_radius_kernel, dim3(s->tet_big_slices), dim3(256), 0, s->stream, views, (const uint32_t *)s->d_tet_big, (const uint32_t *)s->d_tet_big_slices, s->d_tet_big_r2);
				hipLaunchKernelGGL(tet_big_finish_kernel, dim3((s->tet_big_count + 255) / 256), dim3(256), 0, s->stream, views, (const uint32_t *)s->d_tet_big, s->tet_big_count, (const uint32_t *)s->d_tet_big_r2);
				(void)hipStreamWaitEvent(s->stream, s->ev_join, 0);
			}
		});
		s->tet_timed(2, [&] {
			// all workgroups resident at once (they meet at barriers); the launch's scratch words start at zero
			(void)hipMemsetAsync(s->tet_work.trav, 0, kTrWords * sizeof(uint32_t), s->stream);
			// a quarter of the CUs: a generation is a few microseconds of work, and what it costs is the barrier -- 256 arrivals on one
			// counter take longer than the work they separate (measured, medium / large scene: 32 workgroups 0.30 ms, 64: 0.28 / 0.74,
			// 128: 0.30, 256: 0.42 / 0.80)
			const uint32_t cus = (uint32_t)std::max(1, s->prop.multiProcessorCount);
			uint32_t wgs = std::max(1u, cus / 4u);
			if (const char *e = getenv("PBDX_TET_TRAVERSE_WGS")) { const int v = atoi(e); if (v >= 1 && (uint32_t)v <= cus) wgs = (uint32_t)v; }      // developer aid
			hipLaunchKernelGGL(tet_traverse_kernel, dim3(wgs), dim3(256), 0, s->stream, views, nc, (const float *)s->d_tet_aabb, s->tet_work);
		});
		const uint32_t grid = (uint32_t)std::max(1, s->prop.multiProcessorCount) * 8u;
		s->tet_timed(3, [&] { hipLaunchKernelGGL(tet_candidates_kernel<false>, dim3(grid), dim3(256), 0, s->stream, views, pos, rest, vel, s->tet_work, s->d_tet_contacts); });
		s->tet_timed(4, [&] { hipLaunchKernelGGL(tet_chunk_scan_kernel, dim3(1), dim3(1024), 0, s->stream, s->tet_work); });
		s->tet_timed(5, [&] { hipLaunchKernelGGL(tet_candidates_kernel<true>, dim3(grid), dim3(256), 0, s->stream, views, pos, rest, vel, s->tet_work, s->d_tet_contacts); });
		s->tet_timed(6, [&] { hipLaunchKernelGGL(tet_levels_kernel, dim3(1), dim3(1024), 0, s->stream, (const TetContact *)s->d_tet_contacts, s->tet_work); });
	}
	// which contacts carry a velocity impulse (pMax < 0), in list order; marks on the dynamic particles they touch
	hipLaunchKernelGGL(tet_impulse_list_kernel, dim3(1), dim3(1024), 0, s->stream, (const TetContact *)s->d_tet_contacts, pos, s->tet_work);
	HIPCHECK(hipGetLastError());
	return PBDX_OK;
}

// The detection's scratch (node pairs of the traversal, overlapping leaf pairs, candidate chunks, the contact list) has capacities; the
// reference's vectors have none.  The kernels check every capacity and flag an overflow instead of writing; the flags are read back
// after EVERY detection (one small synchronisation per step, for scenes with deformable colliders only), the exhausted buffer is made
// four times as large and the detection -- which only reads the particle state -- is repeated.  What cannot be grown away (an abandoned
// barrier, more than 256 generations of the recursion, the address space of 32-bit indices) stays flagged and is reported by the step.
int enqueue_tet_detection(pbdx_solver *s)
{
	if (!s->tet_active() || !s->n) return PBDX_OK;
	for (int attempt = 0; attempt < 8; attempt++)
	{
		int r = launch_tet_detection(s);
		if (r) return r;
		uint32_t c[kTcWords];
		HIPCHECK(hipMemcpyAsync(c, s->d_tet_counters, sizeof(c), hipMemcpyDeviceToHost, s->stream));
		HIPCHECK(hipStreamSynchronize(s->stream));
		const bool more_nodes = c[kTcStack] == 1u || c[kTcStack] == 4u || (s->tet_serial && c[kTcStack]);
		const bool more_contacts = c[kTcOverflow] == 1u;
		if (!more_nodes && !more_contacts) { s->tet_impulses_last = c[kTcImpulses]; s->tet_impulses_total += c[kTcImpulses]; return PBDX_OK; }
		// (a repeated detection starts from clean marks: the ones the abandoned attempt set are removed by the list it compacted)
		hipLaunchKernelGGL(tet_impulse_clear_kernel, dim3(64), dim3(256), 0, s->stream, (const TetContact *)s->d_tet_contacts, s->tet_work);
		const uint64_t nodes = more_nodes ? (uint64_t)s->tet_work.node_cap * 4u : s->tet_work.node_cap;
		const uint64_t contacts = more_contacts ? (uint64_t)s->tet_work.max_contacts * 4u : s->tet_work.max_contacts;
		// (an abandoned detection carries no impulse list: the count of the previous step must not drive this step's impulse kernels)
		if (nodes > (1ull << 29) || contacts > (1ull << 24) || (s->tet_serial && more_nodes)) { s->tet_impulses_last = 0; return PBDX_OK; }      // reported by the step
		s->tet_grown++;
		r = alloc_tet_work(s, nodes, (uint32_t)contacts);
		if (r) return r;
		HIPCHECK(hipMemsetAsync(s->d_tet_counters, 0, kTcWords * sizeof(uint32_t), s->stream));
	}
	s->tet_impulses_last = 0;
	return PBDX_OK;
}

// scratch and order tables of the contact solve with dynamic bodies (include/pbdx.h); range_pos[r] = place of range r among the ranges by object index
int prepare_dynamic_contacts(pbdx_solver *s, std::vector<uint32_t> &range_pos)
{
	const uint32_t nc = (uint32_t)s->colliders.size(), nr = (uint32_t)s->ranges.size();
	if (s->tet_active()) { set_error("dynamic rigid bodies are not combined with contacts between deformable solids"); return PBDX_ERR_UNSUPPORTED; }
	if (nc > kMaxDynColliders || (uint64_t)nc * nr > 255u) { set_error("dynamic rigid bodies: at most %u colliders and 255 (range, collider) pairs", kMaxDynColliders); return PBDX_ERR_UNSUPPORTED; }
	if (s->range_object.size() != nr || s->rank_count != s->n || !s->d_rank) { set_error("dynamic rigid bodies: pbdx_solver_set_contact_order has not been called for these ranges / particles"); return PBDX_ERR_INVALID; }
	for (uint32_t r = 0; r < nr; r++) { range_pos[r] = 0; for (uint32_t q = 0; q < nr; q++) if (s->range_object[q] < s->range_object[r] || (s->range_object[q] == s->range_object[r] && q < r)) range_pos[r]++; }
	uint64_t in_ranges = 0;
	for (const pbdx_collision_range &r : s->ranges) in_ranges += r.count;
	for (const pbdx_collision_range &r : s->ranges) if (r.count > (1u << 24)) { set_error("dynamic rigid bodies: a collision range of more than 2^24 particles (the contact order's key holds 24 bits of a particle's place)"); return PBDX_ERR_UNSUPPORTED; }
	const uint32_t cap = (uint32_t)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1024u, in_ranges * std::min<uint32_t>(nc, PBDX_MAX_CONTACTS_PER_PARTICLE)));
	if (cap != s->dyn_cap)
	{
		s->free_dynamic_work();
		HIPCHECK(hipMalloc(&s->d_dyn_list, (size_t)cap * sizeof(DynContact)));
		HIPCHECK(hipMalloc(&s->d_dyn_keys, (size_t)cap * 4)); HIPCHECK(hipMalloc(&s->d_dyn_keys_sorted, (size_t)cap * 4));
		HIPCHECK(hipMalloc(&s->d_dyn_vals, (size_t)cap * 4)); HIPCHECK(hipMalloc(&s->d_dyn_vals_sorted, (size_t)cap * 4));
		HIPCHECK(hipMalloc(&s->d_dyn_info, (size_t)cap * sizeof(DynContactInfo))); HIPCHECK(hipMalloc(&s->d_dyn_sum, (size_t)cap * 4));
		s->dyn_sort_temp_bytes = 0;
		{ int rs = sort_pairs_u32(nullptr, &s->dyn_sort_temp_bytes, s->d_dyn_keys, s->d_dyn_keys_sorted, s->d_dyn_vals, s->d_dyn_vals_sorted, cap, s->stream); if (rs) return rs; }
		HIPCHECK(hipMalloc(&s->d_dyn_sort_temp, std::max<size_t>(s->dyn_sort_temp_bytes, 16)));
		s->dyn_cap = cap;
	}
	if (!s->d_dyn_counters) HIPCHECK(hipMalloc(&s->d_dyn_counters, 2 * sizeof(unsigned int)));
	if (!s->d_dyn_body) HIPCHECK(hipMalloc(&s->d_dyn_body, (size_t)kMaxDynColliders * 8 * sizeof(float)));
	HIPCHECK(hipMemsetAsync(s->d_dyn_counters, 0, 2 * sizeof(unsigned int), s->stream));
	HIPCHECK(hipMemsetAsync(s->d_dyn_keys, 0xff, (size_t)cap * 4, s->stream));
	HIPCHECK(hipMemsetAsync(s->d_dyn_vals, 0, (size_t)cap * 4, s->stream));
	return PBDX_OK;
}

int enqueue_contacts(pbdx_solver *s)
{
	{
		int rt = enqueue_tet_detection(s);
		if (rt) return rt;
	}
	const bool rigid = !s->colliders.empty() && !s->ranges.empty() && s->n;
	const bool dynamic = rigid && s->dynamics.size() == s->colliders.size() && s->any_dynamic();
	std::vector<uint32_t> range_pos(s->ranges.size(), 0u);
	if (dynamic)
	{
		int rd = prepare_dynamic_contacts(s, range_pos);
		if (rd) return rd;
	}
	if (rigid)
	{
		HIPCHECK(hipMemsetAsync(s->d_contact_counters, 0, sizeof(unsigned int), s->stream));      // [0] contacts of this step; [1] (overflow) is reset per call
		for (const pbdx_collision_range &r : s->ranges)
		{
			if (!r.count) continue;
			ContactArgs a;
			a.dyn = nullptr; a.rank = nullptr; a.pair_base = 0u; a.collider_pos = nullptr; a.range_index = (uint32_t)(&r - s->ranges.data());
			a.list = nullptr; a.list_keys = a.list_vals = nullptr; a.list_cap = 0u; a.list_counters = nullptr;
			if (dynamic)
			{
				a.dyn = s->d_dynamics; a.rank = s->d_rank; a.pair_base = range_pos[a.range_index] * (uint32_t)s->colliders.size(); a.collider_pos = s->d_collider_pos;
				a.list = s->d_dyn_list; a.list_keys = s->d_dyn_keys; a.list_vals = s->d_dyn_vals; a.list_cap = s->dyn_cap; a.list_counters = s->d_dyn_counters;
			}
			a.pos = s->d_pos[0]; a.vel = s->d_vel; a.colliders = s->d_colliders; a.num_colliders = (uint32_t)s->colliders.size();
			a.first = r.first; a.count = r.count;
			a.tolerance = s->contact_tolerance; a.stiffness = s->contact_stiffness; a.restitution = r.restitution; a.friction = r.friction;
			a.iterations = s->max_iterations_v;
			a.counters = s->d_contact_counters;
			a.ctl = ctl_of(s, nullptr);
			a.imp_mark = s->tet_active() ? s->tet_work.imp_mark : nullptr;
			hipLaunchKernelGGL(contact_kernel, dim3((r.count + 255) / 256), dim3(256), 0, s->stream, a);
			HIPCHECK(hipGetLastError());
		}
		if (dynamic)
		{
			// the list in the reference's order (unused places carry the largest key), then the sequential solve
			size_t tb = s->dyn_sort_temp_bytes;
			int rs = sort_pairs_u32(s->d_dyn_sort_temp, &tb, s->d_dyn_keys, s->d_dyn_keys_sorted, s->d_dyn_vals, s->d_dyn_vals_sorted, s->dyn_cap, s->stream);
			if (rs) return rs;
			DynSolveArgs d;
			d.pos = s->d_pos[0]; d.vel = s->d_vel; d.colliders = s->d_colliders; d.dyn = s->d_dynamics; d.num_colliders = (uint32_t)s->colliders.size();
			d.ranges = s->d_ranges; d.list = s->d_dyn_list; d.order = s->d_dyn_vals_sorted; d.info = s->d_dyn_info; d.sum_impulses = s->d_dyn_sum;
			d.list_counters = s->d_dyn_counters; d.body = s->d_dyn_body; d.stiffness = s->contact_stiffness; d.iterations = s->max_iterations_v; d.ctl = ctl_of(s, nullptr);
			hipLaunchKernelGGL(dyn_contact_solve_kernel, dim3(1), dim3(256), 0, s->stream, d);
			HIPCHECK(hipGetLastError());
		}
	}
	// velocity solve of the particle-tet contacts (friction 0: only contacts with pMax < 0 carry an impulse; the host knows how many from the
	// counters the detection read back)
	if (s->tet_active() && s->n && s->tet_impulses_last)
	{
		TetImpulseArgs a;
		a.contacts = s->d_tet_contacts; a.pos = s->d_pos[0]; a.vel = s->d_vel; a.w = s->tet_work;
		a.colliders = rigid ? s->d_colliders : nullptr; a.num_colliders = rigid ? (uint32_t)s->colliders.size() : 0u;
		a.ranges = rigid ? s->d_ranges : nullptr; a.num_ranges = rigid ? (uint32_t)s->ranges.size() : 0u;
		a.tolerance = s->contact_tolerance; a.stiffness = s->contact_stiffness; a.iterations = s->max_iterations_v;
		a.contact_counters = rigid ? s->d_contact_counters : nullptr;
		const uint32_t blocks = std::min(256u, (5u * s->tet_impulses_last + 255u) / 256u);
		// a particle's (contact, role) entries side by side and in list order: pairs keyed by particle, stable radix sort
		hipLaunchKernelGGL(tet_impulse_pairs_kernel, dim3(blocks), dim3(256), 0, s->stream, (const TetContact *)s->d_tet_contacts, s->tet_work);
		{
			size_t tb = s->tet_sort_temp_bytes;
			int rs = sort_pairs_u32(s->tet_work_alloc[21], &tb, s->tet_work.imp_keys, s->tet_work.imp_keys_sorted, s->tet_work.imp_slots, s->tet_work.imp_slots_sorted,
				5u * s->tet_impulses_last, s->stream);
			if (rs) return rs;
		}
		hipLaunchKernelGGL(tet_impulse_kernel, dim3(blocks), dim3(256), 0, s->stream, a);
		hipLaunchKernelGGL(tet_impulse_clear_kernel, dim3(blocks), dim3(256), 0, s->stream, (const TetContact *)s->d_tet_contacts, s->tet_work);
		HIPCHECK(hipGetLastError());
	}
	return PBDX_OK;
}

int collect_profile(pbdx_solver *s, ProfCursor *pc)
{
	HIPCHECK(hipStreamSynchronize(s->stream));
	for (size_t i = 0; i < pc->kinds.size(); i++)
	{
		const int t = pc->kinds[i];
		float ms = 0.0f;
		HIPCHECK(hipEventElapsedTime(&ms, s->prof_events[2 * i], s->prof_events[2 * i + 1]));
		if (t >= 0)
		{
			s->type_ms[t] += ms;
			s->type_launches[t]++;
			s->type_projections[t] += pc->counts[i];
		}
		else if (t == -1)
		{
			s->persist_ms += ms;
			s->persist_launches++;
		}
		else
		{
			DeviceSegment &d = s->dsegs[(size_t)(-2 - t)];
			d.ms += ms;
			d.launches++;
		}
		s->stats.projection_ms += ms;
		s->stats.projection_launches++;
	}
	return PBDX_OK;
}

// the staging buffer holds the caller's packed arrays in the caller's scalar type
int ensure_stage(pbdx_solver *s, size_t elem)
{
	if (elem <= s->stage_elem || !s->n) return PBDX_OK;
	HIPCHECK(hipStreamSynchronize(s->stream));
	if (s->d_stage) { (void)hipFree(s->d_stage); s->d_stage = nullptr; }
	HIPCHECK(hipMalloc(&s->d_stage, (size_t)s->n * 14 * elem));
	s->stage_elem = elem;
	return PBDX_OK;
}

template <class T>
int set_particles_impl(pbdx_solver *s, uint32_t n, const T *x, const T *v, const T *old_x, const T *last_x, const T *mass, const T *inv_mass)
{
	if (!s || !x || !mass || !inv_mass) { set_error("set_particles: x, mass and inv_mass are required"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	if (n != s->n)
	{
		HIPCHECK(hipStreamSynchronize(s->stream));
		s->free_particles();
		s->free_tet_colliders();   // they refer to particle ranges of the old image as well
		s->drop_graph();
		s->free_batches();         // a schedule refers to particle indices of the old image: it must be re-added
		s->schedule_version++;
		if (n)
		{
			HIPCHECK(hipMalloc(&s->d_pos[0], (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_pos[1], (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_vel, (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_old, (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_last, (size_t)n * sizeof(float4)));
			HIPCHECK(hipMalloc(&s->d_stage, (size_t)n * 14 * sizeof(T)));
			s->stage_elem = sizeof(T);
		}
		s->n = n;
	}
	if (!n) return PBDX_OK;
	{ int rs = ensure_stage(s, sizeof(T)); if (rs) return rs; }
	if (!s->plan_ok)     // tile partition input (first upload wins)
	{
		s->h_x.resize((size_t)3 * n);
		for (size_t i = 0; i < (size_t)3 * n; i++) s->h_x[i] = (float)x[i];
	}
	// raw packed arrays -> device staging (stream ordered), repacked into float4 on the device
	T *st_x = reinterpret_cast<T *>(s->d_stage), *st_v = st_x + (size_t)3 * n, *st_o = st_v + (size_t)3 * n, *st_l = st_o + (size_t)3 * n;
	T *st_m = st_l + (size_t)3 * n, *st_w = st_m + n;
	const size_t b3 = (size_t)3 * n * sizeof(T), b1 = (size_t)n * sizeof(T);
	{
		// (with the mirror: the host copy of an array runs while the previous array is on the bus)
		char *mir = s->mirror((size_t)14 * n * sizeof(T));
		struct { T *dst; const T *src; size_t bytes; } jobs[6] = { { st_x, x, b3 }, { st_v, v, b3 }, { st_o, old_x, b3 }, { st_l, last_x, b3 }, { st_m, mass, b1 }, { st_w, inv_mass, b1 } };
		for (auto &j : jobs)
		{
			if (!j.src) continue;
			// the library's own page-locked memory (pbdx_model's arrays): straight onto the bus, no host-side copy at all
			if (pbdx::is_library_pinned(j.src, j.bytes)) { HIPCHECK(pbdx::copy_pinned_to_device_async(j.dst, j.src, j.bytes, s->stream)); continue; }
			if (!mir) { HIPCHECK(pbdx::copy_to_device(j.dst, j.src, j.bytes)); continue; }       // (the library's bounce buffer)
			char *slot = mir + (reinterpret_cast<char *>(j.dst) - reinterpret_cast<char *>(s->d_stage));
			host_copy(slot, j.src, j.bytes);
			HIPCHECK(hipMemcpyAsync(j.dst, slot, j.bytes, hipMemcpyHostToDevice, s->stream));
		}
	}
	const dim3 grid((n + 255) / 256), block(256);
	hipLaunchKernelGGL(pack_kernel<T>, grid, block, 0, s->stream, (const T *)st_x, (const T *)st_w, s->d_pos[0], n);
	if (!s->rest_set)
	{
		// rest positions (ParticleData::m_x0) default to the first uploaded positions; pbdx_solver_set_rest_positions overrides
		if (!s->d_rest) HIPCHECK(hipMalloc(&s->d_rest, (size_t)n * sizeof(float4)));
		hipLaunchKernelGGL(pack_kernel<T>, grid, block, 0, s->stream, (const T *)st_x, (const T *)st_w, s->d_rest, n);
		s->rest_set = true;
	}
	if (v) hipLaunchKernelGGL(pack_kernel<T>, grid, block, 0, s->stream, (const T *)st_v, (const T *)st_m, s->d_vel, n);
	else hipLaunchKernelGGL(pack_zero_kernel<T>, grid, block, 0, s->stream, (const T *)st_m, s->d_vel, n);
	hipLaunchKernelGGL(pack_kernel<T>, grid, block, 0, s->stream, (const T *)(old_x ? st_o : st_x), (const T *)nullptr, s->d_old, n);
	hipLaunchKernelGGL(pack_kernel<T>, grid, block, 0, s->stream, (const T *)(last_x ? st_l : st_x), (const T *)nullptr, s->d_last, n);
	HIPCHECK(hipGetLastError());
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}


template <class T>
int get_particles_impl(pbdx_solver *s, uint32_t n, T *x, T *v, T *old_x, T *last_x)
{
	if (!s || n != s->n) { set_error("get_particles: particle count mismatch (%u vs %u)", n, s ? s->n : 0); return PBDX_ERR_INVALID; }
	if (!n) return PBDX_OK;
	ENTER_DEVICE(s->device);
	{ int rs = ensure_stage(s, sizeof(T)); if (rs) return rs; }
	struct { T *dst; const float4 *src; } jobs[4] = { { x, s->d_pos[0] }, { v, s->d_vel }, { old_x, s->d_old }, { last_x, s->d_last } };
	const size_t b3 = (size_t)3 * n * sizeof(T);
	char *mir = s->mirror((size_t)14 * n * sizeof(T));
	bool direct[4] = { false, false, false, false };
	int k = 0;
	for (auto &j : jobs)
	{
		const int q = k++;
		T *st = reinterpret_cast<T *>(s->d_stage) + (size_t)3 * n * q;
		if (!j.dst) continue;
		hipLaunchKernelGGL(unpack_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, s->stream, j.src, st, n);
		HIPCHECK(hipGetLastError());
		if (pbdx::is_library_pinned(j.dst, b3)) { HIPCHECK(pbdx::copy_device_to_pinned_async(j.dst, st, b3, s->stream)); direct[q] = true; continue; }      // (the library's own page-locked memory)
		if (!mir) { HIPCHECK(pbdx::copy_from_device(j.dst, st, b3)); continue; }       // (waits for the kernel; the library's bounce buffer)
		HIPCHECK(hipMemcpyAsync(mir + b3 * q, st, b3, hipMemcpyDeviceToHost, s->stream));
		HIPCHECK(hipEventRecord(s->mirror_ev[q], s->stream));
	}
	if (mir)
	{
		// array q leaves the mirror while array q + 1 is still on the bus
		k = 0;
		for (auto &j : jobs)
		{
			const int q = k++;
			if (!j.dst || direct[q]) continue;
			HIPCHECK(hipEventSynchronize(s->mirror_ev[q]));
			host_copy(j.dst, mir + b3 * q, b3);
		}
	}
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

template <class T>
int get_particles_hashed_impl(pbdx_solver *s, uint32_t n, T *x, T *v, T *old_x, T *last_x, uint64_t *hx, uint64_t *hv, uint64_t *ho, uint64_t *hl)
{
	if (!s || n != s->n) { set_error("get_particles: particle count mismatch (%u vs %u)", n, s ? s->n : 0); return PBDX_ERR_INVALID; }
	if (!n) return PBDX_OK;
	ENTER_DEVICE(s->device);
	{ int rs = ensure_stage(s, sizeof(T)); if (rs) return rs; }
	const uint32_t nb = pbdx_hash_num_blocks(n);
	if (s->hash_blocks < nb)
	{
		HIPCHECK(hipStreamSynchronize(s->stream));
		if (s->d_hash) { (void)hipFree(s->d_hash); s->d_hash = nullptr; }
		HIPCHECK(hipMalloc(&s->d_hash, (size_t)4 * nb * sizeof(uint64_t)));
		s->hash_blocks = nb;
	}
	struct { T *dst; const float4 *src; uint64_t *h; } jobs[4] = { { x, s->d_pos[0], hx }, { v, s->d_vel, hv }, { old_x, s->d_old, ho }, { last_x, s->d_last, hl } };
	const size_t b3 = (size_t)3 * n * sizeof(T);
	const uint32_t elem_words = 3 * (uint32_t)(sizeof(T) / 4);
	const size_t hb = (size_t)nb * sizeof(uint64_t), stage_bytes = (size_t)14 * n * sizeof(T);
	char *mir = s->mirror(stage_bytes + 4 * hb);
	int k = 0;
	for (auto &j : jobs)
	{
		const int q = k++;
		T *st = reinterpret_cast<T *>(s->d_stage) + (size_t)3 * n * q;
		if (!j.dst && !j.h) continue;
		hipLaunchKernelGGL(unpack_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, s->stream, j.src, st, n);
		if (j.h)
			hipLaunchKernelGGL(hash_blocks_kernel, dim3(nb), dim3(256), 0, s->stream, reinterpret_cast<const uint32_t *>(st), (uint64_t)n * elem_words * 4u,
				PBDX_HASH_BLOCK * elem_words * 4u, s->d_hash + (size_t)q * nb);
		HIPCHECK(hipGetLastError());
		if (!mir)
		{
			if (j.dst) HIPCHECK(pbdx::copy_from_device(j.dst, st, b3));
			if (j.h) HIPCHECK(pbdx::copy_from_device(j.h, s->d_hash + (size_t)q * nb, hb));
			continue;
		}
		if (j.dst) HIPCHECK(hipMemcpyAsync(mir + b3 * q, st, b3, hipMemcpyDeviceToHost, s->stream));
		if (j.h) HIPCHECK(hipMemcpyAsync(mir + stage_bytes + hb * q, s->d_hash + (size_t)q * nb, hb, hipMemcpyDeviceToHost, s->stream));
		HIPCHECK(hipEventRecord(s->mirror_ev[q], s->stream));
	}
	if (mir)
	{
		k = 0;
		for (auto &j : jobs)
		{
			const int q = k++;
			if (!j.dst && !j.h) continue;
			HIPCHECK(hipEventSynchronize(s->mirror_ev[q]));
			if (j.dst) host_copy(j.dst, mir + b3 * q, b3);
			if (j.h) memcpy(j.h, mir + stage_bytes + hb * q, hb);
		}
	}
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

template <class T>
int update_ranges_impl(pbdx_solver *s, int array, const T *base, uint32_t num_ranges, const uint32_t *ranges)
{
	if (!s || !base || (num_ranges && !ranges) || array < PBDX_ARRAY_X || array > PBDX_ARRAY_INV_MASS) { set_error("update_particle_ranges: bad arguments"); return PBDX_ERR_INVALID; }
	if (!s->n || !s->d_pos[0]) { set_error("update_particle_ranges: no particle image yet (pbdx_solver_set_particles first)"); return PBDX_ERR_INVALID; }
	for (uint32_t r = 0; r < num_ranges; r++)
		if (ranges[2 * r] > s->n || ranges[2 * r + 1] > s->n - ranges[2 * r]) { set_error("update_particle_ranges: range %u (%u, %u) outside the %u particles", r, ranges[2 * r], ranges[2 * r + 1], s->n); return PBDX_ERR_INVALID; }
	if (!num_ranges) return PBDX_OK;
	ENTER_DEVICE(s->device);
	{ int rs = ensure_stage(s, sizeof(T)); if (rs) return rs; }
	const uint32_t n = s->n;
	const bool vec = array <= PBDX_ARRAY_LAST_X;
	// the staging slot of the array (layout of set_particles_impl)
	T *st = reinterpret_cast<T *>(s->d_stage) + (vec ? (size_t)3 * n * array : (size_t)12 * n + (size_t)n * (array - PBDX_ARRAY_MASS));
	const size_t per = vec ? 3 : 1;
	char *mir = s->mirror((size_t)14 * n * sizeof(T));
	float4 *dst = array == PBDX_ARRAY_X ? s->d_pos[0] : array == PBDX_ARRAY_V ? s->d_vel : array == PBDX_ARRAY_OLD_X ? s->d_old : array == PBDX_ARRAY_LAST_X ? s->d_last :
		array == PBDX_ARRAY_MASS ? s->d_vel : s->d_pos[0];
	for (uint32_t r = 0; r < num_ranges; r++)
	{
		const uint32_t first = ranges[2 * r], count = ranges[2 * r + 1];
		if (!count) continue;
		if (mir)
		{
			char *slot = mir + (reinterpret_cast<char *>(st + per * first) - reinterpret_cast<char *>(s->d_stage));
			host_copy(slot, base + per * first, per * count * sizeof(T));
			HIPCHECK(hipMemcpyAsync(st + per * first, slot, per * count * sizeof(T), hipMemcpyHostToDevice, s->stream));
		}
		else HIPCHECK(pbdx::copy_to_device(st + per * first, base + per * first, per * count * sizeof(T)));
		if (vec) hipLaunchKernelGGL(update_xyz_kernel<T>, dim3((count + 255) / 256), dim3(256), 0, s->stream, (const T *)st, dst, first, count);
		else hipLaunchKernelGGL(update_w_kernel<T>, dim3((count + 255) / 256), dim3(256), 0, s->stream, (const T *)st, dst, first, count);
	}
	HIPCHECK(hipGetLastError());
	HIPCHECK(hipStreamSynchronize(s->stream));
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
	DeviceScope device_scope_(device);
	hipError_t e = device_scope_.err;
	if (e == hipSuccess) e = hipGetDeviceProperties(&s->prop, device);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream_side, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreate(&s->ev_start);
	if (e == hipSuccess) e = hipEventCreate(&s->ev_stop);
	if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s->h_error), 4 * sizeof(uint32_t), hipHostMallocMapped);
	if (e == hipSuccess) { memset(s->h_error, 0, 4 * sizeof(uint32_t)); e = hipHostGetDevicePointer(reinterpret_cast<void **>(&s->d_error), s->h_error, 0); }
	if (e == hipSuccess) e = hipMalloc(&s->d_ctl, kCtlWords * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMemset(s->d_ctl, 0, kCtlWords * sizeof(uint32_t));
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
	DeviceScope device_scope_(s->device);
	if (s->stream) (void)hipStreamSynchronize(s->stream);
	s->drop_graph();
	s->free_batches();
	if (s->tet_profile && s->tet_launches[0])
	{
		static const char *names[8] = { "spheres + boxes", "-", "traverse", "candidates (ballot)", "chunk scan", "candidates (write)", "levels", "solve (per iteration)" };
		for (int q = 0; q < 8; q++)
			fprintf(stderr, "[pbdx tet profile] %-22s %8llu launches  %10.3f ms total  %8.4f ms each\n", names[q], (unsigned long long)s->tet_launches[q], s->tet_ms[q], s->tet_launches[q] ? s->tet_ms[q] / s->tet_launches[q] : 0.0);
	}
	s->free_particles();
	s->free_tet_colliders();
	if (s->d_tet_contacts) (void)hipFree(s->d_tet_contacts);
	if (s->d_tet_counters) (void)hipFree(s->d_tet_counters);
	s->free_mirror();
	if (s->d_colliders) (void)hipFree(s->d_colliders);
	if (s->d_ranges) (void)hipFree(s->d_ranges);
	s->free_dynamic_work();
	if (s->d_dynamics) (void)hipFree(s->d_dynamics);
	if (s->d_rank) (void)hipFree(s->d_rank);
	if (s->d_collider_pos) (void)hipFree(s->d_collider_pos);
	if (s->d_dyn_body) (void)hipFree(s->d_dyn_body);
	if (s->d_dyn_counters) (void)hipFree(s->d_dyn_counters);
	if (s->d_contact_counters) (void)hipFree(s->d_contact_counters);
	for (hipEvent_t e : s->prof_events) (void)hipEventDestroy(e);
	for (hipEvent_t e : s->sub_events) (void)hipEventDestroy(e);
	if (s->ev_start) (void)hipEventDestroy(s->ev_start);
	if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
	if (s->stream_side) (void)hipStreamDestroy(s->stream_side);
	if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
	if (s->ev_join) (void)hipEventDestroy(s->ev_join);
	if (s->stream) (void)hipStreamDestroy(s->stream);
	if (s->h_error) (void)hipHostFree(s->h_error);
	if (s->d_ctl) (void)hipFree(s->d_ctl);
	delete s;
}

int pbdx_solver_set_particles(pbdx_solver *s, uint32_t n, const float *x, const float *v, const float *old_x,
	const float *last_x, const float *mass, const float *inv_mass)
{
	return set_particles_impl<float>(s, n, x, v, old_x, last_x, mass, inv_mass);
}
int pbdx_solver_set_particles_f64(pbdx_solver *s, uint32_t n, const double *x, const double *v, const double *old_x,
	const double *last_x, const double *mass, const double *inv_mass)
{
	return set_particles_impl<double>(s, n, x, v, old_x, last_x, mass, inv_mass);
}

int pbdx_solver_set_positions(pbdx_solver *s, uint32_t n, const float *x)
{
	if (!s || !x || n != s->n) { set_error("set_positions: particle count mismatch"); return PBDX_ERR_INVALID; }
	if (!n) return PBDX_OK;
	ENTER_DEVICE(s->device);
	HIPCHECK(pbdx::copy_to_device(s->d_stage, x, (size_t)3 * n * sizeof(float)));
	hipLaunchKernelGGL(set_xyz_kernel, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_stage, s->d_pos[0], n);
	HIPCHECK(hipGetLastError());
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_get_particles(pbdx_solver *s, uint32_t n, float *x, float *v, float *old_x, float *last_x)
{
	return get_particles_impl<float>(s, n, x, v, old_x, last_x);
}
int pbdx_solver_get_particles_f64(pbdx_solver *s, uint32_t n, double *x, double *v, double *old_x, double *last_x)
{
	return get_particles_impl<double>(s, n, x, v, old_x, last_x);
}

int pbdx_solver_get_particles_hashed(pbdx_solver *s, uint32_t n, float *x, float *v, float *old_x, float *last_x,
	uint64_t *hash_x, uint64_t *hash_v, uint64_t *hash_old, uint64_t *hash_last)
{
	return get_particles_hashed_impl<float>(s, n, x, v, old_x, last_x, hash_x, hash_v, hash_old, hash_last);
}
int pbdx_solver_get_particles_hashed_f64(pbdx_solver *s, uint32_t n, double *x, double *v, double *old_x, double *last_x,
	uint64_t *hash_x, uint64_t *hash_v, uint64_t *hash_old, uint64_t *hash_last)
{
	return get_particles_hashed_impl<double>(s, n, x, v, old_x, last_x, hash_x, hash_v, hash_old, hash_last);
}
int pbdx_solver_update_particle_ranges(pbdx_solver *s, int array, const float *base, uint32_t num_ranges, const uint32_t *ranges)
{
	return update_ranges_impl<float>(s, array, base, num_ranges, ranges);
}
int pbdx_solver_update_particle_ranges_f64(pbdx_solver *s, int array, const double *base, uint32_t num_ranges, const uint32_t *ranges)
{
	return update_ranges_impl<double>(s, array, base, num_ranges, ranges);
}

int pbdx_solver_begin_schedule(pbdx_solver *s)
{
	if (!s) return PBDX_ERR_INVALID;
	ENTER_DEVICE(s->device);
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
	ENTER_DEVICE(s->device);

	// host image only: the device arrays of schedule (B) are created on first use (ensure_device_batches);
	// the fused schedule (A) never needs them
	s->batches.emplace_back();
	Batch &b = s->batches.back();
	b.type = type; b.group = group; b.count = count; b.seq = (uint32_t)s->batches.size() - 1;
	b.h_idx.assign(indices, indices + (size_t)count * nb);
	b.h_params.assign(params, params + (size_t)count * ti->param_stride);
	compute_type_view(type, { { params, count } }, b.view);
	b.par_stride = (count + 3u) & ~3u;
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

// ---- parameter refresh without replanning (run-time edits of stiffness / rest data in the host application) ----------
int pbdx_solver_update_batch_params(pbdx_solver *s, uint32_t batch_index, uint32_t count, const float *params, uint32_t param_stride)
{
	if (!s || s->schedule_open || batch_index >= s->batches.size() || !params) { set_error("update_batch_params: bad batch / schedule open"); return PBDX_ERR_INVALID; }
	Batch &b = s->batches[batch_index];
	if (count != b.count || param_stride != type_info(b.type)->param_stride) { set_error("update_batch_params: count / stride do not match the batch"); return PBDX_ERR_INVALID; }
	b.h_params.assign(params, params + (size_t)count * param_stride);
	s->params_dirty = true;
	return PBDX_OK;
}

int pbdx_solver_commit_params(pbdx_solver *s)
{
	if (!s || s->schedule_open) return PBDX_ERR_INVALID;
	if (!s->params_dirty) return PBDX_OK;
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->params_dirty = false;
	s->drop_graph();                       // the scalar parameters (TypeView::u) are kernel arguments of the captured launches
	s->schedule_version++;
	// schedule (B): the planar parameter arrays are rebuilt from the host image the next time that schedule runs
	for (Batch &b : s->batches)
	{
		compute_type_view(b.type, { { b.h_params.data(), b.count } }, b.view);
		if (b.d_params) { (void)hipFree(b.d_params); b.d_params = nullptr; }
		if (b.d_idx) { (void)hipFree(b.d_idx); b.d_idx = nullptr; }
	}
	if (!s->plan_ok) return PBDX_OK;
	// schedule (A): same tiles, same slots; only the parameter streams and the scalar views change -- unless a type's
	// records stopped (or started) being compactable, which changes the stream layout: then the plan is rebuilt
	TypeView views[PBDX_NUM_CONSTRAINT_TYPES];
	bool same_layout = true;
	for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++)
	{
		std::vector<ParamSpan> spans;
		for (uint32_t bi : s->order) if (s->batches[bi].type == t && s->batches[bi].count) spans.push_back({ s->batches[bi].h_params.data(), s->batches[bi].count });
		compute_type_view(t, spans, views[t]);
		if (views[t].compact != s->plan.views[t].compact) same_layout = false;
	}
	if (!same_layout) { s->free_plan(); return PBDX_OK; }
	// dictionary-form steps index tables of DISTINCT records: if a record that sits in a table changed (a streamed parameter: a rest geometry entry),
	// the tables are stale and the plan is rebuilt by the next step; scalar edits (stiffnesses of the compact layout) leave them as they are
	for (const FusedSegment &seg : s->plan.segs)
	{
		if (!seg.max_tab_f4) continue;
		for (const FusedTile &t : seg.tiles)
			for (uint32_t si2 = t.step_begin; si2 < t.step_end; si2++)
			{
				const FusedStep &st = seg.steps[si2];
				if (!st.dict) continue;
				const int type = (int)st.type;
				const TypeInfo *ti = type_info(type);
				const bool compact = s->plan.views[type].compact != 0;
				const uint16_t *entry = reinterpret_cast<const uint16_t *>(&seg.params[st.par_off]);
				for (uint32_t q = 0; q < st.count; q++)
				{
					const uint32_t cid = seg.slot_cid[st.cid_off + q];
					const size_t pos = (size_t)(std::upper_bound(s->plan.batch_base.begin(), s->plan.batch_base.end(), cid) - s->plan.batch_base.begin()) - 1;
					const Batch &b = s->batches[s->order[pos]];
					const float *rec = b.h_params.data() + (size_t)(cid - s->plan.batch_base[pos]) * ti->param_stride;
					const float *have = &seg.params[((size_t)t.tab_off + entry[q]) * 4];
					for (uint32_t k = 0; k < ti->param_stride; k++)
						if (param_streams(type, compact, (int)k) && memcmp(&have[param_plane(type, compact, (int)k)], &rec[k], 4)) { s->free_plan(); return PBDX_OK; }
				}
			}
	}
	for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++) s->plan.views[t] = views[t];
	for (size_t si = 0; si < s->plan.segs.size(); si++)
	{
		FusedSegment &seg = s->plan.segs[si];
		for (const FusedStep &st : seg.steps)
		{
			const int type = (int)st.type;
			const TypeInfo *ti = type_info(type);
			const bool compact = s->plan.views[type].compact != 0;
			const uint32_t np_stream = (uint32_t)num_planes(type, compact);
			if (!np_stream || st.dict) continue;       // (dictionary form: checked above, nothing to rewrite)
			for (uint32_t q = 0; q < st.count; q++)
			{
				const uint32_t cid = seg.slot_cid[st.cid_off + q];
				// execution-order batch of the constraint: the last batch_base <= cid
				const size_t pos = (size_t)(std::upper_bound(s->plan.batch_base.begin(), s->plan.batch_base.end(), cid) - s->plan.batch_base.begin()) - 1;
				const Batch &b = s->batches[s->order[pos]];
				const float *rec = b.h_params.data() + (size_t)(cid - s->plan.batch_base[pos]) * ti->param_stride;
				for (uint32_t k = 0; k < ti->param_stride; k++)
					if (param_streams(type, compact, (int)k))
						seg.params[st.par_off + param_float_index(seg.vector_params, np_stream, (uint32_t)param_plane(type, compact, (int)k), q)] = rec[k];
			}
		}
		if (!seg.params.empty())
			HIPCHECK(pbdx::copy_to_device(s->dsegs[si].d_params, seg.params.data(), seg.params.size() * sizeof(float)));
		// the packed records of the index image carry a copy of the one-plane types' parameter: rebuilt (same layout, same offsets)
		{
			std::vector<uint8_t> idx_img;
			std::vector<StepImage> step_img;
			build_idx_image(seg, s->plan.views, s->dsegs[si].block, s->pack_plain, idx_img, step_img);
			if (idx_img.size() != s->dsegs[si].idx_bytes) { s->free_plan(); return PBDX_OK; }      // (cannot happen: the layout depends on counts and views only)
			if (!idx_img.empty()) HIPCHECK(pbdx::copy_to_device(s->dsegs[si].d_idx, idx_img.data(), idx_img.size()));
		}
	}
	return PBDX_OK;
}

int pbdx_solver_set_instancing(pbdx_solver *s, uint32_t particles_per_instance, uint32_t instances)
{
	if (!s) return PBDX_ERR_INVALID;
	if (instances > 1 && !particles_per_instance) { set_error("set_instancing: particles_per_instance must be positive"); return PBDX_ERR_INVALID; }
	const uint32_t k = instances ? instances : 1u;
	if (k != s->inst_count || (k > 1 && particles_per_instance != s->inst_particles))
	{
		ENTER_DEVICE(s->device);
		HIPCHECK(hipStreamSynchronize(s->stream));
		s->inst_count = k; s->inst_particles = k > 1 ? particles_per_instance : 0;
		s->drop_graph();
		if (s->plan_built) s->free_plan();
	}
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
	bool replan = false;
	switch (option)
	{
	case PBDX_OPT_USE_GRAPH: s->use_graph = value != 0; break;
	case PBDX_OPT_BLOCK_SIZE:
		if (value != 64 && value != 128 && value != 256) { set_error("block size must be 64, 128 or 256"); return PBDX_ERR_INVALID; }
		s->block_size = (int)value; break;
	case PBDX_OPT_XCD_REMAP: s->xcd_remap = value != 0; break;
	case PBDX_OPT_FUSE:
		if (value < 0 || value > 2) { set_error("fuse must be 0 (per-colour), 1 (fused) or 2 (auto)"); return PBDX_ERR_INVALID; }
		s->fuse = (int)value; replan = true; break;
	case PBDX_OPT_TILE_PARTICLES:
		if (value < 0 || value > 10240) { set_error("tile_particles must be 0 (auto) .. 10240"); return PBDX_ERR_INVALID; }
		s->tile_particles = (uint32_t)value; replan = true; break;
	case PBDX_OPT_FUSE_BLOCK:
		if (value != 0 && value != 256 && value != 512 && value != 768 && value != 1024) { set_error("fused block size must be 0 (auto), 256, 512, 768 or 1024"); return PBDX_ERR_INVALID; }
		s->fuse_block = (int)value; replan = true; break;
	case PBDX_OPT_MAX_SEGMENT_COLOURS:
		if (value < 1) { set_error("max_segment_colours must be >= 1"); return PBDX_ERR_INVALID; }
		s->max_segment_colours = (uint32_t)value; replan = true; break;
	case PBDX_OPT_LDS_PARTICLES:
		if (value < 64 || value > 10240) { set_error("lds_particles must be 64 .. 10240"); return PBDX_ERR_INVALID; }
		s->lds_particles = (uint32_t)value; replan = true; break;
	case PBDX_OPT_TRACE: s->trace = value != 0; break;
	case PBDX_OPT_PAIRS: break;        // removed (the packed two-slots-per-lane projection measured 10-30 % slower): accepted and ignored
	case PBDX_OPT_PIN_HOST: s->pin_host = value != 0; if (!s->pin_host) { (void)hipStreamSynchronize(s->stream); s->free_mirror(); } break;
	case PBDX_OPT_PERSISTENT:
		if (value < 0 || value > 4) { set_error("persistent must be 0 .. 4"); return PBDX_ERR_INVALID; }
		s->persistent = (int)value; replan = true; break;
	case PBDX_OPT_PERSISTENT_WGS_PER_CU:
		if (value < 1 || value > 4) { set_error("persistent workgroups per CU must be 1 .. 4"); return PBDX_ERR_INVALID; }
		s->persist_wgs_per_cu = (uint32_t)value; replan = true; break;
	case PBDX_OPT_TET_CONTACTS_SERIAL:
		s->tet_serial = value ? 1 : 0; break;
	case PBDX_OPT_TET_FORCE_IMPULSES:
		s->tet_force_impulses = value ? 1 : 0; s->tet_work.force_impulses = s->tet_force_impulses; break;
	case PBDX_OPT_SUBSTEP_EVENTS:
		// measurement only: the captured graph stays.  A value n > 1 also creates n events now, so that the first measured call does not
		// pay for them (a few µs each: 0.1-0.2 ms of a 20-step call)
		s->substep_events = value ? 1 : 0;
		if (value > 1 && value <= 65536)
		{
			ENTER_DEVICE(s->device);
			while (s->sub_events.size() < (size_t)value) { hipEvent_t e = nullptr; HIPCHECK(hipEventCreate(&e)); s->sub_events.push_back(e); }
		}
		return PBDX_OK;
	case PBDX_OPT_PERSISTENT_TIMEOUT_MS:
		if (value < 1 || value > 10000) { set_error("persistent timeout must be 1 .. 10000 ms"); return PBDX_ERR_INVALID; }
		s->persist_timeout_ms = (uint32_t)value; break;
	default: set_error("unknown option %d", option); return PBDX_ERR_INVALID;
	}
	s->drop_graph();
	if (replan && s->plan_built)
	{
		ENTER_DEVICE(s->device);
		HIPCHECK(hipStreamSynchronize(s->stream));
		s->free_plan();
	}
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
	// bodies of finite mass move between steps and the host integrates them: one step per call (include/pbdx.h, dynamic rigid bodies as impulse sinks)
	if (num_steps > 1 && !s->ranges.empty() && s->dynamics.size() == s->colliders.size() && s->any_dynamic())
	{ set_error("step: %u steps in one call with dynamic rigid bodies among the colliders (their poses are the host's to update between steps)", num_steps); return PBDX_ERR_UNSUPPORTED; }
	ENTER_DEVICE(s->device);
	const bool fresh_plan = !s->plan_built;
	int rp = ensure_plan(s);
	if (!rp && (fresh_plan || s->autotuned_for_tet != s->tet_active())) rp = autotune_schedule(s);
	if (!rp) rp = ensure_trace(s);
	if (!rp && !s->fused_active()) rp = ensure_device_batches(s);
	if (rp) return rp;
	const float hs = h / (float)sub_steps;                  // TimeStepController.cpp:91
	const float inv_h = (float)(1.0 / (double)hs);          // TimeIntegration.cpp:50 evaluates 1.0/h in double

	uint64_t proj_per_sweep = 0, bytes_per_sweep = 0;
	for (const Batch &b : s->batches) { proj_per_sweep += b.count; bytes_per_sweep += (uint64_t)b.count * type_info(b.type)->algorithmic_bytes; }
	const uint64_t substeps_total = (uint64_t)sub_steps * num_steps;
	const uint64_t launches_per_sweep = s->fused_active() ? s->dsegs.size() : s->order.size();
	s->stats = pbdx_step_stats();
	s->stats.projections = proj_per_sweep * max_iterations * substeps_total;
	s->stats.kernel_launches = (s->persistent_active() ? 1 : s->persistent_iter_active() ? 2 * max_iterations + 2 : launches_per_sweep * max_iterations + 2) * substeps_total;
	s->stats.algorithmic_bytes = (bytes_per_sweep * max_iterations + (uint64_t)s->n * 140) * substeps_total;
	memset(s->type_ms, 0, sizeof(s->type_ms));
	memset(s->type_launches, 0, sizeof(s->type_launches));
	memset(s->type_projections, 0, sizeof(s->type_projections));
	for (DeviceSegment &d : s->dsegs) { d.ms = 0.0; d.launches = 0; }
	s->persist_ms = 0.0; s->persist_launches = 0;

	if (s->persistent_active())
	{
		HIPCHECK(hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream));
		int r = snapshot_state(s, false);
		if (r) return r;
	}
	if (s->d_contact_counters && !s->colliders.empty())
		HIPCHECK(hipMemsetAsync(s->d_contact_counters, 0, 2 * sizeof(unsigned int), s->stream));     // [1] = overflow flag, sticky for the whole call
	// One launch per ITERATION (scenes with contacts between deformable solids, persistent_iter_active): a launch can be refused (residency) or a
	// tile can time out at any iteration of any substep, after earlier iterations and contact solves of the same step have changed the state.  The
	// state is therefore saved per STEP -- the contact list is not written before the detection at the end of the step, and that detection
	// synchronises with the host anyway -- and a failed step is restored, the schedule switched off and the step repeated with one launch per segment.
	auto step_begin = [&]() -> int
	{
		if (!s->persistent_iter_active()) return PBDX_OK;
		HIPCHECK(hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream));
		return snapshot_state(s, false);
	};
	auto step_end = [&]() -> int          // before the step's contact detection
	{
		if (!s->persistent_iter_active()) return PBDX_OK;
		HIPCHECK(hipStreamSynchronize(s->stream));
		if (!s->h_error[0] && !s->h_error[1]) return PBDX_OK;
		if (s->h_error[1]) s->persist_refusals++; else s->persist_timeouts++;
		s->h_error[0] = s->h_error[1] = s->h_error[2] = 0u;
		s->persist_ok = false;                   // persistent_iter_active() is false from here on
		s->drop_graph();
		float4 *dst[4] = { s->d_pos[0], s->d_vel, s->d_old, s->d_last };
		for (int q = 0; q < 4; q++)
		{
			if (!s->d_snap[q]) { set_error("one-launch-per-iteration schedule: a launch failed and no snapshot exists"); return PBDX_ERR_HIP; }
			HIPCHECK(hipMemcpyAsync(dst[q], s->d_snap[q], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
		}
		HIPCHECK(hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream));
		for (uint32_t q = 0; q < sub_steps; q++)
		{
			int r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr);
			if (r) return r;
		}
		return PBDX_OK;
	};
	// per-substep events (PBDX_OPT_SUBSTEP_EVENTS; not in the per-launch profiling mode, at most 65 536 substeps per call)
	const bool sub_ev = s->substep_events && !s->profile && substeps_total <= 65536;
	s->substep_ms.clear();
	if (sub_ev)
		while (s->sub_events.size() < substeps_total) { hipEvent_t e = nullptr; HIPCHECK(hipEventCreate(&e)); s->sub_events.push_back(e); }
	if (s->profile)
	{
		HIPCHECK(hipEventRecord(s->ev_start, s->stream));
		for (uint64_t k = 0; k < substeps_total; k++)
		{
			if (k % sub_steps == 0) { int rb = step_begin(); if (rb) return rb; }
			ProfCursor pc; pc.s = s;
			int r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, &pc);
			if (r) return r;
			if (s->last_flips) s->swap_state();
			r = collect_profile(s, &pc);
			if (r) return r;
			if ((k + 1) % sub_steps == 0) { r = step_end(); if (!r) r = enqueue_contacts(s); if (r) return r; }
		}
		HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
	}
	else if (s->use_graph)
	{
		pbdx_solver::GraphKey k = { hs, max_iterations, vel, { gravity[0], gravity[1], gravity[2] }, s->schedule_version, s->block_size, s->xcd_remap, s->n, s->fused_active() ? 1 : 0, s->persistent_active() ? 1 : (s->persistent_iter_active() ? 2 : 0) };
		if (memcmp(&k, &s->key, sizeof(k)) != 0) { s->drop_graph(); s->key = k; }
		bool flips = false;
		auto capture = [&]() -> int
		{
			// one substep in the orientation of the position buffers that holds now
			HIPCHECK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
			int r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr);
			hipGraph_t g = nullptr;
			hipError_t e = hipStreamEndCapture(s->stream, &g);
			if (r) { if (g) (void)hipGraphDestroy(g); return r; }
			if (e != hipSuccess) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return PBDX_ERR_HIP; }
			s->graph[s->phys] = g;
			HIPCHECK(hipGraphInstantiate(&s->graph_exec[s->phys], g, nullptr, nullptr, 0));
			s->graph_valid[s->phys] = true;
			return PBDX_OK;
		};
		if (!s->graph_valid[s->phys]) { int r = capture(); if (r) return r; }
		else (void)0;
		// (last_flips is a property of the schedule, set by enqueue_substep during a capture; recompute it for a cached graph)
		flips = s->persistent_active() && max_iterations && s->n && sweep_flips(s, max_iterations) != 0;
		s->last_folded = s->persistent_active() && max_iterations && s->n;
		s->last_flips = flips;
		HIPCHECK(hipEventRecord(s->ev_start, s->stream));
		for (uint64_t k2 = 0; k2 < substeps_total; k2++)
		{
			if (k2 % sub_steps == 0) { int rb = step_begin(); if (rb) return rb; }
			if (!s->graph_valid[s->phys]) { int r = capture(); if (r) return r; }
			HIPCHECK(hipGraphLaunch(s->graph_exec[s->phys], s->stream));
			if (flips) s->swap_state();
			if (sub_ev) HIPCHECK(hipEventRecord(s->sub_events[k2], s->stream));
			if ((k2 + 1) % sub_steps == 0) { int r = step_end(); if (!r) r = enqueue_contacts(s); if (r) return r; }
		}
		HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
	}
	else
	{
		HIPCHECK(hipEventRecord(s->ev_start, s->stream));
		for (uint64_t k = 0; k < substeps_total; k++)
		{
			if (k % sub_steps == 0) { int rb = step_begin(); if (rb) return rb; }
			int r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr);
			if (r) return r;
			if (s->last_flips) s->swap_state();
			if (sub_ev) HIPCHECK(hipEventRecord(s->sub_events[k], s->stream));
			if ((k + 1) % sub_steps == 0) { r = step_end(); if (!r) r = enqueue_contacts(s); if (r) return r; }
		}
		HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
	}
	HIPCHECK(hipStreamSynchronize(s->stream));
	if (sub_ev && !s->h_error[0] && !s->h_error[1])
	{
		// device time of substep k = from the event after substep k - 1 (the call's start event for k = 0) to the event after substep k
		s->substep_ms.resize((size_t)substeps_total);
		for (uint64_t k = 0; k < substeps_total; k++)
			HIPCHECK(hipEventElapsedTime(&s->substep_ms[(size_t)k], k ? s->sub_events[(size_t)k - 1] : s->ev_start, s->sub_events[(size_t)k]));
	}
	if (s->h_error[1])
	{
		// A persistent launch refused to start (its workgroups were not all resident within kArriveLimitTicks): the
		// integrate of substep `k` ran, nothing after it did.  Complete that substep and the remaining ones with one
		// launch per segment -- the result is the one an undisturbed run produces -- and stop using the schedule.
		const uint64_t k = s->h_error[2];
		const bool folded = s->last_folded;      // then not even the integration of substep k has happened
		// substeps k .. end were no-ops, but the host swapped the buffer roles once per enqueued substep if the launch flips:
		// the state still sits where it was when substep k started
		if (s->last_flips && ((substeps_total - k) & 1u)) s->swap_state();
		s->h_error[1] = s->h_error[2] = 0u;
		s->persist_ok = false;
		s->persist_refusals++;
		s->drop_graph();
		HIPCHECK(hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream));
		int r = folded ? enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr) : enqueue_substep_tail(s, hs, inv_h, max_iterations, vel, nullptr);
		if (!r && (k + 1) % sub_steps == 0) r = enqueue_contacts(s);
		for (uint64_t k2 = k + 1; !r && k2 < substeps_total; k2++)
		{
			r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr);
			if (!r && (k2 + 1) % sub_steps == 0) r = enqueue_contacts(s);
		}
		if (r) return r;
		HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
		HIPCHECK(hipStreamSynchronize(s->stream));
	}
	{
		// a tile of a persistent launch timed out: the state was restored from the snapshot; repeat the whole call
		// with one launch per segment
		const int rec = recover_persistent(s, false);
		if (rec < 0) { set_error("persistent schedule: recovery after a timed-out launch failed"); return -rec; }
		if (rec > 0)
		{
			if (s->d_contact_counters && !s->colliders.empty())
				HIPCHECK(hipMemsetAsync(s->d_contact_counters, 0, 2 * sizeof(unsigned int), s->stream));
			int r = PBDX_OK;
			for (uint64_t k2 = 0; !r && k2 < substeps_total; k2++)
			{
				r = enqueue_substep(s, hs, inv_h, max_iterations, vel, gravity, nullptr);
				if (!r && (k2 + 1) % sub_steps == 0) r = enqueue_contacts(s);
			}
			if (r) return r;
			HIPCHECK(hipEventRecord(s->ev_stop, s->stream));
			HIPCHECK(hipStreamSynchronize(s->stream));
		}
	}
	float ms = 0.0f;
	HIPCHECK(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
	s->stats.total_ms = ms;
	if (s->tet_active())
	{
		uint32_t c[kTcWords] = { 0, 0, 0, 0 };
		HIPCHECK(hipMemcpy(c, s->d_tet_counters, sizeof(c), hipMemcpyDeviceToHost));
		if (c[kTcOverflow] || c[kTcStack])
		{
			(void)hipMemset(s->d_tet_counters, 0, kTcWords * sizeof(uint32_t));
			if (getenv("PBDX_TET_PROFILE"))
				fprintf(stderr, "[pbdx tet] detection failed: flag %u (1 node pairs, 2 barrier abandoned, 3 generations, 4 leaf pairs / chunks); generations %u, node pairs %u of %u, leaf pairs %u of %u, chunks %u of %u\n",
					c[kTcStack], c[kTcGenerations], c[kTcTreeNodes], s->tet_work.node_cap, c[kTcLeafPairs], s->tet_work.front_cap, c[kTcChunks], s->tet_work.chunk_cap);
			set_error(c[kTcOverflow] ? "contacts between solids: more than %u contacts in one step" :
				"contacts between solids: the traversal of the bounding-sphere hierarchies could not be completed (more than %u node pairs, more than 256 levels of the recursion, or a workgroup that never arrived at a barrier; serial form: stack of 128)",
				c[kTcOverflow] ? s->tet_work.max_contacts : s->tet_work.node_cap);
			return PBDX_ERR_UNSUPPORTED;
		}
	}
	if (s->d_dyn_counters && s->any_dynamic())
	{
		unsigned int c[2] = { 0, 0 };
		HIPCHECK(hipMemcpy(c, s->d_dyn_counters, sizeof(c), hipMemcpyDeviceToHost));
		if (c[1]) { set_error("contacts with dynamic rigid bodies: more than %u contacts in the sequential list; their response was skipped", s->dyn_cap); return PBDX_ERR_UNSUPPORTED; }
	}
	if (s->d_contact_counters && !s->colliders.empty())
	{
		// the reference has no per-particle contact limit: exceeding the engine's is an error, not a silent divergence
		unsigned int c[2] = { 0, 0 };
		HIPCHECK(hipMemcpy(c, s->d_contact_counters, sizeof(c), hipMemcpyDeviceToHost));
		if (c[1])
		{
			set_error("a particle had more than %d simultaneous contacts: its contact response was skipped (the reference has no such limit)", PBDX_MAX_CONTACTS_PER_PARTICLE);
			return PBDX_ERR_UNSUPPORTED;
		}
	}
	return PBDX_OK;
}

// ---- one substep in pieces (mixed models) ----------------------------------------------------------------------------------
// A host that interleaves constraints of its own with the engine's colour groups (include/pbdx.h) drives the substep itself:
// integrate, the engine's batches of a range of groups for ONE iteration, ..., velocity update.  Plain launches of the
// per-colour kernels on the state buffer, synchronous; the arithmetic is that of every other schedule.
int pbdx_solver_integrate(pbdx_solver *s, float h_sub, const float gravity[3])
{
	if (!s || !gravity || s->schedule_open) { set_error("integrate: bad arguments / schedule still open"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	if (s->n)
	{
		const uint32_t bs = 256, nb = (s->n + bs - 1) / bs;
		hipLaunchKernelGGL(integrate_kernel, dim3(nb), dim3(bs), 0, s->stream, s->d_pos[0], s->d_pos[0], s->d_vel, s->d_old, s->d_last, s->n, h_sub,
			gravity[0], gravity[1], gravity[2], (const uint32_t *)nullptr);
		HIPCHECK(hipGetLastError());
	}
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_project_groups(pbdx_solver *s, float h_sub, uint32_t iteration, uint32_t group_begin, uint32_t group_end)
{
	if (!s || s->schedule_open) { set_error("project_groups: bad state"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	int r = ensure_device_batches(s);
	if (r) return r;
	for (uint32_t bi : s->order)        // (group, type) order
	{
		const Batch &b = s->batches[bi];
		if (b.group < group_begin || b.group >= group_end) continue;
		r = launch_batch(s, b, h_sub, iteration == 0);
		if (r) return r;
	}
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_update_velocities(pbdx_solver *s, float h_sub, int velocity_update_method)
{
	if (!s || s->schedule_open || !(h_sub > 0.0f)) { set_error("update_velocities: bad arguments"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	if (s->n)
	{
		const uint32_t bs = 256, nb = (s->n + bs - 1) / bs;
		const float inv_h = (float)(1.0 / (double)h_sub);          // TimeIntegration.cpp:50 evaluates 1.0/h in double
		hipLaunchKernelGGL(velocity_kernel, dim3(nb), dim3(bs), 0, s->stream, s->d_pos[0], s->d_vel, s->d_old, s->d_last, s->n, inv_h, velocity_update_method != 0, (uint32_t *)nullptr);
		HIPCHECK(hipGetLastError());
	}
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_project(pbdx_solver *s, float h_sub, uint32_t iterations)
{
	if (!s || s->schedule_open) { set_error("project: bad state"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	const bool fresh_plan = !s->plan_built;
	int r = ensure_plan(s);
	if (!r && (fresh_plan || s->autotuned_for_tet != s->tet_active())) r = autotune_schedule(s);
	if (!r) r = ensure_trace(s);
	if (!r && !s->fused_active()) r = ensure_device_batches(s);
	if (r) return r;
	const int start = (int)sweep_flips(s, iterations);
	if (start)
		HIPCHECK(hipMemcpyAsync(s->d_pos[1], s->d_pos[0], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
	if (s->persistent_active())
	{
		HIPCHECK(hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream));
		r = snapshot_state(s, true);
		if (r) return r;
	}
	r = projection_sweeps(s, h_sub, iterations, start, nullptr);
	if (r) return r;
	HIPCHECK(hipStreamSynchronize(s->stream));
	if (s->h_error[1])
	{
		// the persistent launch refused to start and modified nothing: sweep with one launch per segment instead
		s->h_error[1] = s->h_error[2] = 0u;
		s->persist_ok = false;
		s->persist_refusals++;
		HIPCHECK(hipMemsetAsync(s->d_ctl, 0, kCtlWords * sizeof(uint32_t), s->stream));
		r = projection_sweeps(s, h_sub, iterations, start, nullptr);
		if (r) return r;
		HIPCHECK(hipStreamSynchronize(s->stream));
	}
	const int rec = recover_persistent(s, true);
	if (rec < 0) { set_error("persistent schedule: recovery after a timed-out launch failed"); return -rec; }
	if (rec > 0)
	{
		if (start)
			HIPCHECK(hipMemcpyAsync(s->d_pos[1], s->d_pos[0], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
		r = projection_sweeps(s, h_sub, iterations, start, nullptr);
		if (r) return r;
		HIPCHECK(hipStreamSynchronize(s->stream));
	}
	return PBDX_OK;
}

// A caller's checkpoint of the particle state on the device (pos, vel, old, last: four stream-ordered device-to-device copies, no host
// synchronisation) and its restoration.  The reference-side plug-in steps SPECULATIVELY while its exact parameter scan runs on the host's worker
// threads; if the scan finds an edit the step is undone with this and repeated on the refreshed parameter streams.
int pbdx_solver_save_state(pbdx_solver *s)
{
	if (!s) return PBDX_ERR_INVALID;
	ENTER_DEVICE(s->device);
	s->ckpt_n = 0;
	if (!s->n) return PBDX_OK;
	float4 *src[4] = { s->d_pos[0], s->d_vel, s->d_old, s->d_last };
	for (int k = 0; k < 4; k++)
	{
		if (!s->d_ckpt[k]) HIPCHECK(hipMalloc(&s->d_ckpt[k], (size_t)s->n * sizeof(float4)));
		HIPCHECK(hipMemcpyAsync(s->d_ckpt[k], src[k], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
	}
	s->ckpt_n = s->n;
	return PBDX_OK;
}
int pbdx_solver_restore_state(pbdx_solver *s)
{
	if (!s) return PBDX_ERR_INVALID;
	ENTER_DEVICE(s->device);
	if (!s->ckpt_n || s->ckpt_n != s->n) { set_error("restore_state: no checkpoint of the current particle set (pbdx_solver_save_state)"); return PBDX_ERR_INVALID; }
	float4 *dst[4] = { s->d_pos[0], s->d_vel, s->d_old, s->d_last };
	for (int k = 0; k < 4; k++)
		HIPCHECK(hipMemcpyAsync(dst[k], s->d_ckpt[k], (size_t)s->n * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
	return PBDX_OK;
}

int pbdx_solver_synchronize(pbdx_solver *s)
{
	if (!s) return PBDX_ERR_INVALID;
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	return PBDX_OK;
}

int pbdx_solver_set_colliders(pbdx_solver *s, uint32_t n, const pbdx_collider *colliders)
{
	if (!s || (n && !colliders)) { set_error("set_colliders: bad arguments"); return PBDX_ERR_INVALID; }
	for (uint32_t i = 0; i < n; i++)
		if (colliders[i].shape < PBDX_SHAPE_BOX || colliders[i].shape > PBDX_SHAPE_HOLLOW_BOX) { set_error("set_colliders: unknown shape %d", colliders[i].shape); return PBDX_ERR_UNSUPPORTED; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	// (the same number of colliders again -- a host that moves its bodies every step -- keeps the allocation)
	if (s->d_colliders && n != s->colliders.size()) { (void)hipFree(s->d_colliders); s->d_colliders = nullptr; }
	s->colliders.assign(colliders, colliders + n);
	if (n)
	{
		if (!s->d_colliders) HIPCHECK(hipMalloc(&s->d_colliders, (size_t)n * sizeof(pbdx_collider)));
		HIPCHECK(pbdx::copy_to_device(s->d_colliders, colliders, (size_t)n * sizeof(pbdx_collider)));
	}
	if (!s->d_contact_counters)
	{
		HIPCHECK(hipMalloc(&s->d_contact_counters, 2 * sizeof(unsigned int)));
		HIPCHECK(hipMemset(s->d_contact_counters, 0, 2 * sizeof(unsigned int)));
	}
	return PBDX_OK;
}

int pbdx_solver_set_collision_ranges(pbdx_solver *s, uint32_t n, const pbdx_collision_range *ranges)
{
	if (!s || (n && !ranges)) { set_error("set_collision_ranges: bad arguments"); return PBDX_ERR_INVALID; }
	for (uint32_t i = 0; i < n; i++)
		if ((uint64_t)ranges[i].first + ranges[i].count > s->n) { set_error("set_collision_ranges: range %u exceeds the %u uploaded particles", i, s->n); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->ranges.assign(ranges, ranges + n);
	if (s->d_ranges) { (void)hipFree(s->d_ranges); s->d_ranges = nullptr; }
	if (n)
	{
		HIPCHECK(hipMalloc(&s->d_ranges, (size_t)n * sizeof(pbdx_collision_range)));
		HIPCHECK(pbdx::copy_to_device(s->d_ranges, ranges, (size_t)n * sizeof(pbdx_collision_range)));
	}
	return PBDX_OK;
}

int pbdx_solver_set_contact_params(pbdx_solver *s, float tolerance, float contact_stiffness, uint32_t max_iterations_v)
{
	if (!s) return PBDX_ERR_INVALID;
	s->contact_tolerance = tolerance; s->contact_stiffness = contact_stiffness; s->max_iterations_v = max_iterations_v;
	return PBDX_OK;
}

int pbdx_solver_set_rest_positions(pbdx_solver *s, uint32_t n, const float *x0)
{
	if (!s || !x0 || n != s->n || !n) { set_error("set_rest_positions: particle count mismatch (upload the particles first)"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	{ int rs = ensure_stage(s, sizeof(float)); if (rs) return rs; }
	if (!s->d_rest) HIPCHECK(hipMalloc(&s->d_rest, (size_t)n * sizeof(float4)));
	HIPCHECK(pbdx::copy_to_device(s->d_stage, x0, (size_t)3 * n * sizeof(float)));
	hipLaunchKernelGGL(pack_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s->stream, (const float *)s->d_stage, (const float *)nullptr, s->d_rest, n);
	HIPCHECK(hipGetLastError());
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->rest_set = true;
	return PBDX_OK;
}

static int set_tet_colliders_impl(pbdx_solver *s, uint32_t n, const pbdx_tet_collider *colliders, float tolerance);
int pbdx_solver_set_tet_colliders(pbdx_solver *s, uint32_t n, const pbdx_tet_collider *colliders, float tolerance)
{
	const int r = set_tet_colliders_impl(s, n, colliders, tolerance);
	if (r && s) s->free_tet_colliders();          // nothing half-uploaded stays active
	return r;
}
static int set_tet_colliders_impl(pbdx_solver *s, uint32_t n, const pbdx_tet_collider *colliders, float tolerance)
{
	if (!s || (n && !colliders) || n > 256) { set_error("set_tet_colliders: bad arguments (at most 256 colliders)"); return PBDX_ERR_INVALID; }
	{ const int rv = validate_tet_colliders(n, colliders, s->n); if (rv) return rv; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->free_tet_colliders();
	s->drop_graph();
	if (!n) return PBDX_OK;
	auto up = [](auto **dst, const auto *src, size_t count) -> hipError_t
	{
		hipError_t e = hipMalloc(dst, count * sizeof(**dst));
		if (e == hipSuccess) e = pbdx::copy_to_device(*dst, src, count * sizeof(**dst));
		return e;
	};
	s->tet_dev.resize(n);
	s->tet_views.resize(n);
	for (uint32_t i = 0; i < n; i++)
	{
		const pbdx_tet_collider &c = colliders[i];
		pbdx_solver::DevTetCollider &d = s->tet_dev[i];
		TetColliderView &v = s->tet_views[i];
		memset(&v, 0, sizeof(v));
		v.sdf.shape = c.shape; v.sdf.invert = c.invert; memcpy(v.sdf.params, c.params, sizeof(c.params));
		v.first = c.first_particle; v.num_vertices = c.num_vertices; v.num_tets = c.num_tets;
		memcpy(v.X0, c.initial_x, sizeof(v.X0)); memcpy(v.R0, c.initial_R, sizeof(v.R0));
		v.tolerance = tolerance; v.test_mesh = c.test_mesh; v.body_index = c.body_index;
		HIPCHECK(up(&d.tets, c.tets, (size_t)4 * c.num_tets));
		v.tets = d.tets;
		const pbdx_bvh *src[3] = { &c.points, &c.tets_bvh, &c.tets_rest };
		pbdx_solver::DevBvh *dst[3] = { &d.points, &d.tet_bvh, &d.tet_bvh0 };
		BvhView *view[3] = { &v.points, &v.tet_bvh, &v.tet_bvh0 };
		for (int q = 0; q < 3; q++)
		{
			HIPCHECK(up(&dst[q]->lst, src[q]->entities, src[q]->num_entities));
			HIPCHECK(up(&dst[q]->nodes, src[q]->nodes, (size_t)4 * src[q]->num_nodes));
			HIPCHECK(hipMalloc(&dst[q]->hulls, (size_t)src[q]->num_nodes * sizeof(P4)));
			if (q == 2) HIPCHECK(pbdx::copy_to_device(dst[q]->hulls, src[q]->hulls, (size_t)src[q]->num_nodes * sizeof(P4)));
			else HIPCHECK(hipMemset(dst[q]->hulls, 0, (size_t)src[q]->num_nodes * sizeof(P4)));
			dst[q]->num_nodes = src[q]->num_nodes;
			*view[q] = BvhView{ dst[q]->lst, dst[q]->nodes, dst[q]->hulls, dst[q]->num_nodes, nullptr, nullptr, nullptr, 0, 0 };
			if (q < 2)
			{
				// the entities' vertices in list order (static), and room for their positions (gathered every step)
				const uint32_t per = q == 0 ? 1u : 4u;
				std::vector<uint32_t> flat((size_t)per * src[q]->num_entities);
				for (uint32_t e = 0; e < src[q]->num_entities; e++)
					for (uint32_t k = 0; k < per; k++) flat[(size_t)per * e + k] = q == 0 ? src[q]->entities[e] : c.tets[4 * src[q]->entities[e] + k];
				HIPCHECK(up(&dst[q]->flat, flat.data(), flat.size()));
				HIPCHECK(hipMalloc(&dst[q]->gathered, flat.size() * sizeof(P4)));
				HIPCHECK(hipMalloc(&dst[q]->soa, 3 * flat.size() * sizeof(float)));
				view[q]->flat = dst[q]->flat; view[q]->gathered = dst[q]->gathered; view[q]->soa = dst[q]->soa; view[q]->per_entity = per; view[q]->num_elements = (uint32_t)flat.size();
			}
		}
	}
	{
		// the nodes whose chains are long, longest first
		struct Big { uint32_t m, which, node; };
		std::vector<Big> big;
		for (uint32_t i = 0; i < n; i++)
			for (int q = 0; q < 2; q++)
			{
				const pbdx_bvh &b = q ? colliders[i].tets_bvh : colliders[i].points;
				for (uint32_t nd = 0; nd < b.num_nodes; nd++)
				{
					const uint32_t m = (uint32_t)b.nodes[4 * nd + 3] * (q ? 4u : 1u);
					if (m >= kTcBigNode) big.push_back(Big{ m, 2 * i + (uint32_t)q, nd });
				}
			}
		std::sort(big.begin(), big.end(), [](const Big &a, const Big &b) { return a.m > b.m; });
		std::vector<uint32_t> flat, slices;
		for (size_t bi = 0; bi < big.size(); bi++)
		{
			flat.push_back(big[bi].which); flat.push_back(big[bi].node);
			for (uint32_t sl = 0; sl * kTcRadiusSlice < big[bi].m; sl++) { slices.push_back((uint32_t)bi); slices.push_back(sl); }
		}
		s->tet_big_count = (uint32_t)big.size();
		s->tet_big_slices = (uint32_t)(slices.size() / 2);
		if (!flat.empty())
		{
			HIPCHECK(up(&s->d_tet_big, flat.data(), flat.size()));
			HIPCHECK(up(&s->d_tet_big_slices, slices.data(), slices.size()));
			HIPCHECK(hipMalloc(&s->d_tet_big_r2, big.size() * sizeof(uint32_t)));
		}
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(tet_hull_kernel2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTcBigNodeLds);
	}
	HIPCHECK(hipMalloc(&s->d_tet_views, (size_t)n * sizeof(TetColliderView)));
	HIPCHECK(pbdx::copy_to_device(s->d_tet_views, s->tet_views.data(), (size_t)n * sizeof(TetColliderView)));
	HIPCHECK(hipMalloc(&s->d_tet_aabb, (size_t)6 * n * sizeof(float)));
	if (!s->d_tet_counters) HIPCHECK(hipMalloc(&s->d_tet_counters, kTcWords * sizeof(uint32_t)));
	HIPCHECK(hipMemset(s->d_tet_counters, 0, kTcWords * sizeof(uint32_t)));      // no contacts before the first detection
	{
		// sized by the scene to begin with: the 2 x 70875-tet scene of the tests walks 0.71 M node pairs for 14 k overlapping leaf pairs
		// (4.3 per tet) -- and 50 M when the bars are pushed far into each other: the scratch grows on demand (enqueue_tet_detection)
		uint64_t total_tets = 0;
		for (uint32_t i = 0; i < n; i++) total_tets += colliders[i].num_tets;
		s->tet_num_colliders = n;
		// (PBDX_TET_SCRATCH="nodes,contacts": developer aid, lets a test start from capacities that must grow)
		uint64_t nodes0 = std::min<uint64_t>((1ull << 22) + 32ull * total_tets, 1ull << 27);
		uint32_t contacts0 = kTetContactsAtFirst;
		if (const char *e = getenv("PBDX_TET_SCRATCH"))
		{
			unsigned long long a = 0; unsigned b = 0;
			if (sscanf(e, "%llu,%u", &a, &b) == 2 && a >= 64 && b >= 1) { nodes0 = a; contacts0 = b; }
		}
		s->tet_grown = 0;
		int r = alloc_tet_work(s, nodes0, contacts0);
		if (r) return r;
	}
	return PBDX_OK;
}

int pbdx_debug_tet_capacity(pbdx_solver *s, uint32_t out[4])
{
	if (!s || !out) return PBDX_ERR_INVALID;
	out[0] = s->tet_work.node_cap; out[1] = s->tet_work.front_cap; out[2] = s->tet_work.max_contacts; out[3] = s->tet_grown;
	return PBDX_OK;
}

int pbdx_debug_tet_counters(pbdx_solver *s, uint32_t out[8])
{
	if (!s || !out) return PBDX_ERR_INVALID;
	memset(out, 0, 8 * sizeof(uint32_t));
	if (!s->d_tet_counters) return PBDX_OK;
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	HIPCHECK(hipMemcpy(out, s->d_tet_counters, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
	return PBDX_OK;
}

int pbdx_debug_tet_impulses(pbdx_solver *s, uint32_t *last, uint64_t *total)
{
	if (!s) return PBDX_ERR_INVALID;
	if (last) *last = s->tet_impulses_last;
	if (total) *total = s->tet_impulses_total;
	return PBDX_OK;
}

int pbdx_debug_tet_hulls(pbdx_solver *s, uint32_t collider, int which, uint32_t capacity, uint32_t *count, float *out)
{
	if (!s || !count || collider >= s->tet_dev.size() || which < 0 || which > 2) return PBDX_ERR_INVALID;
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	const pbdx_solver::DevBvh &b = which == 0 ? s->tet_dev[collider].points : which == 1 ? s->tet_dev[collider].tet_bvh : s->tet_dev[collider].tet_bvh0;
	*count = b.num_nodes;
	if (out && capacity) HIPCHECK(pbdx::copy_from_device(out, b.hulls, (size_t)std::min(capacity, b.num_nodes) * sizeof(P4)));
	return PBDX_OK;
}

int pbdx_solver_get_tet_contacts(pbdx_solver *s, uint32_t capacity, uint32_t *count, float *out)
{
	if (!s || !count) return PBDX_ERR_INVALID;
	*count = 0;
	if (!s->tet_active() || !s->d_tet_counters) return PBDX_OK;
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	uint32_t c[kTcWords];
	HIPCHECK(hipMemcpy(c, s->d_tet_counters, sizeof(c), hipMemcpyDeviceToHost));
	*count = c[kTcCount];
	const uint32_t m = std::min(capacity, c[kTcCount]);
	if (m && out)
	{
		std::vector<TetContact> tmp(m);
		HIPCHECK(pbdx::copy_from_device(tmp.data(), s->d_tet_contacts, (size_t)m * sizeof(TetContact)));
		for (uint32_t i = 0; i < m; i++) contact_to_floats(tmp[i], out + (size_t)i * PBDX_TET_CONTACT_FLOATS);
	}
	return PBDX_OK;
}

int pbdx_solver_get_num_contacts(pbdx_solver *s, uint32_t *out)
{
	if (!s || !out) return PBDX_ERR_INVALID;
	*out = 0;
	if (!s->d_contact_counters) return PBDX_OK;
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	unsigned int c[2] = { 0, 0 };
	HIPCHECK(hipMemcpy(c, s->d_contact_counters, sizeof(c), hipMemcpyDeviceToHost));
	*out = c[0];
	if (c[1]) { set_error("a particle had more than %d simultaneous contacts", PBDX_MAX_CONTACTS_PER_PARTICLE); return PBDX_ERR_INVALID; }
	return PBDX_OK;
}

int pbdx_solver_set_collider_dynamics(pbdx_solver *s, uint32_t n, const pbdx_collider_dynamics *dyn)
{
	if (!s || (n && !dyn)) { set_error("set_collider_dynamics: null argument"); return PBDX_ERR_INVALID; }
	if (n && n != s->colliders.size()) { set_error("set_collider_dynamics: %u records for %zu colliders", n, s->colliders.size()); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->dynamics.assign(dyn, dyn + n);
	if (s->d_dynamics) { (void)hipFree(s->d_dynamics); s->d_dynamics = nullptr; }
	if (s->d_collider_pos) { (void)hipFree(s->d_collider_pos); s->d_collider_pos = nullptr; }
	if (!n) return PBDX_OK;
	// a collider's place among the colliders ordered by the object index of their bodies
	std::vector<uint8_t> pos(n, 0);
	for (uint32_t k = 0; k < n; k++) for (uint32_t q = 0; q < n; q++) if (dyn[q].object_index < dyn[k].object_index || (dyn[q].object_index == dyn[k].object_index && q < k)) pos[k]++;
	HIPCHECK(hipMalloc(&s->d_dynamics, (size_t)n * sizeof(pbdx_collider_dynamics)));
	HIPCHECK(pbdx::copy_to_device(s->d_dynamics, dyn, (size_t)n * sizeof(pbdx_collider_dynamics)));
	HIPCHECK(hipMalloc(&s->d_collider_pos, std::max<size_t>(n, 16)));
	HIPCHECK(pbdx::copy_to_device(s->d_collider_pos, pos.data(), n));
	return PBDX_OK;
}

int pbdx_solver_set_contact_order(pbdx_solver *s, uint32_t n_ranges, const uint32_t *range_object_index, uint32_t n_particles, const uint32_t *rank)
{
	if (!s || (n_ranges && !range_object_index) || (n_particles && !rank)) { set_error("set_contact_order: null argument"); return PBDX_ERR_INVALID; }
	if (n_ranges != s->ranges.size() || n_particles != s->n) { set_error("set_contact_order: %u ranges / %u particles, the engine holds %zu / %u", n_ranges, n_particles, s->ranges.size(), s->n); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	s->range_object.assign(range_object_index, range_object_index + n_ranges);
	if (s->d_rank) { (void)hipFree(s->d_rank); s->d_rank = nullptr; }
	s->rank_count = 0;
	if (!n_particles) return PBDX_OK;
	HIPCHECK(hipMalloc(&s->d_rank, (size_t)n_particles * 4));
	HIPCHECK(pbdx::copy_to_device(s->d_rank, rank, (size_t)n_particles * 4));
	s->rank_count = n_particles;
	return PBDX_OK;
}

int pbdx_solver_get_body_velocities(pbdx_solver *s, uint32_t n, float *v, float *omega)
{
	if (!s || !v || !omega) { set_error("get_body_velocities: null argument"); return PBDX_ERR_INVALID; }
	if (n != s->colliders.size()) { set_error("get_body_velocities: %u bodies asked for, %zu colliders", n, s->colliders.size()); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	std::vector<float> b((size_t)kMaxDynColliders * 8, 0.0f);
	const bool solved = s->d_dyn_body && s->dynamics.size() == s->colliders.size() && s->any_dynamic();
	if (solved) HIPCHECK(pbdx::copy_from_device(b.data(), s->d_dyn_body, b.size() * sizeof(float)));
	for (uint32_t k = 0; k < n; k++)
		for (int q = 0; q < 3; q++)
		{
			v[3 * k + q] = solved ? b[8 * k + q] : s->colliders[k].body_v[q];
			omega[3 * k + q] = solved ? b[8 * k + 3 + q] : s->colliders[k].body_omega[q];
		}
	return PBDX_OK;
}

int pbdx_solver_get_lambdas(pbdx_solver *s, uint32_t batch_index, uint32_t count, float *out)
{
	if (!s || !out || batch_index >= s->batches.size()) { set_error("get_lambdas: bad batch"); return PBDX_ERR_INVALID; }
	const Batch &b = s->batches[batch_index];
	if (!type_info(b.type)->xpbd || count != b.count) { set_error("get_lambdas: batch has no multipliers or count mismatch"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	if (!s->fused_active())
	{
		if (!b.d_lambda) { memset(out, 0, (size_t)count * sizeof(float)); return PBDX_OK; }     // schedule never ran
		HIPCHECK(pbdx::copy_from_device(out, b.d_lambda, (size_t)count * sizeof(float)));
		return PBDX_OK;
	}
	// fused schedule: the multiplier of a constraint lives in the lambda stream of every tile that
	// executes it (identical values); fetch the first copy
	uint32_t pos = 0;
	for (size_t oi = 0; oi < s->order.size(); oi++) if (s->order[oi] == batch_index) pos = (uint32_t)oi;
	const uint32_t cid0 = s->plan.batch_base[pos];
	std::vector<uint8_t> have(count, 0);
	for (size_t si = 0; si < s->plan.segs.size(); si++)
	{
		const FusedSegment &seg = s->plan.segs[si];
		if (!seg.lam_count) continue;
		std::vector<float> lam(seg.lam_count);
		bool fetched = false;
		for (const FusedStep &st : seg.steps)
		{
			if (!type_info((int)st.type)->xpbd) continue;
			for (uint32_t q = 0; q < st.count; q++)
			{
				const uint32_t cid = seg.slot_cid[st.cid_off + q];
				if (cid < cid0 || cid >= cid0 + count || have[cid - cid0]) continue;
				if (!fetched)
				{
					HIPCHECK(pbdx::copy_from_device(lam.data(), s->dsegs[si].d_lambda, (size_t)seg.lam_count * sizeof(float)));
					fetched = true;
				}
				out[cid - cid0] = lam[st.lam_off + q];
				have[cid - cid0] = 1;
			}
		}
	}
	return PBDX_OK;
}

int pbdx_solver_get_stats(pbdx_solver *s, pbdx_step_stats *out)
{
	if (!s || !out) return PBDX_ERR_INVALID;
	*out = s->stats;
	return PBDX_OK;
}

int pbdx_solver_get_substep_times(pbdx_solver *s, float *out_ms, uint32_t capacity, uint32_t *count)
{
	if (!s || !count) return PBDX_ERR_INVALID;
	*count = (uint32_t)s->substep_ms.size();
	if (out_ms) for (uint32_t k = 0; k < capacity && k < *count; k++) out_ms[k] = s->substep_ms[k];
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

int pbdx_solver_get_plan_info(pbdx_solver *s, pbdx_plan_info *out)
{
	if (!s || !out) return PBDX_ERR_INVALID;
	memset(out, 0, sizeof(*out));
	out->active = s->fused_active() ? 1 : 0;
	out->built = s->plan_built ? 1 : 0;
	if (!s->plan_ok) return PBDX_OK;
	out->num_segments = (uint32_t)s->plan.segs.size();
	out->num_tiles = s->plan.num_tiles;
	out->num_colours = s->plan.num_colours;
	out->redundancy = s->plan.redundancy;
	out->build_seconds = s->plan.build_seconds;
	for (const FusedSegment &seg : s->plan.segs)
	{
		out->max_local = std::max(out->max_local, seg.max_local);
		out->stream_bytes_per_sweep += seg.stream_bytes;
		out->slots_per_sweep += seg.slots;
	}
	// what one sweep has to stream if no constraint were executed twice: every distinct constraint's record once
	for (const Batch &b : s->batches)
	{
		const TypeInfo *ti = type_info(b.type);
		const uint32_t rec = (ti->num_bodies == 2 ? 4u : 8u) + (uint32_t)num_planes(b.type, s->plan.views[b.type].compact != 0) * 4u + (ti->xpbd ? 8u : 0u);
		out->compulsory_stream_bytes_per_sweep += (uint64_t)b.count * rec;
	}
	return PBDX_OK;
}

int pbdx_solver_get_segment_info(pbdx_solver *s, uint32_t segment, pbdx_segment_info *out)
{
	if (!s || !out || !s->plan_ok || segment >= s->plan.segs.size()) { set_error("get_segment_info: no such segment"); return PBDX_ERR_INVALID; }
	const FusedSegment &seg = s->plan.segs[segment];
	const DeviceSegment &d = s->dsegs[segment];
	memset(out, 0, sizeof(*out));
	out->colour_begin = seg.colour_begin; out->colour_end = seg.colour_end;
	out->num_tiles = d.num_tiles; out->block = (uint32_t)d.block; out->lds_bytes = d.lds_bytes; out->type_mask = seg.type_mask;
	out->constraints = seg.constraints; out->slots = seg.slots; out->stream_bytes = seg.stream_bytes;
	out->algorithmic_bytes = d.algorithmic_bytes;
	out->profiled_ms = d.ms; out->profiled_launches = d.launches;
	return PBDX_OK;
}

int pbdx_solver_get_persistent_info(pbdx_solver *s, pbdx_persistent_info *out)
{
	if (!s || !out) return PBDX_ERR_INVALID;
	memset(out, 0, sizeof(*out));
	out->eligible = s->persist_ok ? 1 : 0;
	out->active = s->persistent_active() ? 1 : (s->persistent_iter_active() ? 2 : 0);      // 2: one launch per ITERATION (contacts between deformable solids are solved between the iterations)
	out->grid = s->persist_grid; out->block = (uint32_t)s->persist_block; out->lds_bytes = s->persist_lds;
	out->refusals = s->persist_refusals;
	out->timeouts = s->persist_timeouts;
	out->last_folded = s->last_folded ? 1 : 0;
	out->autotune_fused_ms = s->autotune_ms[1];
	out->autotune_persistent_ms = s->autotune_ms[2];
	out->profiled_ms = s->persist_ms;
	out->profiled_launches = s->persist_launches;
	for (const DeviceSegment &d : s->dsegs) out->algorithmic_bytes_per_sweep += d.algorithmic_bytes;
	return PBDX_OK;
}

int pbdx_solver_get_trace(pbdx_solver *s, uint32_t segment, uint64_t *out, uint32_t capacity, uint32_t *stride)
{
	if (!s || !out || !s->plan_ok || segment >= s->dsegs.size() || !s->dsegs[segment].d_trace) { set_error("get_trace: no trace for this segment (set PBDX_OPT_TRACE and step once)"); return PBDX_ERR_INVALID; }
	const DeviceSegment &d = s->dsegs[segment];
	const size_t need = (size_t)d.num_tiles * kTraceStride;
	if (capacity < need) { set_error("get_trace: need room for %zu stamps", need); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(s->device);
	HIPCHECK(hipStreamSynchronize(s->stream));
	HIPCHECK(pbdx::copy_from_device(out, d.d_trace, need * sizeof(uint64_t)));
	if (stride) *stride = kTraceStride;
	return PBDX_OK;
}

// Record of the range checks of a PBDX_BOUNDS build on `device` (after a device synchronisation): out[0] violations since the last reset, out[1..6] the
// first one (kind, workgroup, thread, index, limit, tile / aux), out[7] = 1 if this library IS such a build (0: the product build, which checks nothing).
int pbdx_debug_bounds_report(int device, uint32_t out[8], int reset)
{
	if (!out) return PBDX_ERR_INVALID;
	memset(out, 0, 8 * sizeof(uint32_t));
#if PBDX_BOUNDS
	ENTER_DEVICE(device);
	HIPCHECK(hipDeviceSynchronize());
	if (sweep_bounds_report(out, reset)) { set_error("pbdx_debug_bounds_report: reading the record failed"); return PBDX_ERR_HIP; }
#else
	(void)device; (void)reset;
#endif
	return PBDX_OK;
}

int pbdx_solver_describe(pbdx_solver *s, char *buf, size_t n)
{
	if (!s || !buf || !n) return PBDX_ERR_INVALID;
	uint64_t nc = 0; uint32_t ng = 0, last = 0xffffffffu;
	for (uint32_t bi : s->order) { nc += s->batches[bi].count; if (s->batches[bi].group != last) { ng++; last = s->batches[bi].group; } }
	int w = snprintf(buf, n, "device=%d name=%s arch=%s CUs=%d particles=%u constraints=%llu groups=%u batches=%zu graph=%d block=%d xcd_remap=%d",
		s->device, s->prop.name, s->prop.gcnArchName, s->prop.multiProcessorCount, s->n, (unsigned long long)nc, ng, s->batches.size(),
		s->use_graph, s->block_size, s->xcd_remap);
	if (w > 0 && (size_t)w < n)
	{
		if (s->fused_active())
		{
			uint32_t ml = 0;
			for (const FusedSegment &seg : s->plan.segs) ml = std::max(ml, seg.max_local);
			int w2 = snprintf(buf + w, n - w, " schedule=%s segments=%zu tiles=%u redundancy=%.3f max_tile_particles=%u plan_s=%.2f%s",
				s->persistent_active() ? "fused-persistent" : (s->persistent_iter_active() ? "fused-persistent-per-iteration" : "fused"), s->plan.segs.size(), s->plan.num_tiles, s->plan.redundancy, ml, s->plan.build_seconds,
				s->plan_instanced ? " (one instance planned, replicated)" : "");
			if (s->persist_refusals && w2 > 0 && (size_t)(w + w2) < n)
				w2 += snprintf(buf + w + w2, n - w - w2, " persistent_refusals=%u", s->persist_refusals);
			if ((s->autotune_ms[1] > 0.0f || s->autotune_ms[2] > 0.0f) && w2 > 0 && (size_t)(w + w2) < n)
				snprintf(buf + w + w2, n - w - w2, " autotune(per-colour %.3f ms, fused %.3f ms, persistent %.3f ms; %d rounds, %.1f ms)", s->autotune_ms[0], s->autotune_ms[1], s->autotune_ms[2],
					s->autotune_rounds, s->autotune_spent_ms);
		}
		else
		{
			int w2 = snprintf(buf + w, n - w, " schedule=per-colour%s%s", s->plan_built && !s->plan_ok ? " plan_failed=" : "", s->plan_built && !s->plan_ok ? s->plan_why.c_str() : "");
			if (s->autotune_ms[1] > 0.0f && w2 > 0 && (size_t)(w + w2) < n)
				snprintf(buf + w + w2, n - w - w2, " autotune(per-colour %.3f ms, fused %.3f ms; %d rounds, %.1f ms)", s->autotune_ms[0], s->autotune_ms[1], s->autotune_rounds, s->autotune_spent_ms);
		}
	}
	return PBDX_OK;
}

} // extern "C"
