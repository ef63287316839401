// pbdx_tetcontact_dev.h -- the parallel form of the contacts between deformable solids (device only; arithmetic in pbdx_tetcontact.h).
//
// What is sequential in the reference and what that means here (DistanceFieldCollisionDetection.cpp:160-177, 361-483;
// BoundingSphereHierarchy.cpp:124-174; TimeStepController.cpp:288-291):
//   * the contact list is appended to in the order in which ONE recursive dual-hierarchy traversal reaches the leaf pairs, and
//   * that list is solved one contact after the other inside every iteration of the next step (Gauss-Seidel): the order is part of
//     the result.
// Neither needs one thread:
//   detection  The recursion tree of BVHTest::traverse is built generation by generation (tet_traverse_kernel): every node pair of a
//              generation is tested in parallel and appends its two children (children[0] first) to the next generation.  The order
//              in which the reference's depth-first recursion reaches the overlapping leaf pairs is the left-to-right order of the
//              leaves of that tree: a bottom-up pass counts the leaf pairs below every node, a top-down pass turns the counts into
//              every leaf pair's position (children[0]'s subtree before children[1]'s) -- no thread ever walks the tree in order.
//              The candidates of a leaf pair (points x tets, row-major as the reference's two loops) are then evaluated by the whole
//              GPU, 64 candidates per wavefront; a ballot per wavefront, a scan over the ballots' populations and a second
//              evaluation of the (few) hits put the contacts where the reference's push_back would have
//              (tet_candidates_kernel<false>, tet_chunk_scan_kernel, tet_candidates_kernel<true>).
//   solve      Two contacts commute bit for bit unless they share a particle (the solve writes the contact's particle and the four
//              vertices of its tet, and reads nothing else that changes).  The list is levelled once per detection: a contact's
//              level is one more than the highest level among EARLIER contacts it shares a particle with (tet_levels_kernel: rounds
//              of "who is the first unscheduled contact at each particle").  Every level is a set of contacts on disjoint particles;
//              the levels run in order, each in parallel (tet_contact_solve_levels_kernel).
//   spheres    KDTree::update recomputes every node's sphere from ITS OWN entity range: the centre is a float sum in list order -- a
//              dependent chain of n additions per component, 283500 for the root of a 70875-tet hierarchy, and the critical path
//              of the whole detection.  The vertices are gathered into list order first (tet_gather_kernel), so a node reads a
//              contiguous range; one wavefront per node and component, whose lanes hold the values in their own registers and take
//              turns at the running sum (tet_hull_kernel2).  The radius is a maximum (order-free, parallel).
#ifndef PBDX_TETCONTACT_DEV_H
#define PBDX_TETCONTACT_DEV_H

#include "pbdx_tetcontact.h"
#include <hip/hip_runtime.h>

namespace pbdx {

constexpr uint32_t kTetContactsAtFirst = 1u << 16;      // capacities grow on demand (enqueue_tet_detection / alloc_tet_work in pbdx_solver.hip)
enum { kTcCount = 0, kTcOverflow = 1, kTcStack = 2, kTcLeafPairs = 3, kTcChunks = 4, kTcLevels = 5, kTcGenerations = 6, kTcTreeNodes = 7,
	kTcImpulses = 8 /* contacts of the list that carry a non-zero velocity impulse */, kTcWords = 10 };

struct TetWork                    // device scratch of the detection; *_cap are capacities in elements
{
	uint32_t *leaf_pairs;         // the overlapping leaf pairs in the reference's visiting order, 3 words each: collider pair, node of the
	                              // point hierarchy, node of the tet hierarchy
	uint32_t front_cap;
	// recursion tree of the traversal, generation after generation.  Written and read by different workgroups (= CUs, XCDs) of ONE launch:
	// every access is a relaxed agent-scope atomic (an sc1 load / store: served past the CU's L1, written through), 8 bytes at most
	unsigned long long *node_rec; // 2 per node: collider pair | node of the point hierarchy << 32, node of the tet hierarchy
	unsigned long long *node_cnt; // overlapping leaf pairs | candidate chunks << 32 in the node's subtree
	uint32_t *node_child;         // index of children[0]'s node (children[1]'s follows), 0xffffffff: none
	unsigned long long *node_off; // position of the subtree's first leaf pair | first chunk << 32
	uint32_t node_cap;
	uint32_t *trav;               // kTrWords words of the traversal launch: barrier, flags, size of every generation (zeroed before the launch)
	uint32_t *pair_ik;            // 2 words per ordered collider pair (i, k) whose boxes intersect
	uint32_t *chunk_off;          // per leaf pair: its first 64-candidate chunk (front_cap + 1 entries)
	uint32_t *chunk_pair;         // per chunk: its leaf pair
	unsigned long long *chunk_mask; // per chunk: which of its candidates are contacts
	uint32_t *chunk_base;         // per chunk: number of its contacts, then (scanned) the index of its first contact
	uint32_t chunk_cap;
	uint32_t max_contacts;        // capacity of the contact list, `order` and `level_of`
	uint32_t *order;              // contact indices grouped by level
	uint32_t *level_start;        // max_contacts + 1 (a level holds at least one contact)
	uint32_t *level_of;           // per contact
	uint32_t *owner;              // per particle: first unscheduled contact that touches it (written and read at the L2: the minima are atomics)
	uint32_t *counters;           // kTcWords
	uint32_t *imp_list;           // the contacts that carry a velocity impulse (pMax < 0), in list order (max_contacts entries)
	uint8_t *imp_mark;            // per particle: 1 while it takes part in such a contact as a dynamic particle (all zero between steps)
	// the (particle, slot) pairs of the impulse-carrying contacts -- slot = 5 * (index in imp_list) + role (0 = the contact particle, 1 .. 4 = the tet's
	// vertices) -- before and after a stable sort by particle: a particle's entries then sit side by side, in list order (5 * max_contacts entries each)
	uint32_t *imp_keys, *imp_slots, *imp_keys_sorted, *imp_slots_sorted;
	int force_impulses;           // developer aid (PBDX_OPT_TET_FORCE_IMPULSES): see tet_contact_velocity_impulse
};

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
	const uint32_t lane = threadIdx.x & 63u;
	for (int d = 1; d < 64; d <<= 1)
	{
		const uint32_t t = __shfl_up(v, d);
		if (lane >= (uint32_t)d) v += t;
	}
	return v;
}
// exclusive scan over the threads of a workgroup (a multiple of 64, at most 1024); lds: 65 words.  Every thread must call it.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &total)
{
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	const uint32_t inc = wave_inclusive_scan(v);
	if (lane == 63u) lds[w] = inc;
	__syncthreads();
	if (w == 0)
	{
		const uint32_t s = lane < (blockDim.x >> 6) ? lds[lane] : 0u;
		const uint32_t si = wave_inclusive_scan(s);
		lds[lane] = si - s;
		if (lane == 63u) lds[64] = si;
	}
	__syncthreads();
	const uint32_t r = lds[w] + inc - v;
	total = lds[64];
	__syncthreads();
	return r;
}

// ---- bounding spheres: one wavefront per node ----------------------------------------------------------------------------------
__device__ __forceinline__ float lane_value(float v, uint32_t j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)j)); }
__device__ __forceinline__ float wave_max(float v)
{
	for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(v, d); v = (v < o) ? o : v; }
	return v;
}
// grid: (ceil(max elements / 256), 2 * colliders)
__global__ __launch_bounds__(256) void tet_gather_kernel(const TetColliderView *views, const P4 *pos)
{
	const TetColliderView &v = views[blockIdx.y >> 1];
	const BvhView &b = (blockIdx.y & 1u) ? v.tet_bvh : v.points;
	const uint32_t total = (blockIdx.y & 1u) ? 4u * v.num_tets : v.num_vertices;
	const uint32_t e = blockIdx.x * 256u + threadIdx.x;
	if (e < total)
	{
		const P4 p = pos[v.first + b.flat[e]];
		b.gathered[e] = p;
		b.soa[e] = p.x; b.soa[(size_t)total + e] = p.y; b.soa[2 * (size_t)total + e] = p.z;
	}
}
// PointCloudBSH / TetMeshBSH::compute_hull_approx (hull_points / hull_tets of pbdx_tetcontact.h: same operations in the same order).
// The running sum is three dependent chains (x, y, z) of n additions each; nothing but the latency of a dependent v_add_f32 can bound a
// chain (1.75 ns on this GPU), so each chain gets a wavefront (= a SIMD) of its own -- and no instruction besides the additions:
// every LANE loads B consecutive values of its chain into its own registers (64 B values per chunk, one load instruction per 64 values),
// and the running sum visits the lanes in turn: all lanes execute `acc += v[0..B)`, then acc moves one lane up (v_mov_b32_dpp
// wave_ror:1).  After turn t lane t + 1 holds the sum through lane t's values and adds its own in the next turn; what the other lanes
// compute meanwhile is never looked at.  After 64 turns lane 0 holds the sum through the chunk and starts the next one.
// A partial last chunk is padded with +0, which is exact: a running sum that starts at +0 is never -0.
// (Measured on the 70875-tet root, 283500 values per component: operands by v_readlane 7.7 ms, by v_add_f32_dpp wave_shr:1 through the
// lanes -- a DPP add issues at under half rate -- 3.2 ms, by LDS broadcast read 5.3 / 1.0 ms (compiler- / hand-placed waits, staged by
// the fourth wavefront), this form: scripts/microbench/chain.hip variant G)
template <int B> __device__ __forceinline__ float chain_turns(float acc, const float (&v)[32])
{
	for (int t = 0; t < 64; t++)
	{
#pragma unroll
		for (int k = 0; k < B; k++) acc += v[k];
		acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x13C /* wave_ror:1 */, 0xf, 0xf, false));
	}
	return acc;
}
template <int B> __device__ __forceinline__ void chain_fetch(float (&v)[32], const float *g, uint32_t first, uint32_t m)
{
#pragma unroll
	for (int k = 0; k < B; k++) { const uint32_t e = first + (uint32_t)k; v[k] = e < m ? g[e] : 0.0f; }
}
// the sum of g[0 .. m) in index order, in every lane
template <int B> __device__ __forceinline__ float chain_sum(const float *g, uint32_t m)
{
	const uint32_t lane = threadIdx.x & 63u;
	float v[32], nx[32];
	chain_fetch<B>(v, g, lane * B, m);
	float acc = 0.0f;
	for (uint32_t c0 = 0; c0 < m; c0 += 64u * B)
	{
		chain_fetch<B>(nx, g, c0 + 64u * B + lane * B, m);
		acc = chain_turns<B>(acc, v);
#pragma unroll
		for (int k = 0; k < B; k++) v[k] = nx[k];
	}
	return lane_value(acc, 0);
}
// the same, continuing from `start` (wavefront-uniform)
__device__ __forceinline__ float chain_sum_from(float start, const float *g, uint32_t m)
{
	const uint32_t lane = threadIdx.x & 63u;
	float v[32];
	float acc = start;                                  // lane 0's is the one that counts
	for (uint32_t c0 = 0; c0 < m; c0 += 2048u)
	{
		chain_fetch<32>(v, g, c0 + lane * 32u, m);
		acc = chain_turns<32>(acc, v);
	}
	return lane_value(acc, 0);
}

// The long chains are the critical path, and a chain that shares its SIMD with other wavefronts' chains runs at a fraction of its speed:
// the nodes with at least kTcBigNode vertices (a static list: `big`, 2 words each: collider * 2 + hierarchy, node) go first, in a launch
// that asks for so much LDS that a CU takes one workgroup; everything else follows in a second launch (big == nullptr;
// grid: (max over colliders of nodes, 2 * colliders)).  Workgroup = 3 wavefronts = the 3 components.
constexpr uint32_t kTcBigNode = 8192;
constexpr uint32_t kTcBigNodeLds = 96 * 1024;
__global__ __launch_bounds__(192) void tet_hull_kernel2(const TetColliderView *views, const uint32_t *big, uint32_t *big_r2)
{
	__shared__ float s_sum[3];
	__shared__ float s_max[3];
	const uint32_t which = big ? big[2 * blockIdx.x] : blockIdx.y;
	const uint32_t node = big ? big[2 * blockIdx.x + 1] : blockIdx.x;
	const TetColliderView &v = views[which >> 1];
	const bool tets = (which & 1u) != 0;
	const BvhView &b = tets ? v.tet_bvh : v.points;
	if (node >= b.num_nodes) return;
	const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
	const uint32_t n = (uint32_t)b.nodes[4 * node + 3];
	const uint32_t m = n * b.per_entity;
	if (!big && m >= kTcBigNode) return;
	const size_t first = (size_t)(uint32_t)b.nodes[4 * node + 2] * b.per_entity;
	const float *gc = b.soa + (size_t)wave * b.num_elements + first;
	const float sum = m <= 64u ? chain_sum<1>(gc, m) : m <= 256u ? chain_sum<4>(gc, m) : chain_sum<32>(gc, m);
	if (lane == 0) s_sum[wave] = sum;
	__syncthreads();
	V3 x = mk(s_sum[0], s_sum[1], s_sum[2]);
	x = tets ? x / (4.0f * (float)n) : x / (float)n;
	if (big)
	{
		// the radius of a long node is a maximum over megabytes: not for three wavefronts (tet_big_radius_kernel)
		if (tid == 0) { P4 h; h.x = x.x; h.y = x.y; h.z = x.z; h.w = 0.0f; b.hulls[node] = h; big_r2[blockIdx.x] = 0u; }
		return;
	}
	const float4 *g = reinterpret_cast<const float4 *>(b.gathered) + first;
	float radius2 = 0.0f;
	for (uint32_t e = tid; e < m; e += 192)
	{
		const float4 q = g[e];
		const float d = sqn(x - mk(q.x, q.y, q.z));
		radius2 = (radius2 < d) ? d : radius2;
	}
	radius2 = wave_max(radius2);
	if (lane == 0) s_max[wave] = radius2;
	__syncthreads();
	if (tid == 0)
	{
		for (int q = 1; q < 3; q++) radius2 = (radius2 < s_max[q]) ? s_max[q] : radius2;
		P4 h; h.x = x.x; h.y = x.y; h.z = x.z;
		h.w = tets ? (float)(sqrt((double)radius2) + (double)v.tolerance) : sqrtf(radius2);
		b.hulls[node] = h;
	}
}

// radius of the long nodes.  slices: 2 words per workgroup (index into `big`, slice of kTcRadiusSlice vertices); squared distances are
// non-negative floats, whose order is the order of their bit patterns
constexpr uint32_t kTcRadiusSlice = 4096;
__global__ __launch_bounds__(256) void tet_big_radius_kernel(const TetColliderView *views, const uint32_t *big, const uint32_t *slices, uint32_t *big_r2)
{
	__shared__ float s_max[4];
	const uint32_t bi = slices[2 * blockIdx.x], slice = slices[2 * blockIdx.x + 1];
	const uint32_t which = big[2 * bi], node = big[2 * bi + 1];
	const TetColliderView &v = views[which >> 1];
	const BvhView &b = (which & 1u) ? v.tet_bvh : v.points;
	const uint32_t m = (uint32_t)b.nodes[4 * node + 3] * b.per_entity;
	const float4 *g = reinterpret_cast<const float4 *>(b.gathered) + (size_t)(uint32_t)b.nodes[4 * node + 2] * b.per_entity;
	const V3 x = p3(b.hulls[node]);
	float radius2 = 0.0f;
	const uint32_t end = (slice + 1u) * kTcRadiusSlice < m ? (slice + 1u) * kTcRadiusSlice : m;
	for (uint32_t e = slice * kTcRadiusSlice + threadIdx.x; e < end; e += 256)
	{
		const float4 q = g[e];
		const float d = sqn(x - mk(q.x, q.y, q.z));
		radius2 = (radius2 < d) ? d : radius2;
	}
	radius2 = wave_max(radius2);
	if ((threadIdx.x & 63u) == 0) s_max[threadIdx.x >> 6] = radius2;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		for (int q = 1; q < 4; q++) radius2 = (radius2 < s_max[q]) ? s_max[q] : radius2;
		atomicMax(&big_r2[bi], __float_as_uint(radius2));
	}
}
__global__ __launch_bounds__(256) void tet_big_finish_kernel(const TetColliderView *views, const uint32_t *big, uint32_t count, const uint32_t *big_r2)
{
	const uint32_t bi = blockIdx.x * 256u + threadIdx.x;
	if (bi >= count) return;
	const uint32_t which = big[2 * bi], node = big[2 * bi + 1];
	const TetColliderView &v = views[which >> 1];
	const BvhView &b = (which & 1u) ? v.tet_bvh : v.points;
	const float radius2 = __uint_as_float(big_r2[bi]);
	b.hulls[node].w = (which & 1u) ? (float)(sqrt((double)radius2) + (double)v.tolerance) : sqrtf(radius2);
}

// ---- traversal: BVHTest::traverse's recursion tree, generation by generation ----------------------------------------------------------
// One launch of `gridDim.x` co-resident workgroups (a quarter of the CUs is asked for; the launch is alone on its stream) that meet at a
// barrier between generations.  trav: zeroed before the launch.
constexpr uint32_t kTcMaxGenerations = 256;
constexpr uint32_t kTcNone = 0xffffffffu;
enum { kTrArrive = 0, kTrBad = 1, kTrLeaves = 2, kTrChunks = 3, kTrCount = 4, kTrWords = kTrCount + kTcMaxGenerations + 2 };

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long pack2(uint32_t lo, uint32_t hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); }
// all workgroups of the launch: every thread's (written-through) stores have completed, then one arrival per workgroup; returns false if
// the wait was abandoned (a workgroup never arrived within ~50 ms)
__device__ inline bool grid_barrier(uint32_t *trav, uint32_t &target, uint32_t *counters)
{
	__shared__ uint32_t s_ok;
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
	__syncthreads();
	target += gridDim.x;
	if (threadIdx.x == 0)
	{
		uint32_t ok = 1u;
		__hip_atomic_fetch_add(&trav[kTrArrive], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint64_t t0 = wall_clock64();
		while (ld_agent(&trav[kTrArrive]) < target)
		{
			__builtin_amdgcn_s_sleep(1);
			// abandoned: the flag the HOST reads back after every detection (kTcStack = 2) is raised by whoever gives up, so that the
			// step reports the failure whichever workgroup was late (the later kernels then see an empty list, but the step fails)
			if (wall_clock64() - t0 > 5000000ull) { ok = 0u; st_agent(&trav[kTrBad], 2u); st_agent(&counters[kTcStack], 2u); break; }
		}
		s_ok = ok;
	}
	__syncthreads();
	return s_ok != 0u;
}

__global__ __launch_bounds__(256) void tet_traverse_kernel(const TetColliderView *views, uint32_t n, const float *aabb, TetWork w)
{
	__shared__ uint32_t lds[65];
	__shared__ uint32_t gstart[kTcMaxGenerations + 2];
	struct PairPtr { const P4 *h1, *h2; const int32_t *n1, *n2; };
	constexpr uint32_t kCached = 64;
	__shared__ PairPtr s_pair[kCached];
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t gtid = blockIdx.x * 256u + tid, gthreads = gridDim.x * 256u;
	uint32_t *trav = w.trav;
	uint32_t target = 0;
	if (blockIdx.x == 0)
	{
		if (tid == 0) { w.counters[kTcCount] = 0; w.counters[kTcLeafPairs] = 0; w.counters[kTcChunks] = 0; w.counters[kTcLevels] = 0; }
		// generation 0: the ordered collider pairs (i, k), i outer (DistanceFieldCollisionDetection.cpp:33-46), at the roots
		uint32_t roots = 0;
		for (uint32_t c0 = 0; c0 < n * n; c0 += 256)
		{
			const uint32_t e = c0 + tid;
			uint32_t ok = 0, i = 0, k = 0;
			if (e < n * n)
			{
				i = e / n; k = e % n;
				ok = (i != k && views[i].test_mesh && views[i].points.num_nodes && views[k].tet_bvh.num_nodes && aabb_intersect(aabb + 6 * i, aabb + 6 * k)) ? 1u : 0u;
			}
			uint32_t total;
			const uint32_t at = roots + block_exclusive_scan(ok, lds, total);
			if (ok && at < w.node_cap)
			{
				st_agent(&w.pair_ik[2 * at], i); st_agent(&w.pair_ik[2 * at + 1], k);
				st_agent(&w.node_rec[2 * at], pack2(at, 0u)); st_agent(&w.node_rec[2 * at + 1], 0ull);
			}
			roots += total;
		}
		if (tid == 0)
		{
			if (roots > w.node_cap) { st_agent(&trav[kTrBad], 1u); roots = 0; }
			st_agent(&trav[kTrCount], roots);
		}
	}
	if (!grid_barrier(trav, target, w.counters)) return;
	const uint32_t roots = ld_agent(&trav[kTrCount]);
	if (tid < kCached && tid < roots)
	{
		const BvhView &b1 = views[ld_agent(&w.pair_ik[2 * tid])].points, &b2 = views[ld_agent(&w.pair_ik[2 * tid + 1])].tet_bvh;
		s_pair[tid] = PairPtr{ b1.hulls, b2.hulls, b1.nodes, b2.nodes };
	}
	if (tid == 0) gstart[0] = 0;
	__syncthreads();
	// expansion
	uint32_t generations = 0, base = 0;
	while (true)
	{
		const uint32_t cnt_g = ld_agent(&trav[kTrCount + generations]);
		if (tid == 0) gstart[generations + 1] = base + cnt_g;
		if (cnt_g == 0 || ld_agent(&trav[kTrBad]) || generations >= kTcMaxGenerations) break;
		const uint32_t ge = base + cnt_g;
		constexpr int kUnroll = 2;
		for (uint32_t i0 = base + gtid; i0 - lane < ge; i0 += gthreads * kUnroll)      // wave-uniform trip count
		{
			uint32_t rp[kUnroll], ra[kUnroll], rb[kUnroll]; P4 bs1[kUnroll], bs2[kUnroll]; int4 nd1[kUnroll], nd2[kUnroll]; bool live[kUnroll];
#pragma unroll
			for (int u = 0; u < kUnroll; u++)
			{
				const uint32_t idx = i0 + (uint32_t)u * gthreads;
				live[u] = idx < ge;
				if (live[u])
				{
					const unsigned long long r0 = ld_agent(&w.node_rec[2 * idx]), r1 = ld_agent(&w.node_rec[2 * idx + 1]);
					rp[u] = (uint32_t)r0; ra[u] = (uint32_t)(r0 >> 32); rb[u] = (uint32_t)r1;
				}
			}
#pragma unroll
			for (int u = 0; u < kUnroll; u++)
			{
				if (!live[u]) continue;
				PairPtr pp;
				if (rp[u] < kCached) pp = s_pair[rp[u]];
				else
				{
					const BvhView &b1 = views[ld_agent(&w.pair_ik[2 * rp[u]])].points, &b2 = views[ld_agent(&w.pair_ik[2 * rp[u] + 1])].tet_bvh;
					pp = PairPtr{ b1.hulls, b2.hulls, b1.nodes, b2.nodes };
				}
				bs1[u] = pp.h1[ra[u]]; bs2[u] = pp.h2[rb[u]];
				nd1[u] = *reinterpret_cast<const int4 *>(pp.n1 + 4 * ra[u]); nd2[u] = *reinterpret_cast<const int4 *>(pp.n2 + 4 * rb[u]);
			}
#pragma unroll
			for (int u = 0; u < kUnroll; u++)
			{
				const uint32_t idx = i0 + (uint32_t)u * gthreads;
				unsigned long long cnt = 0ull;
				bool expand = false, leaf1 = false, leaf2 = false;
				if (live[u] && spheres_overlap(bs1[u], bs2[u]))
				{
					leaf1 = nd1[u].x < 0 && nd1[u].y < 0; leaf2 = nd2[u].x < 0 && nd2[u].y < 0;
					if (leaf1 && leaf2) cnt = pack2(1u, ((uint32_t)nd1[u].w * (uint32_t)nd2[u].w + 63u) / 64u);
					else expand = true;
				}
				// two slots of the next generation per expanding node pair: one atomic per wavefront
				const unsigned long long m = __ballot(expand);
				uint32_t child = kTcNone;
				if (m)
				{
					const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
					uint32_t first = 0;
					if (lane == leader) first = __hip_atomic_fetch_add(&trav[kTrCount + generations + 1], 2u * (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					first = __shfl(first, (int)leader);
					if (expand)
					{
						const uint32_t c = ge + first + 2u * (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
						if (c + 1u < w.node_cap)
						{
							// descend the smaller sphere's hierarchy first unless it is at a leaf; children[0] before children[1]
							const bool descend1 = (bs1[u].w < bs2[u].w) ? !leaf1 : leaf2;
							child = c;
							if (descend1)
							{
								st_agent(&w.node_rec[2 * c], pack2(rp[u], (uint32_t)nd1[u].x)); st_agent(&w.node_rec[2 * c + 1], (unsigned long long)rb[u]);
								st_agent(&w.node_rec[2 * c + 2], pack2(rp[u], (uint32_t)nd1[u].y)); st_agent(&w.node_rec[2 * c + 3], (unsigned long long)rb[u]);
							}
							else
							{
								st_agent(&w.node_rec[2 * c], pack2(rp[u], ra[u])); st_agent(&w.node_rec[2 * c + 1], (unsigned long long)(uint32_t)nd2[u].x);
								st_agent(&w.node_rec[2 * c + 2], pack2(rp[u], ra[u])); st_agent(&w.node_rec[2 * c + 3], (unsigned long long)(uint32_t)nd2[u].y);
							}
						}
						else st_agent(&trav[kTrBad], 1u);
					}
				}
				if (live[u]) { st_agent(&w.node_cnt[idx], cnt); st_agent(&w.node_child[idx], child); }
			}
		}
		if (!grid_barrier(trav, target, w.counters)) return;
		base = ge;
		generations++;
	}
	__syncthreads();
	const uint32_t bad = ld_agent(&trav[kTrBad]);
	if (bad || (generations >= kTcMaxGenerations && gstart[generations + 1] != gstart[generations]))
	{
		// kTcStack: 1 node pairs, 2 a barrier was abandoned, 3 generations, 4 leaf pairs / chunks
		if (gtid == 0) { w.counters[kTcStack] = bad ? bad : 3u; w.counters[kTcGenerations] = generations; w.counters[kTcTreeNodes] = gstart[generations + 1]; }
		return;
	}
	// bottom-up: leaf pairs and candidate chunks below every node (the two halves never carry into each other: both are bounded by capacities)
	for (uint32_t g = generations; g-- > 0;)
	{
		for (uint32_t idx = gstart[g] + gtid; idx < gstart[g + 1]; idx += gthreads)
		{
			const uint32_t child = ld_agent(&w.node_child[idx]);
			if (child != kTcNone) st_agent(&w.node_cnt[idx], ld_agent(&w.node_cnt[child]) + ld_agent(&w.node_cnt[child + 1]));
		}
		if (!grid_barrier(trav, target, w.counters)) return;
	}
	// top-down: where every subtree's leaf pairs / chunks start
	if (blockIdx.x == 0)
	{
		uint32_t leaves = 0, chunks = 0;
		for (uint32_t c0 = 0; c0 < roots; c0 += 256)
		{
			const uint32_t r = c0 + tid;
			const unsigned long long cnt = r < roots ? ld_agent(&w.node_cnt[r]) : 0ull;
			uint32_t t0, t1;
			const uint32_t o0 = leaves + block_exclusive_scan((uint32_t)cnt, lds, t0);
			const uint32_t o1 = chunks + block_exclusive_scan((uint32_t)(cnt >> 32), lds, t1);
			if (r < roots) st_agent(&w.node_off[r], pack2(o0, o1));
			leaves += t0; chunks += t1;
		}
		if (tid == 0) { st_agent(&trav[kTrLeaves], leaves); st_agent(&trav[kTrChunks], chunks); }
	}
	if (!grid_barrier(trav, target, w.counters)) return;
	const uint32_t leaves = ld_agent(&trav[kTrLeaves]), chunks = ld_agent(&trav[kTrChunks]);
	if (leaves > w.front_cap || chunks > w.chunk_cap)
	{
		if (gtid == 0) { w.counters[kTcStack] = 4u; w.counters[kTcGenerations] = generations; w.counters[kTcTreeNodes] = gstart[generations]; w.counters[kTcLeafPairs] = leaves; w.counters[kTcChunks] = chunks; }
		return;
	}
	for (uint32_t g = 0; g < generations; g++)
	{
		for (uint32_t idx = gstart[g] + gtid; idx < gstart[g + 1]; idx += gthreads)
		{
			const uint32_t child = ld_agent(&w.node_child[idx]);
			const unsigned long long off = ld_agent(&w.node_off[idx]);
			if (child != kTcNone)
			{
				st_agent(&w.node_off[child], off);
				st_agent(&w.node_off[child + 1], off + ld_agent(&w.node_cnt[child]));
			}
			else
			{
				const unsigned long long cnt = ld_agent(&w.node_cnt[idx]);
				if ((uint32_t)cnt == 1u)
				{
					const unsigned long long r0 = ld_agent(&w.node_rec[2 * idx]), r1 = ld_agent(&w.node_rec[2 * idx + 1]);
					const uint32_t at = (uint32_t)off, chunk0 = (uint32_t)(off >> 32), nchunks = (uint32_t)(cnt >> 32);
					w.leaf_pairs[3 * at] = (uint32_t)r0; w.leaf_pairs[3 * at + 1] = (uint32_t)(r0 >> 32); w.leaf_pairs[3 * at + 2] = (uint32_t)r1;
					w.chunk_off[at] = chunk0;
					for (uint32_t j = 0; j < nchunks; j++) w.chunk_pair[chunk0 + j] = at;
				}
			}
		}
		if (g + 1 < generations && !grid_barrier(trav, target, w.counters)) return;
	}
	if (gtid == 0)
	{
		w.chunk_off[leaves] = chunks;
		w.counters[kTcGenerations] = generations;
		w.counters[kTcLeafPairs] = leaves;
		w.counters[kTcChunks] = chunks;
		w.counters[kTcTreeNodes] = gstart[generations];
	}
}

// ---- candidates: one wavefront per chunk of 64 (point, tet) candidates of a leaf pair -------------------------------------------------
// kWrite = false: ballot of the candidates that are contacts.  kWrite = true: the contacts go to their place in the list.
template <bool kWrite>
__global__ __launch_bounds__(256) void tet_candidates_kernel(const TetColliderView *views, const P4 *pos, const P4 *rest, const P4 *vel, TetWork w, TetContact *contacts)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t chunks = w.counters[kTcChunks];
	const uint32_t waves = gridDim.x * 4u;
	for (uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6); q < chunks; q += waves)
	{
		unsigned long long mask = 0;
		if (kWrite)
		{
			mask = w.chunk_mask[q];
			if (!mask) continue;
		}
		const uint32_t e = w.chunk_pair[q];
		const uint32_t p = w.leaf_pairs[3 * e], a = w.leaf_pairs[3 * e + 1], b = w.leaf_pairs[3 * e + 2];
		const TetColliderView &co1 = views[w.pair_ik[2 * p]], &co2 = views[w.pair_ik[2 * p + 1]];
		const uint32_t beg1 = (uint32_t)co1.points.nodes[4 * a + 2], n1 = (uint32_t)co1.points.nodes[4 * a + 3];
		const uint32_t beg2 = (uint32_t)co2.tet_bvh.nodes[4 * b + 2], n2 = (uint32_t)co2.tet_bvh.nodes[4 * b + 3];
		const uint32_t idx = (q - w.chunk_off[e]) * 64u + lane;
		bool hit = false;
		TetContact c;
		if (idx < n1 * n2 && (!kWrite || ((mask >> lane) & 1ull)))
			hit = tet_contact_candidate(co2, pos, rest, vel, co1.points.lst[beg1 + idx / n2] + co1.first, co2.tet_bvh.lst[beg2 + idx % n2], c);
		if (!kWrite)
		{
			const unsigned long long m = __ballot(hit);
			if (lane == 0) { w.chunk_mask[q] = m; w.chunk_base[q] = (uint32_t)__popcll(m); }
		}
		else if (hit)
		{
			const uint32_t at = w.chunk_base[q] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
			if (at < w.max_contacts) contacts[at] = c;
		}
	}
}
// chunk populations -> index of every chunk's first contact; total -> the contact count
__global__ __launch_bounds__(1024) void tet_chunk_scan_kernel(TetWork w)
{
	__shared__ uint32_t lds[65];
	const uint32_t chunks = w.counters[kTcChunks];
	uint32_t base = 0;
	for (uint32_t c0 = 0; c0 < chunks; c0 += 1024)
	{
		const uint32_t q = c0 + threadIdx.x;
		const uint32_t v = q < chunks ? w.chunk_base[q] : 0u;
		uint32_t total;
		const uint32_t at = base + block_exclusive_scan(v, lds, total);
		if (q < chunks) w.chunk_base[q] = at;
		base += total;
	}
	if (threadIdx.x == 0)
	{
		w.counters[kTcCount] = base < w.max_contacts ? base : w.max_contacts;
		if (base > w.max_contacts) w.counters[kTcOverflow] = 1u;          // the list is full
	}
}

// ---- levels: order-preserving decomposition of the sequential solve into sets of contacts on disjoint particles ----------------------
// Up to kTcSmallList contacts (one per thread) the rounds run out of LDS: the particles get slots of a hash table, the "first unscheduled
// contact at each particle" is an LDS atomic instead of a round trip to the L2 per round (20 rounds: 0.18 ms -> see DESIGN.md).
constexpr uint32_t kTcSmallList = 1024;
constexpr uint32_t kTcHashSlots = 8192;           // >= 5 * kTcSmallList / 0.63
__global__ __launch_bounds__(1024) void tet_levels_kernel(const TetContact *contacts, TetWork w)
{
	__shared__ uint32_t lds[65];
	__shared__ uint32_t h_key[kTcHashSlots], h_owner[kTcHashSlots];
	const uint32_t n = w.counters[kTcCount];
	const uint32_t tid = threadIdx.x;
	if (n <= kTcSmallList)
	{
		for (uint32_t q = tid; q < kTcHashSlots; q += 1024) h_key[q] = 0xffffffffu;
		__syncthreads();
		uint32_t slot[5];
		const bool mine = tid < n;
		if (mine)
		{
			const TetContact &k = contacts[tid];
			const uint32_t ids[5] = { k.particle, k.vert[0], k.vert[1], k.vert[2], k.vert[3] };
			for (int j = 0; j < 5; j++)
			{
				uint32_t h = (ids[j] * 2654435761u) >> 19;          // 13 bits
				while (true)
				{
					const uint32_t old = atomicCAS(&h_key[h], 0xffffffffu, ids[j]);
					if (old == 0xffffffffu || old == ids[j]) break;
					h = (h + 1u) & (kTcHashSlots - 1u);
				}
				slot[j] = h;
			}
		}
		bool scheduled = !mine;
		uint32_t done = 0, level = 0;
		while (done < n)
		{
			__syncthreads();
			if (!scheduled) for (int j = 0; j < 5; j++) h_owner[slot[j]] = 0xffffffffu;
			__syncthreads();
			if (!scheduled) for (int j = 0; j < 5; j++) atomicMin(&h_owner[slot[j]], tid);
			__syncthreads();
			uint32_t ready = 0;
			if (!scheduled)
			{
				ready = 1u;
				for (int j = 0; j < 5; j++) if (h_owner[slot[j]] != tid) ready = 0u;
			}
			uint32_t total;
			const uint32_t at = done + block_exclusive_scan(ready, lds, total);
			if (ready) { w.order[at] = tid; w.level_of[tid] = level; scheduled = true; }
			if (tid == 0) w.level_start[level] = done;
			done += total;
			level++;
		}
		if (tid == 0) { w.level_start[level] = done; w.counters[kTcLevels] = level; }
		return;
	}
	for (uint32_t c = tid; c < n; c += 1024) w.level_of[c] = 0xffffffffu;
	uint32_t done = 0, level = 0;
	while (done < n && level < w.max_contacts)
	{
		__syncthreads();
		for (uint32_t c = tid; c < n; c += 1024)
			if (w.level_of[c] == 0xffffffffu)
			{
				const TetContact &k = contacts[c];
				__hip_atomic_store(&w.owner[k.particle], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				for (int j = 0; j < 4; j++) __hip_atomic_store(&w.owner[k.vert[j]], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		__threadfence();
		__syncthreads();
		for (uint32_t c = tid; c < n; c += 1024)
			if (w.level_of[c] == 0xffffffffu)
			{
				const TetContact &k = contacts[c];
				atomicMin(&w.owner[k.particle], c);
				for (int j = 0; j < 4; j++) atomicMin(&w.owner[k.vert[j]], c);
			}
		__threadfence();
		__syncthreads();
		uint32_t next = done;
		for (uint32_t c0 = 0; c0 < n; c0 += 1024)
		{
			const uint32_t c = c0 + tid;
			uint32_t ready = 0;
			if (c < n && w.level_of[c] == 0xffffffffu)
			{
				const TetContact &k = contacts[c];
				ready = (__hip_atomic_load(&w.owner[k.particle], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c) ? 1u : 0u;
				for (int j = 0; j < 4; j++) if (__hip_atomic_load(&w.owner[k.vert[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != c) ready = 0;
			}
			uint32_t total;
			const uint32_t at = next + block_exclusive_scan(ready, lds, total);
			if (ready) { w.order[at] = c; w.level_of[c] = level; }
			next += total;
		}
		if (tid == 0) w.level_start[level] = done;
		done = next;
		level++;
	}
	if (tid == 0)
	{
		w.level_start[level] = done;
		w.counters[kTcLevels] = level;
		if (done < n) w.counters[kTcOverflow] = 2u;                 // cannot happen: every round schedules the first unscheduled contact
	}
}

// TimeStepController.cpp:288-291, level by level
struct TetPosAccess
{
	float4 *pos;
	__device__ __forceinline__ P4 get(uint32_t i) const { const float4 v = pos[i]; P4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
	__device__ __forceinline__ void add(uint32_t i, V3 c) { float4 v = pos[i]; v.x += c.x; v.y += c.y; v.z += c.z; pos[i] = v; }
};
__global__ __launch_bounds__(1024) void tet_contact_solve_levels_kernel(float4 *pos, const TetContact *contacts, TetWork w)
{
	__shared__ uint32_t s_start[kTcSmallList + 2];
	TetPosAccess acc = { pos };
	const uint32_t levels = w.counters[kTcLevels];
	const uint32_t n = w.counters[kTcCount];
	if (n <= kTcSmallList)
	{
		// one contact per thread, fetched before the first level: a level then costs one round trip for the particle's position
		for (uint32_t l = threadIdx.x; l <= levels; l += 1024) s_start[l] = w.level_start[l];
		TetContact mine;
		if (threadIdx.x < n) mine = contacts[w.order[threadIdx.x]];
		__syncthreads();
		for (uint32_t l = 0; l < levels; l++)
		{
			if (threadIdx.x >= s_start[l] && threadIdx.x < s_start[l + 1]) tet_contact_position_solve(mine, acc);
			__syncthreads();
		}
		return;
	}
	for (uint32_t l = 0; l < levels; l++)
	{
		const uint32_t end = w.level_start[l + 1];
		for (uint32_t i = w.level_start[l] + threadIdx.x; i < end; i += 1024) tet_contact_position_solve(contacts[w.order[i]], acc);
		__syncthreads();
	}
}


// ---- velocity solve of the particle-tet contacts (friction 0; pbdx_tetcontact.h) --------------------------------------------------------
// A contact's impulse is a constant of the contact (it is computed from the velocities at detection), it is applied once per iteration of
// velocityConstraintProjection after the rigid-body contacts of that iteration, and static rigid bodies couple nothing: the whole velocity
// solve decomposes into per-particle chains  [contacts of p with rigid bodies, sweep it] [impulses of the tet contacts p takes part in,
// in list order]  for it = 0 .. maxIterationsV - 1.  Only contacts with pMax < 0 carry an impulse (usually none, in near-normal impacts
// about every second one): they are compacted in list order, the dynamic particles they touch are marked, and one lane per marked particle
// (the lane of the particle's FIRST appearance in the compacted list) runs that particle's chain; the rigid-body contact kernel leaves
// marked particles alone.
__global__ __launch_bounds__(1024) void tet_impulse_list_kernel(const TetContact *contacts, const P4 *pos, TetWork w)
{
	__shared__ uint32_t lds[65];
	const uint32_t n = w.counters[kTcCount];
	uint32_t base = 0;
	for (uint32_t c0 = 0; c0 < n; c0 += 1024)
	{
		const uint32_t q = c0 + threadIdx.x;
		uint32_t flag = 0;
		if (q < n)
		{
			const TetContact &c = contacts[q];
			V3 pv;
			if (tet_contact_velocity_impulse(c, pos[c.particle].w, pv, w.force_impulses != 0))
			{
				flag = 1;
				if (pos[c.particle].w != 0.0f) w.imp_mark[c.particle] = 1;
				for (int k = 0; k < 4; k++) if (c.w[k] != 0.0f) w.imp_mark[c.vert[k]] = 1;
			}
		}
		uint32_t total;
		const uint32_t at = base + block_exclusive_scan(flag, lds, total);
		if (flag) w.imp_list[at] = q;
		base += total;
	}
	if (threadIdx.x == 0) w.counters[kTcImpulses] = base;
}

// keys of the sort: the particle of every (impulse contact, role) slot, or ~0 where the particle is static in all its contacts (sorts to the end)
__global__ __launch_bounds__(256) void tet_impulse_pairs_kernel(const TetContact *contacts, TetWork w)
{
	const uint32_t count = w.counters[kTcImpulses];
	for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < 5u * count; slot += gridDim.x * blockDim.x)
	{
		const uint32_t e = slot / 5u, r = slot % 5u;
		const TetContact &c = contacts[w.imp_list[e]];
		const uint32_t p = r == 0 ? c.particle : c.vert[r - 1];
		w.imp_keys[slot] = w.imp_mark[p] ? p : 0xffffffffu;
		w.imp_slots[slot] = slot;
	}
}

struct TetImpulseChain          // Extra of particle_contacts (pbdx_contact.h): the tet-contact impulses of particle `p`, one iteration's worth
{
	const TetContact *contacts;
	const P4 *pos;
	const uint32_t *list;
	const uint32_t *keys, *slots;       // sorted by particle (stable: list order inside a particle)
	uint32_t begin, total, p;           // p's entries start at `begin`
	bool force;
	__device__ void after_sweep(V3 &v) const
	{
		// p's contacts in list order; a tet's four vertices are distinct and its contact particle belongs to another solid, so p has ONE role per contact
		for (uint32_t j = begin; j < total && keys[j] == p; j++)
		{
			const uint32_t slot = slots[j], e = slot / 5u, r = slot % 5u;
			const TetContact &c = contacts[list[e]];
			const float w0 = pos[c.particle].w;
			V3 pv, corr;
			if (!tet_contact_velocity_impulse(c, w0, pv, force)) continue;
			if (tet_contact_velocity_share(c, w0, pv, (int)r, corr)) v = v + corr;
		}
	}
};

struct TetImpulseArgs
{
	const TetContact *contacts;
	const float4 *pos;
	float4 *vel;
	TetWork w;
	const pbdx_collider *colliders; uint32_t num_colliders;
	const pbdx_collision_range *ranges; uint32_t num_ranges;
	float tolerance, stiffness;
	uint32_t iterations;
	unsigned int *contact_counters;      // of the rigid-body contacts: [0] contacts, [1] overflow flag (may be null)
};
// One lane per marked particle: the lane at the START of the particle's run in the sorted pairs runs its chain.  Linear in the number of
// (contact, particle) pairs (the first form found a particle's first appearance and its contacts by scanning the whole list per lane).
__global__ __launch_bounds__(256) void tet_impulse_kernel(TetImpulseArgs a)
{
	const uint32_t total = 5u * a.w.counters[kTcImpulses];
	const P4 *pos = reinterpret_cast<const P4 *>(a.pos);
	const uint32_t *keys = a.w.imp_keys_sorted, *slots = a.w.imp_slots_sorted;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
	{
		const uint32_t p = keys[i];
		if (p == 0xffffffffu) continue;                        // static in this contact and in every other one
		if (i && keys[i - 1] == p) continue;                   // not the start of p's run
		const float4 x = a.pos[p];
		float4 vv = a.vel[p];
		if (vv.w == 0.0f) continue;                            // (marks are only set for dynamic particles)
		V3 v = mk(vv.x, vv.y, vv.z);
		const TetImpulseChain chain = { a.contacts, pos, a.w.imp_list, keys, slots, i, total, p, a.w.force_impulses != 0 };
		// p's contacts with the static rigid bodies, if it belongs to a collision range (pbdx_contact.h)
		const pbdx_collision_range *rg = nullptr;
		if (a.num_colliders) for (uint32_t q = 0; q < a.num_ranges; q++) if (p >= a.ranges[q].first && p - a.ranges[q].first < a.ranges[q].count) rg = &a.ranges[q];
		if (rg)
		{
			const int nc = particle_contacts(mk(x.x, x.y, x.z), v, x.w, vv.w, a.colliders, a.num_colliders, a.tolerance, a.stiffness, rg->restitution, rg->friction, a.iterations, chain);
			if (nc < 0) { if (a.contact_counters) atomicExch(&a.contact_counters[1], 1u); continue; }
			if (nc > 0 && a.contact_counters) atomicAdd(&a.contact_counters[0], (unsigned int)nc);
		}
		else
			for (uint32_t it = 0; it < a.iterations; it++) chain.after_sweep(v);
		a.vel[p] = make_float4(v.x, v.y, v.z, vv.w);
	}
}
// the marks go back to zero once every chain has run (separate launch: a chain reads the marks of other particles' lanes' contacts)
__global__ __launch_bounds__(256) void tet_impulse_clear_kernel(const TetContact *contacts, TetWork w)
{
	const uint32_t count = w.counters[kTcImpulses];
	for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x)
	{
		const TetContact &c = contacts[w.imp_list[e]];
		w.imp_mark[c.particle] = 0;
		for (int k = 0; k < 4; k++) w.imp_mark[c.vert[k]] = 0;
	}
}

} // namespace pbdx
#endif
