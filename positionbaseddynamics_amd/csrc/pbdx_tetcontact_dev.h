// pbdx_tetcontact_dev.h -- the parallel form of the contacts between deformable solids (device only; arithmetic in pbdx_tetcontact.h).
//
// What is sequential in the reference and what that means here (DistanceFieldCollisionDetection.cpp:160-177, 361-483;
// BoundingSphereHierarchy.cpp:124-174; TimeStepController.cpp:288-291):
//   * the contact list is appended to in the order in which ONE recursive dual-hierarchy traversal reaches the leaf pairs, and
//   * that list is solved one contact after the other inside every iteration of the next step (Gauss-Seidel): the order is part of
//     the result.
// Neither needs one thread:
//   detection  The recursion tree of BVHTest::traverse is expanded breadth-wise, every generation as an ORDERED list in which a node
//              pair is replaced in place by its (up to) two children, children[0] first -- an order-preserving expansion of a tree
//              leaves its leaves in depth-first order, which is the reference's visiting order.  A generation is one pass of a
//              workgroup-wide exclusive scan (tet_traverse_kernel).  The candidates of a leaf pair (points x tets, row-major as the
//              reference's two loops) are then evaluated by the whole GPU, 64 candidates per wavefront; a ballot per wavefront, a scan
//              over the ballots' populations and a second evaluation of the (few) hits put the contacts where the reference's
//              push_back would have (tet_candidates_kernel<false>, tet_chunk_scan_kernel, tet_candidates_kernel<true>).
//   solve      Two contacts commute bit for bit unless they share a particle (the solve writes the contact's particle and the four
//              vertices of its tet, and reads nothing else that changes).  The list is levelled once per detection: a contact's
//              level is one more than the highest level among EARLIER contacts it shares a particle with (tet_levels_kernel: rounds
//              of "who is the first unscheduled contact at each particle").  Every level is a set of contacts on disjoint particles;
//              the levels run in order, each in parallel (tet_contact_solve_levels_kernel).
//   spheres    KDTree::update recomputes every node's sphere from ITS OWN entity range: the centre is a float sum in list order (a
//              dependent chain; one wavefront per node fetches 64 entities at a time, the chain reads them from registers), the
//              radius a maximum (order-free, lane-parallel).
#ifndef PBDX_TETCONTACT_DEV_H
#define PBDX_TETCONTACT_DEV_H

#include "pbdx_tetcontact.h"
#include <hip/hip_runtime.h>

namespace pbdx {

constexpr uint32_t kMaxTetContacts = 1u << 16;
constexpr uint32_t kMaxTetLevels = 4096;
constexpr uint32_t kTcFinal = 0x80000000u;
enum { kTcCount = 0, kTcOverflow = 1, kTcStack = 2, kTcLeafPairs = 3, kTcChunks = 4, kTcLevels = 5, kTcGenerations = 6, kTcWords = 8 };

struct TetWork                    // device scratch of the detection; *_cap are capacities in elements
{
	uint32_t *front[2];           // 3 words per node pair: collider pair | kTcFinal, node of the point hierarchy, node of the tet hierarchy
	uint32_t front_cap;
	uint32_t *pair_ik;            // 2 words per ordered collider pair (i, k) whose boxes intersect
	uint32_t *chunk_off;          // per leaf pair: its first 64-candidate chunk (front_cap + 1 entries)
	uint32_t *chunk_pair;         // per chunk: its leaf pair
	unsigned long long *chunk_mask; // per chunk: which of its candidates are contacts
	uint32_t *chunk_base;         // per chunk: number of its contacts, then (scanned) the index of its first contact
	uint32_t chunk_cap;
	uint32_t *order;              // contact indices grouped by level
	uint32_t *level_start;        // kMaxTetLevels + 1
	uint32_t *level_of;           // per contact
	uint32_t *owner;              // per particle: first unscheduled contact that touches it (written and read at the L2: the minima are atomics)
	uint32_t *counters;           // kTcWords
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
// PointCloudBSH::compute_hull_approx (hull_points of pbdx_tetcontact.h, same operations in the same order)
__device__ inline void hull_points_wave(const BvhView &b, uint32_t node, const P4 *pos)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t beg = (uint32_t)b.nodes[4 * node + 2], n = (uint32_t)b.nodes[4 * node + 3];
	V3 x = mk(0.0f, 0.0f, 0.0f);
	for (uint32_t c0 = 0; c0 < n; c0 += 64)
	{
		P4 p; p.x = p.y = p.z = p.w = 0.0f;
		if (c0 + lane < n) p = pos[b.lst[beg + c0 + lane]];
		const uint32_t m = (n - c0 < 64u) ? n - c0 : 64u;
		for (uint32_t j = 0; j < m; j++) x = x + mk(lane_value(p.x, j), lane_value(p.y, j), lane_value(p.z, j));
	}
	x = x / (float)n;
	float radius2 = 0.0f;
	for (uint32_t i = lane; i < n; i += 64)
	{
		const float d = sqn(x - p3(pos[b.lst[beg + i]]));
		radius2 = (radius2 < d) ? d : radius2;
	}
	radius2 = wave_max(radius2);
	if (lane == 0) { P4 h; h.x = x.x; h.y = x.y; h.z = x.z; h.w = sqrtf(radius2); b.hulls[node] = h; }
}
// TetMeshBSH::compute_hull_approx (hull_tets)
__device__ inline void hull_tets_wave(const BvhView &b, uint32_t node, const P4 *pos, const uint32_t *tets, float tolerance)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t beg = (uint32_t)b.nodes[4 * node + 2], n = (uint32_t)b.nodes[4 * node + 3];
	V3 x = mk(0.0f, 0.0f, 0.0f);
	for (uint32_t c0 = 0; c0 < n; c0 += 64)
	{
		P4 q[4];
		for (int k = 0; k < 4; k++) q[k].x = q[k].y = q[k].z = q[k].w = 0.0f;
		if (c0 + lane < n)
		{
			const uint32_t t = b.lst[beg + c0 + lane];
			for (int k = 0; k < 4; k++) q[k] = pos[tets[4 * t + k]];
		}
		const uint32_t m = (n - c0 < 64u) ? n - c0 : 64u;
		for (uint32_t j = 0; j < m; j++)
			for (int k = 0; k < 4; k++) x = x + mk(lane_value(q[k].x, j), lane_value(q[k].y, j), lane_value(q[k].z, j));
	}
	x = x / (4.0f * (float)n);
	float radius2 = 0.0f;
	for (uint32_t i = lane; i < n; i += 64)
	{
		const uint32_t t = b.lst[beg + i];
		for (int k = 0; k < 4; k++)
		{
			const float d = sqn(x - p3(pos[tets[4 * t + k]]));
			radius2 = (radius2 < d) ? d : radius2;
		}
	}
	radius2 = wave_max(radius2);
	if (lane == 0) { P4 h; h.x = x.x; h.y = x.y; h.z = x.z; h.w = (float)(sqrt((double)radius2) + (double)tolerance); b.hulls[node] = h; }
}
// grid: (max over colliders of ceil(nodes / 4), 2 * colliders); 4 wavefronts per workgroup
__global__ __launch_bounds__(256) void tet_hull_wave_kernel(const TetColliderView *views, const P4 *pos)
{
	const TetColliderView &v = views[blockIdx.y >> 1];
	const uint32_t node = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (blockIdx.y & 1u) { if (node < v.tet_bvh.num_nodes) hull_tets_wave(v.tet_bvh, node, pos + v.first, v.tets, v.tolerance); }
	else if (node < v.points.num_nodes) hull_points_wave(v.points, node, pos + v.first);
}

// ---- traversal: ordered breadth-wise expansion of BVHTest::traverse's recursion tree ------------------------------------------------
__global__ __launch_bounds__(1024) void tet_traverse_kernel(const TetColliderView *views, uint32_t n, const float *aabb, TetWork w)
{
	__shared__ uint32_t lds[65];
	const uint32_t tid = threadIdx.x;
	if (tid == 0) { w.counters[kTcCount] = 0; w.counters[kTcLeafPairs] = 0; w.counters[kTcChunks] = 0; w.counters[kTcLevels] = 0; }
	// generation 0: the ordered collider pairs (i, k), i outer (DistanceFieldCollisionDetection.cpp:33-46), at the roots
	uint32_t count = 0;
	for (uint32_t c0 = 0; c0 < n * n; c0 += 1024)
	{
		const uint32_t e = c0 + tid;
		uint32_t ok = 0, i = 0, k = 0;
		if (e < n * n)
		{
			i = e / n; k = e % n;
			ok = (i != k && views[i].test_mesh && views[i].points.num_nodes && views[k].tet_bvh.num_nodes && aabb_intersect(aabb + 6 * i, aabb + 6 * k)) ? 1u : 0u;
		}
		uint32_t total;
		const uint32_t at = count + block_exclusive_scan(ok, lds, total);
		if (ok && at < w.front_cap)
		{
			w.pair_ik[2 * at] = i; w.pair_ik[2 * at + 1] = k;
			w.front[0][3 * at] = at; w.front[0][3 * at + 1] = 0; w.front[0][3 * at + 2] = 0;
		}
		count += total;
	}
	bool overflow = count > w.front_cap;
	uint32_t cur = 0, generation = 0;
	int any = overflow ? 0 : (count != 0);
	while (any && !overflow)
	{
		__syncthreads();
		const uint32_t *src = w.front[cur];
		uint32_t *dst = w.front[cur ^ 1u];
		uint32_t next = 0;
		int pending = 0;
		for (uint32_t c0 = 0; c0 < count; c0 += 1024)
		{
			const uint32_t e = c0 + tid;
			uint32_t k = 0, o[6];
			if (e < count)
			{
				const uint32_t p = src[3 * e], a = src[3 * e + 1], b = src[3 * e + 2];
				if (p & kTcFinal) { k = 1; o[0] = p; o[1] = a; o[2] = b; }
				else
				{
					const BvhView &b1 = views[w.pair_ik[2 * p]].points, &b2 = views[w.pair_ik[2 * p + 1]].tet_bvh;
					const P4 bs1 = b1.hulls[a], bs2 = b2.hulls[b];
					if (spheres_overlap(bs1, bs2))
					{
						const int32_t a0 = b1.nodes[4 * a], a1 = b1.nodes[4 * a + 1], d0 = b2.nodes[4 * b], d1 = b2.nodes[4 * b + 1];
						const bool leaf1 = a0 < 0 && a1 < 0, leaf2 = d0 < 0 && d1 < 0;
						if (leaf1 && leaf2) { k = 1; o[0] = p | kTcFinal; o[1] = a; o[2] = b; }
						else
						{
							// descend the smaller sphere's hierarchy first unless it is at a leaf; children[0] before children[1]
							const bool descend1 = (bs1.w < bs2.w) ? !leaf1 : leaf2;
							k = 2; pending = 1;
							o[0] = p; o[3] = p;
							if (descend1) { o[1] = (uint32_t)a0; o[2] = b; o[4] = (uint32_t)a1; o[5] = b; }
							else { o[1] = a; o[2] = (uint32_t)d0; o[4] = a; o[5] = (uint32_t)d1; }
						}
					}
				}
			}
			uint32_t total;
			const uint32_t at = next + block_exclusive_scan(k, lds, total);
			if (k >= 1 && at < w.front_cap) { dst[3 * at] = o[0]; dst[3 * at + 1] = o[1]; dst[3 * at + 2] = o[2]; }
			if (k == 2 && at + 1 < w.front_cap) { dst[3 * at + 3] = o[3]; dst[3 * at + 4] = o[4]; dst[3 * at + 5] = o[5]; }
			next += total;
		}
		any = __syncthreads_or(pending);
		count = next; cur ^= 1u;
		overflow = count > w.front_cap;
		if (++generation >= 512u) { overflow = overflow || any; break; }
	}
	__syncthreads();
	if (overflow)
	{
		if (tid == 0) w.counters[kTcStack] = 1u;
		return;
	}
	// the generation that is left holds the overlapping leaf pairs in the reference's visiting order; if it ended up in front[1], copy
	// it over so that the following kernels read front[0]
	if (cur == 1u)
	{
		for (uint32_t e = tid; e < 3 * count; e += 1024) w.front[0][e] = w.front[1][e];
		__syncthreads();
	}
	// 64-candidate chunks per leaf pair
	uint32_t chunks = 0;
	bool chunk_overflow = false;
	for (uint32_t c0 = 0; c0 < count; c0 += 1024)
	{
		const uint32_t e = c0 + tid;
		uint32_t q = 0;
		if (e < count)
		{
			const uint32_t p = w.front[0][3 * e] & ~kTcFinal;
			const BvhView &b1 = views[w.pair_ik[2 * p]].points, &b2 = views[w.pair_ik[2 * p + 1]].tet_bvh;
			const uint32_t n1 = (uint32_t)b1.nodes[4 * w.front[0][3 * e + 1] + 3], n2 = (uint32_t)b2.nodes[4 * w.front[0][3 * e + 2] + 3];
			q = (n1 * n2 + 63u) / 64u;
		}
		uint32_t total;
		const uint32_t at = chunks + block_exclusive_scan(q, lds, total);
		if (e < count)
		{
			w.chunk_off[e] = at;
			for (uint32_t j = 0; j < q; j++) if (at + j < w.chunk_cap) w.chunk_pair[at + j] = e;
		}
		chunks += total;
		if (chunks > w.chunk_cap) chunk_overflow = true;
	}
	if (tid == 0)
	{
		w.chunk_off[count] = chunks;
		w.counters[kTcGenerations] = generation;
		if (chunk_overflow) w.counters[kTcStack] = 1u;
		else { w.counters[kTcLeafPairs] = count; w.counters[kTcChunks] = chunks; }
	}
}

// ---- candidates: one wavefront per chunk of 64 (point, tet) candidates of a leaf pair -------------------------------------------------
// kWrite = false: ballot of the candidates that are contacts.  kWrite = true: the contacts go to their place in the list.
template <bool kWrite>
__global__ __launch_bounds__(256) void tet_candidates_kernel(const TetColliderView *views, const P4 *pos, const P4 *rest, TetWork w, TetContact *contacts)
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
		const uint32_t p = w.front[0][3 * e] & ~kTcFinal, a = w.front[0][3 * e + 1], b = w.front[0][3 * e + 2];
		const TetColliderView &co1 = views[w.pair_ik[2 * p]], &co2 = views[w.pair_ik[2 * p + 1]];
		const uint32_t beg1 = (uint32_t)co1.points.nodes[4 * a + 2], n1 = (uint32_t)co1.points.nodes[4 * a + 3];
		const uint32_t beg2 = (uint32_t)co2.tet_bvh.nodes[4 * b + 2], n2 = (uint32_t)co2.tet_bvh.nodes[4 * b + 3];
		const uint32_t idx = (q - w.chunk_off[e]) * 64u + lane;
		bool hit = false;
		TetContact c;
		if (idx < n1 * n2 && (!kWrite || ((mask >> lane) & 1ull)))
			hit = tet_contact_candidate(co2, pos, rest, co1.points.lst[beg1 + idx / n2] + co1.first, co2.tet_bvh.lst[beg2 + idx % n2], c);
		if (!kWrite)
		{
			const unsigned long long m = __ballot(hit);
			if (lane == 0) { w.chunk_mask[q] = m; w.chunk_base[q] = (uint32_t)__popcll(m); }
		}
		else if (hit)
		{
			const uint32_t at = w.chunk_base[q] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
			if (at < kMaxTetContacts) contacts[at] = c;
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
		w.counters[kTcCount] = base < kMaxTetContacts ? base : kMaxTetContacts;
		if (base > kMaxTetContacts) w.counters[kTcOverflow] = 1u;
	}
}

// ---- levels: order-preserving decomposition of the sequential solve into sets of contacts on disjoint particles ----------------------
__global__ __launch_bounds__(1024) void tet_levels_kernel(const TetContact *contacts, TetWork w)
{
	__shared__ uint32_t lds[65];
	const uint32_t n = w.counters[kTcCount];
	const uint32_t tid = threadIdx.x;
	for (uint32_t c = tid; c < n; c += 1024) w.level_of[c] = 0xffffffffu;
	uint32_t done = 0, level = 0;
	while (done < n && level < kMaxTetLevels)
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
		if (done < n) w.counters[kTcOverflow] = 1u;
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
	TetPosAccess acc = { pos };
	const uint32_t levels = w.counters[kTcLevels];
	for (uint32_t l = 0; l < levels; l++)
	{
		const uint32_t end = w.level_start[l + 1];
		for (uint32_t i = w.level_start[l] + threadIdx.x; i < end; i += 1024) tet_contact_position_solve(contacts[w.order[i]], acc);
		__syncthreads();
	}
}

} // namespace pbdx
#endif
