// pbdx_access.h -- one constraint per lane: load indices / parameters / positions through an
// accessor, run the projection arithmetic of pbdx_project.h, write the corrections back.
//
// The same wrappers serve both device schedules of the engine:
//   GlobalAccess  per-colour launches; positions are float4 (x,y,z,invMass) in HBM/L2,
//                 32-bit particle indices
//   TileAccess    colour-fused tile launches; positions staged in LDS, 16-bit tile-local indices
// so the two schedules execute literally the same floating-point code.
#ifndef PBDX_ACCESS_H
#define PBDX_ACCESS_H

#include <hip/hip_runtime.h>
#include "pbdx_plan.h"
#include "pbdx_project.h"
#include "pbdx_bounds.h"
#ifndef PBDX_ST96
#define PBDX_ST96 1
#endif

namespace pbdx {

__device__ constexpr PlaneTable kPlanes = PlaneTable();

// parameter k of slot i: a scalar of the view, or one coalesced dword of plane kPlanes[..][k].
// `k` is a literal (or an unrolled loop index) at every call site, so the table lookups and the
// scalar/streamed decision fold away at compile time: no branch, every load unconditional.
// Stream access as  uniform 64-bit base (SGPR pair) + 32-bit byte offset (one VGPR shared by all planes
// of a slot): selects the `global_load v, v_off, s[base:base+1]` addressing form, so a slot costs one
// VALU shift instead of a 64-bit address computation per plane.
template <class T> __device__ __forceinline__ T ld_off(const void *base, uint32_t byte_off)
{
	return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}
template <class T> __device__ __forceinline__ void st_off(void *base, uint32_t byte_off, T v)
{
	*reinterpret_cast<T *>(static_cast<char *>(base) + byte_off) = v;
}

template <int TYPE, bool COMPACT>
__device__ __forceinline__ float param_get(const TypeView &v, const float *par, uint32_t stride, int k, uint32_t i)
{
	if (is_scalar_param(TYPE, COMPACT, k)) return v.u[k];
	return ld_off<float>(par + (size_t)((uint32_t)kPlanes.plane[COMPACT ? 1 : 0][TYPE][k] * stride), i * 4u);
}

template <int TYPE, bool COMPACT> struct GlobalAccess
{
	float4 *pos;
	const uint32_t *idx;
	const float *par;
	uint32_t par_stride;
	float *lambda;
	const TypeView &view;

	__device__ __forceinline__ uint2 idx2(uint32_t i) const { return reinterpret_cast<const uint2 *>(idx)[i]; }
	__device__ __forceinline__ uint4 idx4(uint32_t i) const { return reinterpret_cast<const uint4 *>(idx)[i]; }
	__device__ __forceinline__ float4 ld(uint32_t h) const { return pos[h]; }
	__device__ __forceinline__ void st(uint32_t h, float4 v) const { pos[h] = v; }
	__device__ __forceinline__ float p(int k, uint32_t i) const { return param_get<TYPE, COMPACT>(view, par, par_stride, k, i); }
	__device__ __forceinline__ bool sym() const { return COMPACT; }
	__device__ __forceinline__ float lam_load(uint32_t i) const { return ld_off<float>(lambda, i * 4u); }
	__device__ __forceinline__ void lam_store(uint32_t i, float v) const { st_off<float>(lambda, i * 4u, v); }
};

// The three streams of a fused segment (packed indices, parameter planes, multipliers) are read through
// buffer descriptors: address = descriptor base + SGPR offset (start of the run / of the plane) + one
// 32-bit VGPR byte offset per slot.  No per-plane VALU address arithmetic, no 64-bit address VGPRs,
// hardware bounds check against the stream size.
struct TileStreams
{
	__amdgpu_buffer_rsrc_t idx, par, lam;
#if PBDX_BOUNDS
	uint32_t dbg_n_local, dbg_tab_f4, dbg_tile;      // LDS slots of the tile's particles, size of its dictionary table, tile index (range checks of the debug build)
#endif
};

// COHERENT (persistent schedule): the multiplier stream is re-read by the same workgroup one iteration later
// INSIDE one launch; the CU's vector L1 may still hold the line from before the store (no kernel boundary
// invalidates it), so those loads go past the L1 (sc1: served by the XCD's L2, which does see the CU's own stores).
// VEC: form of the parameter stream (pbdx_plan.h param_float_index)
template <int TYPE, bool COMPACT, bool COHERENT = false, bool VEC = false> struct TileAccess
{
	float4 *pos;               // LDS
	const TileStreams &str;
	uint32_t idx_soff;         // BYTE offsets of the current chunk inside the streams (SGPRs)
	uint32_t par_soff;
	uint32_t lam_soff;
	uint32_t v_par;            // per-lane byte offsets inside a chunk's parameter block (pbdx_plan.h param_float_index): the lane's dword of plane 0, or
	                           // (VEC) its 16 bytes of segment 0
	uint32_t v_tail;           // (VEC) ... and its np % 4 floats of the tail segment
	const TypeView &view;

	// `i` = lane's slot inside the chunk (threadIdx.x): the per-lane offsets are the same for every chunk
	__device__ __forceinline__ uint32_t idx_raw1(uint32_t i) const { return __builtin_amdgcn_raw_buffer_load_b32(str.idx, (int)(i * 4u), (int)idx_soff, 0); }
	__device__ __forceinline__ uint2 idx_raw2(uint32_t i) const
	{
		typedef unsigned int v2u __attribute__((ext_vector_type(2)));
		const v2u v = __builtin_amdgcn_raw_buffer_load_b64(str.idx, (int)(i * 8u), (int)idx_soff, 0);
		return make_uint2(v.x, v.y);
	}
	__device__ __forceinline__ uint2 idx2(uint32_t i) const
	{
		const uint32_t v = idx_raw1(i);
		return make_uint2(v & 0xffffu, v >> 16);
	}
	__device__ __forceinline__ uint4 idx4(uint32_t i) const
	{
		const uint2 v = idx_raw2(i);
		return make_uint4(v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16);
	}
#if PBDX_BOUNDS
	__device__ __forceinline__ float4 ld(uint32_t h) const { return pos[PBDX_BCLAMP(kBndLdsSlot, h, str.dbg_n_local, str.dbg_tile)]; }
	__device__ __forceinline__ void st(uint32_t h, float4 v) const { if (PBDX_BOK(kBndLdsSlot, h, str.dbg_n_local, str.dbg_tile)) pos[h] = v; }
#else
	__device__ __forceinline__ float4 ld(uint32_t h) const { return pos[h]; }
#if PBDX_ST96
	// (A/B build: the inverse mass of a slot never changes -- a 12-byte store moves one dword less from the SIMD to the LDS, MI355X_MICROARCH.md LDS table)
	__device__ __forceinline__ void st(uint32_t h, float4 v) const
	{
		typedef float f3 __attribute__((ext_vector_type(3)));      // (16-byte aligned, 12 bytes stored: ds_write_b96)
		f3 w; w.x = v.x; w.y = v.y; w.z = v.z;
		*reinterpret_cast<f3 *>(pos + h) = w;
	}
#else
	__device__ __forceinline__ void st(uint32_t h, float4 v) const { pos[h] = v; }
#endif
#endif
	__device__ __forceinline__ float p(int k, uint32_t) const
	{
		if (is_scalar_param(TYPE, COMPACT, k)) return view.u[k];
		const uint32_t plane = (uint32_t)kPlanes.plane[COMPACT ? 1 : 0][TYPE][k];
		// the segment / component offsets are compile-time constants: folded into the instruction's immediate offset
		return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(str.par, (int)plane_voff_c(plane), (int)par_soff, 0));
	}
	static constexpr uint32_t kNP = (uint32_t)num_planes(TYPE, COMPACT), kFull = kNP / 4u, kTail = kNP % 4u;
	__device__ __forceinline__ uint32_t plane_voff_c(uint32_t plane) const      // `plane` a constant
	{
		if constexpr (!VEC) return v_par + plane * 256u;
		return plane < 4u * kFull ? v_par + (plane / 4u) * 1024u + (plane % 4u) * 4u : v_tail + (plane - 4u * kFull) * 4u;
	}
	// the same for a plane only known at run time (quad-lane FEM records: a lane fetches ITS column / row of Dm^-1)
	__device__ __forceinline__ uint32_t plane_voff(uint32_t plane) const
	{
		if constexpr (!VEC) return v_par + plane * 256u;
		return plane < 4u * kFull ? v_par + (plane >> 2) * 1024u + (plane & 3u) * 4u : v_tail + (plane - 4u * kFull) * 4u;
	}
	// ALL streamed planes of the lane's slot, w[plane]: one dword load per plane, or (VEC) one 16-byte load per full segment and one load of the tail's width
	__device__ __forceinline__ void par_planes(uint32_t *w) const
	{
		if constexpr (!VEC)
		{
#pragma unroll
			for (uint32_t pl = 0; pl < kNP; pl++) w[pl] = __builtin_amdgcn_raw_buffer_load_b32(str.par, (int)(v_par + pl * 256u), (int)par_soff, 0);
			return;
		}
		typedef unsigned int v4u __attribute__((ext_vector_type(4)));
		typedef unsigned int v3u __attribute__((ext_vector_type(3)));
		typedef unsigned int v2u __attribute__((ext_vector_type(2)));
#pragma unroll
		for (uint32_t sgm = 0; sgm < kFull; sgm++)
		{
			const v4u v = __builtin_amdgcn_raw_buffer_load_b128(str.par, (int)(v_par + sgm * 1024u), (int)par_soff, 0);
			w[4 * sgm] = v.x; w[4 * sgm + 1] = v.y; w[4 * sgm + 2] = v.z; w[4 * sgm + 3] = v.w;
		}
		if constexpr (kTail == 1u) w[4 * kFull] = __builtin_amdgcn_raw_buffer_load_b32(str.par, (int)v_tail, (int)par_soff, 0);
		if constexpr (kTail == 2u) { const v2u v = __builtin_amdgcn_raw_buffer_load_b64(str.par, (int)v_tail, (int)par_soff, 0); w[4 * kFull] = v.x; w[4 * kFull + 1] = v.y; }
		if constexpr (kTail == 3u) { const v3u v = __builtin_amdgcn_raw_buffer_load_b96(str.par, (int)v_tail, (int)par_soff, 0); w[4 * kFull] = v.x; w[4 * kFull + 1] = v.y; w[4 * kFull + 2] = v.z; }
	}
	// raw dword of the chunk's parameter block at a per-lane byte offset (quad-lane records, pbdx_quad.h: every lane fetches its own planes)
	__device__ __forceinline__ uint32_t par_raw(uint32_t voff) const { return __builtin_amdgcn_raw_buffer_load_b32(str.par, (int)voff, (int)par_soff, 0); }
	__device__ __forceinline__ bool sym() const { return COMPACT; }
	__device__ __forceinline__ float lam_load(uint32_t i) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(str.lam, (int)(i * 4u), (int)lam_soff, COHERENT ? 16 : 0)); }
	__device__ __forceinline__ void lam_store(uint32_t i, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), str.lam, (int)(i * 4u), (int)lam_soff, 0); }
};

template <class A> __device__ __forceinline__ void ldp(const A &a, uint32_t h, V3 &p, float &w)
{
	const float4 v = a.ld(h);
	p = mk(v.x, v.y, v.z); w = v.w;
}
// Constraints.cpp:1198-1204: corrections are added only to dynamic particles
// (PBDX_LIKELY_DYNAMIC: nearly every particle is dynamic -- the hint keeps the store on the fall-through path instead of in an out-of-line block that
// costs two taken branches per endpoint)
#ifndef PBDX_LIKELY_DYNAMIC
#define PBDX_LIKELY_DYNAMIC 1
#endif
#if PBDX_LIKELY_DYNAMIC
#define PBDX_LIKELY(x) __builtin_expect(!!(x), 1)
#else
#define PBDX_LIKELY(x) (x)
#endif
template <class A> __device__ __forceinline__ void apply(const A &a, uint32_t h, V3 p, V3 c, float w)
{
	if (PBDX_LIKELY(w != 0.0f))
		a.st(h, make_float4(p.x + c.x, p.y + c.y, p.z + c.z, w));
}

struct QFull
{
	float q[16];   // column-major Q(j,k) = q[k*4+j]
	__device__ __forceinline__ float operator()(int j, int k) const { return q[k * 4 + j]; }
};

// Q of the isometric bending constraints, column-major; in the compact layout only the upper
// triangle is streamed and the lower one mirrored (see pbdx_plan.h).
template <class A> __device__ __forceinline__ void load_q(const A &a, uint32_t i, QFull &q)
{
#pragma unroll
	for (int c = 0; c < 4; c++)
#pragma unroll
		for (int r = 0; r <= c; r++) q.q[c * 4 + r] = a.p(1 + c * 4 + r, i);
#pragma unroll
	for (int c = 0; c < 4; c++)
#pragma unroll
		for (int r = c + 1; r < 4; r++)
		{
			if (a.sym()) q.q[c * 4 + r] = q.q[r * 4 + c];
			else q.q[c * 4 + r] = a.p(1 + c * 4 + r, i);
		}
}

template <class A> __device__ __forceinline__ M3 load_m3(const A &a, int first, uint32_t i)
{
	M3 R;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int r = 0; r < 3; r++) R.m[r][c] = a.p(first + c * 3 + r, i);
	return R;
}

#define PBDX_LOAD4 \
	const uint4 id = a.idx4(i); \
	V3 p0, p1, p2, p3; float w0, w1, w2, w3; \
	ldp(a, id.x, p0, w0); ldp(a, id.y, p1, w1); ldp(a, id.z, p2, w2); ldp(a, id.w, p3, w3); \
	V3 c0, c1, c2, c3
#define PBDX_APPLY4 \
	apply(a, id.x, p0, c0, w0); apply(a, id.y, p1, c1, w1); apply(a, id.z, p2, c2, w2); apply(a, id.w, p3, c3, w3)
#define PBDX_LOAD3 \
	const uint4 id = a.idx4(i); \
	V3 p0, p1, p2; float w0, w1, w2; \
	ldp(a, id.x, p0, w0); ldp(a, id.y, p1, w1); ldp(a, id.z, p2, w2); \
	V3 c0, c1, c2
#define PBDX_APPLY3 \
	apply(a, id.x, p0, c0, w0); apply(a, id.y, p1, c1, w1); apply(a, id.z, p2, c2, w2)

// `dt` = substep size (XPBD compliance); `first_iter`: iteration 0 of a substep, lambda := 0
// without reading it (Constraints.cpp:1241,1448,1725,1877).
template <int TYPE, class A> struct Project;

template <class A> struct Project<PBDX_DISTANCE, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		const uint2 id = a.idx2(i);
		V3 p0, p1; float w0, w1;
		ldp(a, id.x, p0, w0); ldp(a, id.y, p1, w1);
		V3 c0, c1;
		if (PBDX_LIKELY(solve_distance(p0, w0, p1, w1, a.p(0, i), a.p(1, i), c0, c1)))
		{
			apply(a, id.x, p0, c0, w0); apply(a, id.y, p1, c1, w1);
		}
	}
};

template <class A> struct Project<PBDX_DISTANCE_XPBD, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float dt, int first_iter)
	{
		const uint2 id = a.idx2(i);
		V3 p0, p1; float w0, w1;
		ldp(a, id.x, p0, w0); ldp(a, id.y, p1, w1);
		float lambda = first_iter ? 0.0f : a.lam_load(i);
		V3 c0, c1;
		// (tried in round 4, profiles/HISTORY.md [8]: a wave-uniform fast path -- ballot "no active lane takes one of the two degenerate exits", then
		// straight-line arithmetic without the twelve v_mov and the exec-mask bookkeeping the per-lane selects cost: -0.9 % on the 1 M cloth, nothing
		// elsewhere, and one crash of the test process that was not chased down; not kept)
		if (PBDX_LIKELY(solve_distance_xpbd(p0, w0, p1, w1, a.p(0, i), a.p(1, i), dt, lambda, c0, c1)))
		{
			apply(a, id.x, p0, c0, w0); apply(a, id.y, p1, c1, w1);
		}
		a.lam_store(i, lambda);
	}
};

template <class A> struct Project<PBDX_DIHEDRAL, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		PBDX_LOAD4;
		if (PBDX_LIKELY(solve_dihedral(p0, w0, p1, w1, p2, w2, p3, w3, a.p(0, i), a.p(1, i), c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
	}
};

template <class A> struct Project<PBDX_ISOMETRIC_BENDING, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		PBDX_LOAD4;
		QFull q;
		load_q(a, i, q);
		if (PBDX_LIKELY(solve_isometric_bending(p0, w0, p1, w1, p2, w2, p3, w3, q, a.p(0, i), c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
	}
};

template <class A> struct Project<PBDX_ISOMETRIC_BENDING_XPBD, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float dt, int first_iter)
	{
		PBDX_LOAD4;
		QFull q;
		load_q(a, i, q);
		float lambda = first_iter ? 0.0f : a.lam_load(i);
		if (PBDX_LIKELY(solve_isometric_bending_xpbd(p0, w0, p1, w1, p2, w2, p3, w3, q, a.p(0, i), dt, lambda, c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
		a.lam_store(i, lambda);
	}
};

template <class A> struct Project<PBDX_FEM_TRIANGLE, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		PBDX_LOAD3;
		float im[2][2];
		im[0][0] = a.p(1, i); im[1][0] = a.p(2, i); im[0][1] = a.p(3, i); im[1][1] = a.p(4, i);
		if (PBDX_LIKELY(solve_fem_triangle(p0, w0, p1, w1, p2, w2, a.p(0, i), im, a.p(5, i), a.p(6, i), a.p(7, i), a.p(8, i), a.p(9, i), c0, c1, c2)))
		{
			PBDX_APPLY3;
		}
	}
};

template <class A> struct Project<PBDX_STRAIN_TRIANGLE, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		PBDX_LOAD3;
		float im[2][2];
		im[0][0] = a.p(0, i); im[1][0] = a.p(1, i); im[0][1] = a.p(2, i); im[1][1] = a.p(3, i);
		if (PBDX_LIKELY(solve_strain_triangle(p0, w0, p1, w1, p2, w2, im, a.p(4, i), a.p(5, i), a.p(6, i), a.p(7, i) != 0.0f, a.p(8, i) != 0.0f, c0, c1, c2)))
		{
			PBDX_APPLY3;
		}
	}
};

template <class A> struct Project<PBDX_VOLUME, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		PBDX_LOAD4;
		if (PBDX_LIKELY(solve_volume(p0, w0, p1, w1, p2, w2, p3, w3, a.p(0, i), a.p(1, i), c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
	}
};

template <class A> struct Project<PBDX_VOLUME_XPBD, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float dt, int first_iter)
	{
		PBDX_LOAD4;
		float lambda = first_iter ? 0.0f : a.lam_load(i);
		if (PBDX_LIKELY(solve_volume_xpbd(p0, w0, p1, w1, p2, w2, p3, w3, a.p(0, i), a.p(1, i), dt, lambda, c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
		a.lam_store(i, lambda);
	}
};

template <class A> struct Project<PBDX_FEM_TET, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		PBDX_LOAD4;
		const float vol = a.p(0, i);
		const M3 im = load_m3(a, 1, i);
		const bool hi = fem_tet_handle_inversion(p0, p1, p2, p3, vol);
		if (PBDX_LIKELY(solve_fem_tet(p0, w0, p1, w1, p2, w2, p3, w3, vol, im, a.p(10, i), a.p(11, i), hi, c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
	}
};

template <class A> struct Project<PBDX_FEM_TET_XPBD, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float dt, int first_iter)
	{
		PBDX_LOAD4;
		const float vol = a.p(0, i);
		const M3 im = load_m3(a, 1, i);
		const bool hi = fem_tet_handle_inversion(p0, p1, p2, p3, vol);
		float lambda = first_iter ? 0.0f : a.lam_load(i);
		if (PBDX_LIKELY(solve_fem_tet_xpbd(p0, w0, p1, w1, p2, w2, p3, w3, vol, im, a.p(10, i), a.p(11, i), hi, dt, lambda, c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
		a.lam_store(i, lambda);
	}
};

template <class A> struct Project<PBDX_STRAIN_TET, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		PBDX_LOAD4;
		const M3 im = load_m3(a, 0, i);
		if (PBDX_LIKELY(solve_strain_tet(p0, w0, p1, w1, p2, w2, p3, w3, im, a.p(9, i), a.p(10, i), a.p(11, i) != 0.0f, a.p(12, i) != 0.0f, c0, c1, c2, c3)))
		{
			PBDX_APPLY4;
		}
	}
};

template <class A> struct Project<PBDX_SHAPE_MATCHING, A>
{
	static __device__ __forceinline__ void run(const A &a, uint32_t i, float, int)
	{
		const uint4 id = a.idx4(i);
		const uint32_t ids[4] = { id.x, id.y, id.z, id.w };
		V3 x[4], x0[4], corr[4]; float wl[4], w[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			ldp(a, ids[k], x[k], wl[k]);
			x0[k] = mk(a.p(4 + 3 * k, i), a.p(5 + 3 * k, i), a.p(6 + 3 * k, i));
			w[k] = a.p(16 + k, i);
		}
		const V3 restCm = mk(a.p(1, i), a.p(2, i), a.p(3, i));
		if (PBDX_LIKELY(solve_shape_matching4(x0, x, w, restCm, a.p(0, i), corr)))
		{
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				// (1.0 / m_numClusters[i]) * m_corr[i]   Constraints.cpp:2024 (double quotient narrowed to Real)
				const float f = (float)(1.0 / (double)(unsigned int)a.p(20 + k, i));
				if (w[k] != 0.0f)
					a.st(ids[k], make_float4(x[k].x + f * corr[k].x, x[k].y + f * corr[k].y, x[k].z + f * corr[k].z, wl[k]));
			}
		}
	}
};

// ---- prefetched records ------------------------------------------------------------------------
// The fused kernel separates "fetch everything a constraint streams from HBM" (indices, parameter
// record, multiplier) from "project it on the LDS-resident positions", so that the fetches of
// several slots are in flight before the first projection starts.
__device__ constexpr bool kTwoBodies[PBDX_NUM_CONSTRAINT_TYPES] = { true, true, false, false, false, false, false, false, false, false, false, false, false };
__device__ constexpr bool kHasLambda[PBDX_NUM_CONSTRAINT_TYPES] = { false, true, false, false, true, false, false, false, true, false, true, false, false };

// A record keeps exactly what came back from memory (packed 16-bit indices, raw multiplier): no
// instruction may touch a prefetched value before its slot is projected, otherwise the compiler
// has to drain the whole prefetch queue (s_waitcnt vmcnt(0)) at that instruction.
// Only the STREAMED parameters of the layout are kept (plane order); scalars of the compact layout are
// read from the view at use (loop-invariant for the compiler: e.g. the XPBD compliance 1/(k dt^2) is
// hoisted out of the slot loop), mirrored Q entries come from their upper-triangle twin.
template <int TYPE, bool COMPACT> struct Rec
{
	// homogeneous raw dwords exactly as they came back from memory:
	// [0],[1] packed 16-bit indices, [2] multiplier, [3 + p] streamed parameter plane p
	uint32_t w[3 + num_planes(TYPE, COMPACT)];
	__device__ __forceinline__ float plane(int p) const { return __builtin_bit_cast(float, w[3 + p]); }
	__device__ __forceinline__ float lambda() const { return __builtin_bit_cast(float, w[2]); }
};

template <int TYPE, bool COMPACT, class A> struct RecAccess
{
	const A &base;
	const Rec<TYPE, COMPACT> &r;
	int first_iter;
	__device__ __forceinline__ uint2 idx2(uint32_t) const { return make_uint2(r.w[0] & 0xffffu, r.w[0] >> 16); }
	__device__ __forceinline__ uint4 idx4(uint32_t) const { return make_uint4(r.w[0] & 0xffffu, r.w[0] >> 16, r.w[1] & 0xffffu, r.w[1] >> 16); }
	__device__ __forceinline__ float4 ld(uint32_t h) const { return base.ld(h); }
	__device__ __forceinline__ void st(uint32_t h, float4 v) const { base.st(h, v); }
	__device__ __forceinline__ float p(int k, uint32_t) const
	{
		if (is_scalar_param(TYPE, COMPACT, k)) return base.view.u[k];
		if (is_mirrored_q(TYPE, COMPACT, k))
		{
			const int c = (k - 1) / 4, rr = (k - 1) % 4;            // Q(rr, c) with rr > c  ->  Q(c, rr)
			return r.plane(kPlanes.plane[1][TYPE][1 + rr * 4 + c]);
		}
		return r.plane(kPlanes.plane[COMPACT ? 1 : 0][TYPE][k]);
	}
	__device__ __forceinline__ bool sym() const { return false; }      // p() resolves the mirrored entries itself
	__device__ __forceinline__ float lam_load(uint32_t) const { return r.lambda(); }
	__device__ __forceinline__ void lam_store(uint32_t i, float v) const { base.lam_store(i, v); }
};

// only for TileAccess (packed 16-bit indices)
template <int TYPE, bool COMPACT, class A> __device__ __forceinline__ void load_rec(const A &a, uint32_t i, Rec<TYPE, COMPACT> &r)
{
	if constexpr (kTwoBodies[TYPE]) { r.w[0] = a.idx_raw1(i); r.w[1] = 0u; }
	else { const uint2 v = a.idx_raw2(i); r.w[0] = v.x; r.w[1] = v.y; }
	a.par_planes(r.w + 3);        // (w[3 + plane]: the streamed parameters in plane order)
	r.w[2] = 0u;
	// unconditional load (the stream always exists; iteration 0 ignores the value): branch-free prefetch
	if constexpr (kHasLambda[TYPE]) r.w[2] = __builtin_bit_cast(uint32_t, a.lam_load(i));
}

template <int TYPE, bool COMPACT, class A> __device__ __forceinline__ void exec_rec(const A &a, const Rec<TYPE, COMPACT> &r, uint32_t i, float dt, int first_iter)
{
	const RecAccess<TYPE, COMPACT, A> ra = { a, r, first_iter };
	Project<TYPE, RecAccess<TYPE, COMPACT, A>>::run(ra, i, dt, first_iter);
}

} // namespace pbdx

#endif
