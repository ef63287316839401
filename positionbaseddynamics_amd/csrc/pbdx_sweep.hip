// pbdx_sweep.hip -- the constraint-sweep kernels of the device engine (schedules A, A' and B of pbdx_solver.hip) for gfx950.
//
// Replaces positionConstraintProjection (Simulation/TimeStepController.cpp:251-295) of the reference: the per-constraint code is pbdx_project.h behind the
// accessors of pbdx_access.h; this file holds the three ways of running it -- one launch per (colour, type) batch, one launch per segment of colour-fused LDS
// tiles, and all passes of a substep as ONE persistent launch -- and the selectors the host engine picks a kernel with (pbdx_sweep.h).
#include <hip/hip_runtime.h>
#include "pbdx_sweep.h"
#include "pbdx_device.h"
#include "pbdx_access.h"
#include "pbdx_quad.h"
#include <type_traits>
#include <string.h>

namespace pbdx {
namespace {


// ------------------------------------------------------------------------------------------------
// (B) per-colour kernels
// ------------------------------------------------------------------------------------------------

// blockIdx -> logical block: with xcd_remap the 8 XCDs (hardware places block b on XCD b%8) each
// walk one contiguous eighth of the batch / of the tile list, so the particles a chiplet touches
// stay the same from launch to launch (per-XCD L2 locality); speed only, never correctness.

__device__ __forceinline__ uint32_t logical_block(uint32_t num_blocks, int xcd_remap)
{
	const uint32_t b = blockIdx.x;
	if (!xcd_remap || num_blocks < 16)
		return b;
	const uint32_t per = num_blocks >> 3;          // full blocks per XCD
	const uint32_t body = per << 3;
	if (b >= body)
		return b;                                    // remainder blocks keep their place at the end
	return (b & 7u) * per + (b >> 3);
}

template <int TYPE, bool COMPACT>
__global__ __launch_bounds__(256) void project_kernel(BatchArgs a)
{
	const uint32_t i = logical_block(a.num_blocks, a.xcd_remap) * blockDim.x + threadIdx.x;
	if (i < a.count)
	{
		const GlobalAccess<TYPE, COMPACT> acc = { a.pos, a.idx, a.par, a.par_stride, a.lambda, a.view };
		Project<TYPE, GlobalAccess<TYPE, COMPACT>>::run(acc, i, a.dt, a.first_iter);
	}
}

#define PBDX_PK(T) { project_kernel<T, false>, project_kernel<T, true> }
const project_fn kProjectKernels[PBDX_NUM_CONSTRAINT_TYPES][2] = {
	PBDX_PK(PBDX_DISTANCE), PBDX_PK(PBDX_DISTANCE_XPBD), PBDX_PK(PBDX_DIHEDRAL),
	PBDX_PK(PBDX_ISOMETRIC_BENDING), PBDX_PK(PBDX_ISOMETRIC_BENDING_XPBD),
	PBDX_PK(PBDX_FEM_TRIANGLE), PBDX_PK(PBDX_STRAIN_TRIANGLE),
	PBDX_PK(PBDX_VOLUME), PBDX_PK(PBDX_VOLUME_XPBD),
	PBDX_PK(PBDX_FEM_TET), PBDX_PK(PBDX_FEM_TET_XPBD), PBDX_PK(PBDX_STRAIN_TET),
	PBDX_PK(PBDX_SHAPE_MATCHING),
};

// ------------------------------------------------------------------------------------------------
// (A) colour-fused tile kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rfl(uint32_t v);
// one dword of a read-only array at a wave-uniform index, as a scalar load
__device__ __forceinline__ uint32_t sload_u32(const uint32_t *p, uint32_t index)
{
#if defined(__HIP_DEVICE_COMPILE__)
	typedef __attribute__((address_space(4))) const uint32_t *const_ptr;
	return rfl(*((const_ptr)(uintptr_t)p + index));
#else
	return p[index];
#endif
}
__device__ __forceinline__ FusedTile load_tile(const TileDev *tiles, uint32_t index)
{
	FusedTile t;
#if defined(__HIP_DEVICE_COMPILE__)
	typedef unsigned int u16v __attribute__((ext_vector_type(16)));
	typedef __attribute__((address_space(4))) const u16v *const_ptr16;
	const u16v v = *((const_ptr16)(uintptr_t)tiles + index);
	t.step_begin = rfl(v[0]); t.step_end = rfl(v[1]); t.n_local = rfl(v[2]); t.n_owned = rfl(v[3]); t.gid_off = rfl(v[4]); t.slots = rfl(v[5]);
	t.chunk_begin = rfl(v[6]); t.chunk_end = rfl(v[7]); t.tab_off = rfl(v[8]); t.tab_f4 = rfl(v[9]); t.wb_begin = rfl(v[10]);
#else
	t = tiles[index].t;
#endif
	return t;
}
struct BndLim { uint32_t gid_left, n_particles, lds_f4, tile, ids_cap; };      // what a tile's raw accesses are checked against (PBDX_BOUNDS builds)
// what a run needs besides the streams
struct RunArgs
{
	float dt;
	int first_iter;
	const TypeView *views;
};

// ---- software pipeline over the chunks of a tile --------------------------------------------------
// A tile's steps are expanded on the host into "chunks": one workgroup-wide pass in which lane l projects
// slot k * BLOCK + l of a step (FusedChunk, pbdx_plan.h).  The record of a slot (indices, parameter
// planes, multiplier) only depends on read-only streams, so it may be fetched long before the positions it
// will be applied to are final: every thread keeps a ring of D records and fetches the chunk D positions
// ahead -- across colour barriers -- while it projects the current one.  This decouples the HBM latency of
// the streams from the barrier-synchronised colour sweep.  The pipeline runs over maximal runs of chunks of
// one constraint type.
//
// Everything address-like is either a per-lane constant of the run (VGPR offsets: lane * 4 / * 8 and the
// wave-tiled parameter offset) or a scalar of the chunk descriptor (SGPR stream offsets), and the plane
// offset is an instruction immediate: a fetch costs no VALU and a handful of SALU instructions.  Loads are
// unconditional (lanes beyond a chunk's valid count read neighbouring or out-of-range stream bytes, which the
// buffer descriptor turns into zeros, and are never projected), which lets the compiler wait with an exact
// s_waitcnt vmcnt(N) instead of draining every outstanding prefetch.

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

struct ChunkS { uint32_t info, idx_boff, par_boff, lam_boff, f_idx_boff, f_par_boff, f_lam_boff; };
// A tile's chunk descriptors are read from the plan through the SCALAR cache (constant address space + uniform index -> s_load straight into
// SGPRs).  Until round 4 they were staged in LDS by the fill and read back as scalars (uniform address -> broadcast read -> four
// v_readfirstlane): 8 VALU instructions per sub-iteration of a sweep that is VALU-issue-bound.  Measured (profiles/r04g_*): 1 M cloth 0.654 ->
// 0.628 ms (-4 %), 100 k-tet bar 0.609 -> 0.595, configs[3] block 1.63 -> 1.58, 300x300 cloth -5 %; bit-identical.  The kernels invalidate the
// scalar cache when they start (a plan that is rebuilt may reuse the addresses of the old one).
struct ChunkSrc
{
	const uint4 *lds; const uint4 *glb;      // (glb: the tile's 32-byte descriptors in the plan)
#if PBDX_BOUNDS
	uint32_t n;              // the tile's number of chunk descriptors
#endif
};
__device__ __forceinline__ uint4 chunk_words(const ChunkSrc &cs, uint32_t c)
{
#if PBDX_BOUNDS
	c = PBDX_BCLAMP(kBndChunk, c, cs.n, 0u);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
	typedef unsigned int u4 __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(4))) const u4 *const_ptr;
	const u4 v = *((const_ptr)(uintptr_t)cs.glb + 2u * c);      // (the first half of the 32-byte descriptor)
	return make_uint4(v.x, v.y, v.z, v.w);
#else
	return cs.lds[c];
#endif
}
__device__ __forceinline__ ChunkS load_chunk(const ChunkSrc &cs, uint32_t c)
{
#if PBDX_BOUNDS
	c = PBDX_BCLAMP(kBndChunk, c, cs.n, 0u);
#endif
	ChunkS r;
#if defined(__HIP_DEVICE_COMPILE__)
	typedef unsigned int u8 __attribute__((ext_vector_type(8)));
	typedef __attribute__((address_space(4))) const u8 *const_ptr8;
	const u8 v = *((const_ptr8)(uintptr_t)cs.glb + c);           // one s_load_dwordx8
	r.info = rfl(v[0]); r.idx_boff = rfl(v[1]); r.par_boff = rfl(v[2]); r.lam_boff = rfl(v[3]);
	r.f_idx_boff = rfl(v[4]); r.f_par_boff = rfl(v[5]); r.f_lam_boff = rfl(v[6]);
#else
	r = ChunkS();
#endif
	return r;
}
__device__ __forceinline__ uint32_t chunk_type(uint32_t info) { return info & 0x3fu; }
__device__ __forceinline__ bool chunk_barrier(uint32_t info) { return (info >> 6) & 1u; }
__device__ __forceinline__ bool chunk_last_of_step(uint32_t info) { return (info >> 7) & 1u; }
__device__ __forceinline__ uint32_t chunk_valid(uint32_t info) { return (info >> 8) & 0x7ffu; }
__device__ __forceinline__ uint32_t chunk_run_left(uint32_t info) { return info >> 19; }

// HAZARD (root cause of the intermittent "Memory access fault" of rounds 3-5, profiles/HISTORY.md [9]): on gfx9-family hardware a vector-memory instruction
// that reads an SGPR written by a VALU instruction needs FIVE wait states in between.  The compiler inserts them for its own instructions
// (GCNHazardRecognizer) but cannot see inside an inline-assembly string: in kernels that spill SGPRs to vector-register lanes the base pointer of the
// hand-written copies / stores below is restored with v_readlane_b32 (a VALU write of an SGPR) immediately in front of the statement, the memory
// instruction then reads the OLD register contents -- a garbage address -- and whether it does depends on what else the SIMD issues in between (one wave
// per SIMD in the 256-thread kernels: nothing).  Every hand-written memory instruction with a scalar operand therefore starts with `s_nop 4`.
// HBM -> LDS copy of 16 bytes per lane without a register in between (global_load_lds_dwordx4: lane l of the wave
// lands at lds_wave_base + 16 l).  Issued as inline assembly on purpose: with the builtin the compiler drains
// vmcnt to 0 in front of every such copy (it cannot order them against the other outstanding loads), which
// serialises the batch.  The copies are therefore invisible to the compiler's wait-count bookkeeping; that is
// safe because (a) vmcnt retires in order, so its own waits can only become stricter, and (b) lds_dma_wait()
// drains everything before the barrier that publishes the tile.
// COHERENT (persistent schedule, positions handed from tile to tile inside one launch): agent-scope `sc1` on
// both sides -- sc1 stores are written through, sc1 loads are served past the CU's L1 (MI355X_MICROARCH: "16 B
// sc1 stores AND sc1 loads" is a valid cross-XCD hand-off).
template <bool COHERENT>
__device__ __forceinline__ void lds_dma16(const float4 *base, uint32_t index, float4 *lds_wave_base)
{
	const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)lds_wave_base);
	const uint32_t boff = index * 16u;        // scalar base + 32-bit lane offset: one address register per copy
	// (the base as a SCALAR pair whatever the compiler knows about its uniformity: kernel-argument pointers fold to themselves, a pointer derived from a
	// tile descriptor is read from the first lane)
	const unsigned long long bv = (unsigned long long)(uintptr_t)base;
	const unsigned long long sbase = ((unsigned long long)rfl((uint32_t)(bv >> 32)) << 32) | (unsigned long long)rfl((uint32_t)bv);      // (rfl returns uint32_t: the builtin's int would sign-extend the low half)
	uint32_t saved;      // M0 is a reserved register: preserved around the copy
	if constexpr (COHERENT)
		asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
			: "=&s"(saved) : "v"(boff), "s"(sbase), "s"(m0v) : "memory");
	else
		asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
			: "=&s"(saved) : "v"(boff), "s"(sbase), "s"(m0v) : "memory");
}
// write-back store of one position
template <bool COHERENT>
__device__ __forceinline__ void store_pos(float4 *base, uint32_t index, float4 v)
{
	if constexpr (COHERENT)
	{
		typedef float f4 __attribute__((ext_vector_type(4)));
		f4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
		// (the base as an explicitly scalar pair: a uniform pointer the compiler happens to keep in vector registers would otherwise be printed into the
		// scalar operand as it is -- "invalid operand for instruction")
		const unsigned long long bv = (unsigned long long)(uintptr_t)base;
		const unsigned long long sbase = ((unsigned long long)rfl((uint32_t)(bv >> 32)) << 32) | (unsigned long long)rfl((uint32_t)bv);      // (rfl returns uint32_t: the builtin's int would sign-extend the low half)
		asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1" :: "v"(index * 16u), "v"(w), "s"(sbase) : "memory");
	}
	else
		base[index] = v;
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS fill of a tile.  gid -> position is a dependent
// pair of HBM round trips; written as a plain loop every thread pays that pair once per particle it stages
// (7 x 2 serialised latencies for a 7 000-particle tile).  The loads are issued in batches instead: the chunk
// descriptor, then kFillBatch particle ids, then their kFillBatch positions -- two exposed latencies per batch,
// one batch for most tiles.
// Tried at the pass boundaries and removed (all bit-identical, all slower; the A/B logs are in profiles/):
//  * returning from the fill once the chunk descriptors are visible and waiting for its HBM -> LDS copies behind the first run's ring priming
//    (r03k_deferred_fill_wait_ab.log): the first colour step of a pass drops from 3.2-3.8 to 2.2 us, the fill grows from 3.5 to 4.4-5.6 us (the record
//    fetches queue in front of the copies' completion) and a barrier is added: 0.777-0.787 against 0.759-0.768 ms per substep;
//  * requesting the NEXT pass's first particle ids and chunk descriptor before this pass's write-back stores (r03o_fill_ahead_ab.log): the fill shrank
//    (3.5 -> 2.9-3.5 us) but the write-back grew by more (2.2 -> 3.5-7.4 us: the requests queue in front of the stores), 1 M cloth 0.769 -> 0.788 ms;
//  * requesting the prefetch ring of a pass's first run DURING the fill, behind the particle ids (r03u_ring_primed_in_fill_ab.log): the 1 024-thread cloth
//    kernel has no registers to carry the ring across the fill (79 spilled, and a spill of a requested value waits for its load: fill 3.5 -> 12 us); on the
//    FEM bar, where registers are free, the first step gains 0.24 us and the fill loses 0.3.
template <int BLOCK, bool COHERENT> struct TileFill
{
	const uint4 *src;            // the tile's chunk descriptors in the plan
	const uint32_t *gid;
	const float4 *pos_in;
	uint4 *lchunks;
	float4 *lpos;
	uint32_t num_chunks, n_local;
	uint32_t first;              // local particles [0, first) are already in LDS (multiple of 64; 0 = stage everything)
	unsigned long long *trace;
	BndLim lim;                  // (PBDX_BOUNDS builds; zeros and unused otherwise)
	const uint32_t *lids;        // LDS copy of gid[first .. n_local) (persistent schedule, requested during the previous pass: LdsIds) or null

	// Eight particles per thread and batch.  The positions go from HBM straight into LDS (lds_dma16), so a
	// batch holds eight ids in registers and nothing else.  All eight ids are consumed by one empty asm statement:
	// the compiler waits for them once and places no wait (stricter than necessary, see lds_dma16) between the copies.
	// `wait` runs after the first batch of ids is in flight and before any position is read: the persistent
	// schedule waits for the neighbouring tiles there (the ids do not depend on them).
	// `extra` runs once the copies have been issued and before they are waited for: work that only has to be complete at the fill's closing barrier (the
	// dictionary table into LDS, the request of the next pass's halo ids) overlaps with the copies' latency instead of taking a phase and a barrier of its own
	template <class Wait, class Extra> __device__ __forceinline__ void operator()(const Wait &wait, const Extra &extra) const
	{
		static_assert(BLOCK >= (int)kMaxTileChunks, "one chunk descriptor per thread");
		const uint32_t last = n_local - 1u;
		uint32_t base = first + threadIdx.x;
#if PBDX_BOUNDS
#define PBDX_G(k) const uint32_t i##k = base + k * BLOCK; const uint32_t g##k = lids ? lids[PBDX_BCLAMP(kBndLdsIds, (i##k < last ? i##k : last) - first, lim.ids_cap, lim.tile)] \
	: gid[PBDX_BCLAMP(kBndGid, (i##k < last ? i##k : last), lim.gid_left, lim.tile)];
#define PBDX_D(k) if (i##k < n_local && PBDX_BOK(kBndParticle, g##k, lim.n_particles, lim.tile) && PBDX_BOK(kBndLdsFill, i##k, lim.lds_f4, lim.tile)) lds_dma16<COHERENT>(pos_in, g##k, lpos + (i##k & ~63u));
#else
#define PBDX_G(k) const uint32_t i##k = base + k * BLOCK; const uint32_t g##k = lids ? lids[(i##k < last ? i##k : last) - first] : gid[i##k < last ? i##k : last];
#define PBDX_D(k) if (i##k < n_local) lds_dma16<COHERENT>(pos_in, g##k, lpos + (i##k & ~63u));
#endif
#define PBDX_BATCH(BETWEEN) { \
			PBDX_G(0) PBDX_G(1) PBDX_G(2) PBDX_G(3) PBDX_G(4) PBDX_G(5) PBDX_G(6) PBDX_G(7) \
			BETWEEN; \
			asm volatile("" :: "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5), "v"(g6), "v"(g7)); \
			PBDX_D(0) PBDX_D(1) PBDX_D(2) PBDX_D(3) PBDX_D(4) PBDX_D(5) PBDX_D(6) PBDX_D(7) }
		// first batch: executed by every thread (ids clamped, copies guarded per lane), so that wait() -- which
		// contains a workgroup barrier -- sits at ONE point of the program for all waves
		PBDX_BATCH(wait())
		for (base += 8u * BLOCK; base < n_local; base += 8u * BLOCK)
			PBDX_BATCH((void)0)
#undef PBDX_BATCH
#undef PBDX_G
#undef PBDX_D
		extra();
		lds_dma_wait();
		__syncthreads();
		if (trace && threadIdx.x == 0) trace[1] = wall_clock64();
	}
};
// developer build (-DPBDX_STEP_PROBE=1, scripts/probe_steps.py): cycle stamps inside the colour steps of the traced tile's first thread
#ifndef PBDX_STEP_PROBE
#define PBDX_STEP_PROBE 0
#endif
// workgroups with several tiles walk them in alternating order and keep the tile at the turn in LDS (persistent_kernel); 0 = A/B switch
// timing-only upper bounds (results wrong by construction): the colour barrier / the steady-state record fetch removed
// developer build (-DPBDX_PASS_PROBE=1, scripts/probe_pass.py): wall-clock stamps of the traced tile's first thread along a pass boundary, trace[40 ..]
#ifndef PBDX_PASS_PROBE
#define PBDX_PASS_PROBE 0
#endif
#if PBDX_PASS_PROBE
#define PBDX_PSTAMP(k) do { if (trace && threadIdx.x == 0) trace[40 + (k)] = wall_clock64(); } while (0)
#else
#define PBDX_PSTAMP(k) do { } while (0)
#endif
// DICT: a run of dictionary-form steps (FusedStep::dict, pbdx_plan.h): a slot streams its indices, its multiplier and ONE uint16 -- the offset of its
// parameter record in the tile's table of distinct records, which sits in LDS behind the particles (ltab); the record is read from there when the
// slot is projected.  Same arithmetic on the same values: bit-identical.
struct RecD { uint32_t w[4]; };          // packed indices (2), multiplier, table offset (16-byte units)
// PACKED (round 5): the slot's indices and its ONE streamed dword -- the single parameter plane of a compact two-parameter type (rest length, rest volume)
// or the table offset of a dictionary-form slot -- sit side by side in the index stream as one 8- or 12-byte record (kPackedChunkType, build_idx_image):
// a record fetch is two vector-memory instructions (record, multiplier) instead of three.  Issuing them is what a sub-iteration pays right after the
// colour barrier, when all sixteen waves do it at once (step probes, profiles/HISTORY.md [8]).
template <int TYPE, bool COMPACT, int BLOCK, bool COHERENT, bool QUAD_STEP = false, bool DICT = false, bool PACKED = false>
__device__ __forceinline__ uint32_t run_typed(const RunArgs &a, const TileStreams &str, const ChunkSrc &lchunks, uint32_t c0,
	float4 *lpos, unsigned long long *trace, uint32_t &step_counter, const float4 *ltab = nullptr)
{
	static_assert(!(DICT && (QUAD_STEP || is_quad_type(TYPE))), "dictionary form: one lane per slot");
	static_assert(!PACKED || (!QUAD_STEP && !is_quad_type(TYPE) && (DICT || num_planes(TYPE, COMPACT) == 1)), "packed records: one lane per slot, one streamed dword");
	constexpr int D = Depth<TYPE>::value;
	constexpr bool VEC = vector_params_for_block(BLOCK);
	typedef TileAccess<TYPE, COMPACT, COHERENT, VEC> Acc;
	// quad-lane types (pbdx_quad.h): four lanes share a slot, a chunk holds BLOCK / 4 slots
	constexpr bool QUAD = QUAD_STEP || is_quad_type(TYPE);
	typedef typename std::conditional<DICT, RecD, typename std::conditional<QUAD, RecQ<TYPE, COMPACT>, Rec<TYPE, COMPACT>>::type>::type RecT;
	// per-lane constants of the run
	const uint32_t lane_slot = QUAD ? threadIdx.x >> 2 : threadIdx.x;
	// (parameter block of the lane's 64-slot group: pbdx_plan.h param_float_index -- full segments at 16 bytes per lane, the tail segment after them)
	constexpr uint32_t NP = (uint32_t)num_planes(TYPE, COMPACT);
	const uint32_t v_par = (lane_slot >> 6) * (NP * 256u) + (lane_slot & 63u) * (VEC ? 16u : 4u);
	const uint32_t v_tail = VEC ? (lane_slot >> 6) * (NP * 256u) + (NP / 4u) * 1024u + (lane_slot & 63u) * ((NP % 4u) * 4u) : 0u;
	const QuadLane ql = quad_lane();
	// end of the run (first chunk of another type): precomputed on the host
	const uint32_t run_end = c0 + chunk_run_left(rfl(chunk_words(lchunks, c0).x));

	uint32_t c_ld = c0, c_ex = c0;
	// the ring lives in named records (not an array): keeps every record in registers
	RecT r0, r1, r2, r3;
	// (idx_b, par_b, lam_b: byte offsets of the chunk to fetch in the three streams -- from that chunk's own descriptor while the ring is primed, from
	// the descriptor of the chunk being projected afterwards: FusedChunk::f_*)
	auto fetch = [&](RecT &dst, uint32_t idx_b, uint32_t par_b, uint32_t lam_b)
	{
		// beyond the run the last chunk is fetched again (harmless): the fetch itself stays unconditional.  (Also for waves none of whose lanes
		// has a slot in the chunk -- small scenes: 6 of 8 waves on the 100 k-tet bar.  Letting those skip the fetch was measured SLOWER, 0.638 ->
		// 0.650 ms FEM, 0.745 -> 0.827 XPBD distance + volume, profiles/r03q_*: the compiler can no longer count the loads between a fetch and
		// its use and waits for ALL outstanding loads, i.e. also for the records requested one step ago, and one step is about one memory latency.)
		// (round 4 requested the NEXT fetch's descriptor at the end of this one -- a scalar load whose result was live across the projection: -0.9 ... -2.2 %,
		// and intermittent memory faults in the heavy-type kernels, profiles/HISTORY.md [8], [9]; since round 5 the offsets arrive with the descriptor
		// of the chunk being projected, which is read a whole sub-iteration ahead anyway)
		const Acc acc = { lpos, str, idx_b, par_b, lam_b, v_par, v_tail, a.views[TYPE] };
		if constexpr (PACKED)
		{
			typedef unsigned int v2u __attribute__((ext_vector_type(2)));
			typedef unsigned int v3u __attribute__((ext_vector_type(3)));
			if constexpr (kTwoBodies[TYPE])
			{
				const v2u r = __builtin_amdgcn_raw_buffer_load_b64(str.idx, (int)(lane_slot * 8u), (int)idx_b, 0);
				dst.w[0] = r.x; dst.w[1] = 0u; dst.w[3] = r.y;
			}
			else
			{
				const v3u r = __builtin_amdgcn_raw_buffer_load_b96(str.idx, (int)(lane_slot * 12u), (int)idx_b, 0);
				dst.w[0] = r.x; dst.w[1] = r.y; dst.w[3] = r.z;
			}
			dst.w[2] = 0u;
			if constexpr (kHasLambda[TYPE]) dst.w[2] = __builtin_bit_cast(uint32_t, acc.lam_load(lane_slot));
		}
		else if constexpr (DICT)
		{
			if constexpr (kTwoBodies[TYPE]) { dst.w[0] = acc.idx_raw1(lane_slot); dst.w[1] = 0u; }
			else { const uint2 v = acc.idx_raw2(lane_slot); dst.w[0] = v.x; dst.w[1] = v.y; }
			dst.w[2] = 0u;
			if constexpr (kHasLambda[TYPE]) dst.w[2] = __builtin_bit_cast(uint32_t, acc.lam_load(lane_slot));
			dst.w[3] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(str.par, (int)(lane_slot * 2u), (int)par_b, 0);
		}
		else if constexpr (QUAD) load_rec_quad<TYPE, COMPACT>(acc, ql, lane_slot, dst);
		else load_rec<TYPE, COMPACT>(acc, lane_slot, dst);
		c_ld++;
	};
	// the ring is primed from the first D chunks' own descriptors (once per run)
	auto prime = [&](RecT &dst)
	{
		const ChunkS ch = load_chunk(lchunks, c_ld < run_end ? c_ld : run_end - 1);
		fetch(dst, ch.idx_boff, ch.par_boff, ch.lam_boff);
	};
	prime(r0); prime(r1);
	if constexpr (D == 4) { prime(r2); prime(r3); }
	// Every sub-iteration issues exactly one record fetch, whether or not a projection still happens in it
	// (single loop exit at the bottom): the number of memory operations between a fetch and its use is then
	// the same on every path.
	// the descriptor of the chunk to project next is read BEFORE the colour barrier of the previous one,
	// so that the first thing after a barrier is the LDS gather of the already prefetched record
	ChunkS ch_next = load_chunk(lchunks, c0);
	auto sub = [&](RecT &cur)
	{
#if PBDX_STEP_PROBE
		const bool probing = trace && threadIdx.x == 0 && step_counter < 8u;
		const uint32_t pslot = 20u + 7u * step_counter;
		unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
		if (probing) tA = __builtin_readcyclecounter();
#endif
		// (beyond the run `ch` is the run's last descriptor again and the record fetched with it is never used)
		const ChunkS ch = ch_next;
		if (c_ex < run_end)
		{
			const Acc acc = { lpos, str, ch.idx_boff, ch.par_boff, ch.lam_boff, v_par, v_tail, a.views[TYPE] };
			if (lane_slot < chunk_valid(ch.info))
			{
				if constexpr (DICT)
				{
					// the slot's record: indices and multiplier as streamed, the parameter planes from the tile's table
					Rec<TYPE, COMPACT> full;
					full.w[0] = cur.w[0]; full.w[1] = cur.w[1]; full.w[2] = cur.w[2];
#if PBDX_BOUNDS
					const float4 *e = ltab + (PBDX_BOK(kBndTable, cur.w[3] + dict_entry_f4(NP) - 1u, str.dbg_tab_f4, str.dbg_tile) ? cur.w[3] : 0u);
#else
					const float4 *e = ltab + cur.w[3];
#endif
#pragma unroll
					for (uint32_t q4 = 0; q4 < dict_entry_f4(NP); q4++)
					{
						const float4 v = e[q4];
						const uint32_t vw[4] = { __builtin_bit_cast(uint32_t, v.x), __builtin_bit_cast(uint32_t, v.y), __builtin_bit_cast(uint32_t, v.z), __builtin_bit_cast(uint32_t, v.w) };
#pragma unroll
						for (uint32_t c4 = 0; c4 < 4u; c4++) if (4u * q4 + c4 < NP) full.w[3u + 4u * q4 + c4] = vw[c4];
					}
					exec_rec<TYPE, COMPACT>(acc, full, lane_slot, a.dt, a.first_iter);
				}
				else if constexpr (QUAD) exec_rec_quad<TYPE, COMPACT>(acc, ql, cur, lane_slot, a.dt, a.first_iter);
				else exec_rec<TYPE, COMPACT>(acc, cur, lane_slot, a.dt, a.first_iter);
			}
#if PBDX_STEP_PROBE
			if (probing) tB = __builtin_readcyclecounter();
#endif
			c_ex++;
			ch_next = load_chunk(lchunks, c_ex < run_end ? c_ex : run_end - 1);
#if PBDX_STEP_PROBE
			if (probing) { asm volatile("" :: "s"(ch_next.info)); tC = __builtin_readcyclecounter(); }
#endif
			if (chunk_last_of_step(ch.info))
			{
				if (chunk_barrier(ch.info)) __syncthreads();
#if PBDX_STEP_PROBE
				if (probing) tD = __builtin_readcyclecounter();
#endif
				if (trace && threadIdx.x == 0 && step_counter + 2 < kTraceStride - 1) trace[2 + step_counter] = wall_clock64();
				step_counter++;
			}
		}
		// the record of the chunk D positions ahead, into the ring slot just consumed; its stream offsets came with this chunk's descriptor
		fetch(cur, ch.f_idx_boff, ch.f_par_boff, ch.f_lam_boff);
#if PBDX_STEP_PROBE
		if (probing)
		{
			trace[pslot] = tA; trace[pslot + 1] = 0; trace[pslot + 2] = 0; trace[pslot + 3] = tB;
			trace[pslot + 4] = tC; trace[pslot + 5] = tD; trace[pslot + 6] = __builtin_readcyclecounter();
		}
#endif
	};
	for (;;)
	{
		if constexpr (D == 4) { sub(r0); sub(r1); sub(r2); sub(r3); }
		else { sub(r0); sub(r1); }
		if (c_ex >= run_end) break;
	}
	return run_end;
}

#define PBDX_CASE(T) case T: if constexpr ((MASK >> T) & 1u) { \
		c = ra.views[T].compact ? run_typed<T, true, BLOCK, COHERENT>(ra, str, csrc, c, lpos, trace, step_counter) \
		                        : run_typed<T, false, BLOCK, COHERENT>(ra, str, csrc, c, lpos, trace, step_counter); } \
	else { c = num_chunks; } break;
// a run of dictionary-form steps of type T (chunk type kDictChunkType + T); their records are packed (PBDX_PACK_DICT)
#define PBDX_CASE_DICT(T) case kDictChunkType + T: if constexpr (((MASK >> T) & 1u) && dict_type(T)) { \
		c = ra.views[T].compact ? run_typed<T, true, BLOCK, COHERENT, false, true, PBDX_PACK_DICT != 0>(ra, str, csrc, c, lpos, trace, step_counter, ltab) \
		                        : run_typed<T, false, BLOCK, COHERENT, false, true, PBDX_PACK_DICT != 0>(ra, str, csrc, c, lpos, trace, step_counter, ltab); } \
	else { c = num_chunks; } break;
// a run of packed steps of a compact one-plane type T (chunk type kPackedChunkType + T)
#define PBDX_CASE_PACKED(T) case kPackedChunkType + T: if constexpr (((MASK >> T) & 1u) && packed_plain_type(T)) { \
		c = run_typed<T, true, BLOCK, COHERENT, false, false, true>(ra, str, csrc, c, lpos, trace, step_counter); } \
	else { c = num_chunks; } break;
// a run of StrainTetConstraint steps in quad form (chunk pseudo-type kQuadStrainChunk)
#define PBDX_CASE_QUAD_STRAIN case kQuadStrainChunk: if constexpr (((MASK >> PBDX_STRAIN_TET) & 1u) && PBDX_QUAD_STRAIN) { \
		c = ra.views[PBDX_STRAIN_TET].compact ? run_typed<PBDX_STRAIN_TET, true, BLOCK, COHERENT, true>(ra, str, csrc, c, lpos, trace, step_counter) \
		                                      : run_typed<PBDX_STRAIN_TET, false, BLOCK, COHERENT, true>(ra, str, csrc, c, lpos, trace, step_counter); } \
	else { c = num_chunks; } break;

// LDS: [ chunk descriptors of the tile: kMaxTileChunks x 16 B ][ positions: n_local x float4 ]

__device__ __forceinline__ float4 load_f4_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t index)
{
	typedef float f4 __attribute__((ext_vector_type(4)));
	// sc1: past the CU's L1, which may still hold the line from before this workgroup's own store earlier in the launch
	const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(index * 16u), 0, 16));
	return make_float4(v.x, v.y, v.z, v.w);
}

// pass 0 of a substep: stage x + v' h (TimeIntegration.cpp:7-19, v' = v + g h) for every local particle; for the owned
// ones also last <- old, old <- x (TimeStepController.cpp:112-118).  The state arrays were written by
// earlier launches: plain loads.
template <int BLOCK>
__device__ __forceinline__ void integrate_fill(const FoldArgs &f, const uint4 *src, const uint32_t *gid, const float4 *pos_in, uint4 *lchunks, float4 *lpos,
	uint32_t num_chunks, uint32_t n_local, uint32_t n_owned, unsigned long long *trace, const BndLim &lim)
{
	const uint32_t last_i = n_local - 1u;
	for (uint32_t base = threadIdx.x; base < n_local; base += 4u * BLOCK)
	{
		uint32_t g[4];
		float4 x[4], v[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; k++)
		{
			const uint32_t i = base + k * BLOCK;
			g[k] = gid[PBDX_BCLAMP(kBndGid, (i < last_i ? i : last_i), lim.gid_left, lim.tile)];
			g[k] = PBDX_BCLAMP(kBndParticle, g[k], lim.n_particles, lim.tile);
		}
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) { x[k] = pos_in[g[k]]; v[k] = f.vel[g[k]]; }
#pragma unroll
		for (uint32_t k = 0; k < 4; k++)
		{
			const uint32_t i = base + k * BLOCK;
			if (i >= n_local) continue;
			float4 p = x[k], w = v[k];
			if (w.w != 0.0f)   // mass != 0
			{
				w.x = w.x + f.ghx; w.y = w.y + f.ghy; w.z = w.z + f.ghz;
				p.x = p.x + w.x * f.h; p.y = p.y + w.y * f.h; p.z = p.z + w.z * f.h;
			}
			if (PBDX_BOK(kBndLdsFill, i, lim.lds_f4, lim.tile)) lpos[i] = p;
			if (i < n_owned)
			{
				f.last[g[k]] = f.old[g[k]];
				f.old[g[k]] = x[k];
				// v' is NOT stored: neighbouring tiles integrate this particle as part of their halo from the same v
				// (possibly later than this tile), and the velocity update of the last pass overwrites vel anyway
			}
		}
	}
	__syncthreads();
	if (trace && threadIdx.x == 0) trace[1] = wall_clock64();
}

// last pass of a substep: final positions + TimeIntegration::velocityUpdateFirstOrder / SecondOrder (TimeIntegration.cpp:42-51,
// 69-79) for the owned particles.  old / last / vel were written by this workgroup in pass 0 of the same launch: sc1 loads.
template <int BLOCK>
__device__ __forceinline__ void velocity_write_back(const FoldArgs &f, const uint32_t *gid, float4 *pos_out, const float4 *lpos, uint32_t n_owned, const BndLim &lim)
{
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(f.vel, 0, f.state_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(f.old, 0, f.state_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(f.last, 0, f.state_bytes, 0x00020000);
	// four particles per thread and batch: ids, then all state loads, then the arithmetic -- two exposed memory latencies per batch instead of
	// two per particle (a tile's ~3 900 owned particles are one batch for a 1 024-thread workgroup)
	constexpr uint32_t kBatch = 4;
	const uint32_t last_i = n_owned - 1u;
	for (uint32_t base = threadIdx.x; base < n_owned; base += kBatch * BLOCK)
	{
		uint32_t g[kBatch];
		float4 v[kBatch], o[kBatch], l[kBatch];
#pragma unroll
		for (uint32_t k = 0; k < kBatch; k++)
		{
			const uint32_t i = base + k * BLOCK;
			g[k] = gid[PBDX_BCLAMP(kBndGid, (i < last_i ? i : last_i), lim.gid_left, lim.tile)];
			g[k] = PBDX_BCLAMP(kBndParticle, g[k], lim.n_particles, lim.tile);
		}
#pragma unroll
		for (uint32_t k = 0; k < kBatch; k++)
		{
			v[k] = load_f4_sc1(rv, g[k]);
			o[k] = load_f4_sc1(ro, g[k]);
			if (f.second_order) l[k] = load_f4_sc1(rl, g[k]);
		}
#pragma unroll
		for (uint32_t k = 0; k < kBatch; k++)
		{
			const uint32_t i = base + k * BLOCK;
			if (i >= n_owned) continue;
			const float4 p = lpos[i];
			store_pos<true>(pos_out, g[k], p);
			float4 w = v[k];
			if (w.w == 0.0f) continue;
			if (!f.second_order)
			{
				w.x = f.inv_h * (p.x - o[k].x); w.y = f.inv_h * (p.y - o[k].y); w.z = f.inv_h * (p.z - o[k].z);
			}
			else
			{
				w.x = f.inv_h * (1.5f * p.x - 2.0f * o[k].x + 0.5f * l[k].x);
				w.y = f.inv_h * (1.5f * p.y - 2.0f * o[k].y + 0.5f * l[k].y);
				w.z = f.inv_h * (1.5f * p.z - 2.0f * o[k].z + 0.5f * l[k].z);
			}
			f.vel[g[k]] = w;
		}
	}
}

// Particle ids kept in LDS by the persistent schedule (round 5).  A pass boundary used to pay two dependent memory round trips in front (particle ids ->
// positions: the halo fill) and one behind (ids -> stores: the boundary write-back).  The ids are static, and a tile leaves a quarter of the LDS unused:
//  * halo: right after the fill of pass p every wave requests the id list of the tile's NEXT pass (gid[first .. n_local) of the next segment) with HBM -> LDS
//    copies; they land during the sweep, are complete at the pass's publish (vmcnt(0) + barrier) and the next fill reads its ids with ds_read;
//  * boundary: gid[wb_begin .. n_owned) -- the same in every segment -- is copied once, in pass 0, and every boundary write-back reads it from LDS.
// Only for workgroups with ONE tile (the walk of several tiles changes tile between passes) and where the LDS has the room; the host decides (PersistArgs::ids).
struct LdsIds
{
	uint32_t *halo = nullptr, *bnd = nullptr;      // LDS; halo == null: not in use
	uint32_t halo_cap = 0, bnd_cap = 0;            // entries
};

// One tile of one segment: LDS fill, colour sweep, write-back of the owned particles.
// `keep_owned`: the tile's owned particles are still in LDS from its previous pass (persistent schedule, same
// workgroup, same owned set in every segment): only the halo is staged.  `wait`: see TileFill.
template <uint32_t MASK, int BLOCK, bool COHERENT, class Wait>
__device__ __forceinline__ void process_tile(const SegArgs &sg, const RunArgs &ra, const float4 *pos_in, float4 *pos_out, uint32_t tile_index,
	unsigned long long *trace, uint4 *lchunks, float4 *lpos, bool keep_owned, const Wait &wait, const FoldArgs *fold = nullptr, uint32_t fold_phase = 0,
	bool boundary_only = false, const LdsIds ids = LdsIds(), const SegArgs &sg_next = SegArgs(), bool have_next = false, bool halo_ids_ready = false)
{
	// ids (persistent schedule, one tile per workgroup; ids.halo == null: not in use): see LdsIds.  sg_next / have_next: the segment of this tile's next
	// pass (its halo ids are requested after the fill); halo_ids_ready: the halo ids of THIS pass were requested during the previous one.  (Everything by
	// value or by reference to a kernel argument: a conditional POINTER to one makes the compiler copy the whole argument block into scratch.)
	const bool use_ids = ids.halo != nullptr;
	// boundary_only (persistent schedule, one workgroup per tile, not the last pass of the launch): the owned particles stay in LDS for the next
	// pass, so only the ones another tile stages -- [wb_begin, n_owned): the planner orders the interior first -- have to reach memory
	// fold_phase (persistent schedule only): bit 0 = this pass integrates while it stages, bit 1 = it updates the velocities
#if PBDX_BOUNDS
	tile_index = PBDX_BCLAMP(kBndTile, tile_index, sg.num_tiles, 0u);
#endif
	const FusedTile t = load_tile(sg.tiles, tile_index);
	if (trace && threadIdx.x == 0) trace[0] = wall_clock64();
#if PBDX_PASS_PROBE
	asm volatile("" :: "s"(rfl(t.n_local)));      // the descriptor has arrived
	PBDX_PSTAMP(0);
#endif
	const uint32_t *gid = sg.gid + t.gid_off;
	const uint32_t num_chunks = t.chunk_end - t.chunk_begin;
#if PBDX_BOUNDS
	// the tile descriptor itself: its slices of the gid and chunk streams, its LDS footprint
	(void)PBDX_BOK(kBndGid, t.gid_off + t.n_local - 1u, sg.gid_count, tile_index);
	(void)PBDX_BOK(kBndChunkRange, t.chunk_end, sg.chunk_count + 1u, tile_index);
	(void)PBDX_BOK(kBndChunkRange, num_chunks, kMaxTileChunks + 1u, tile_index);
	(void)PBDX_BOK(kBndLdsFill, t.n_local + t.tab_f4 - 1u, sg.lds_f4, tile_index);
	const BndLim lim = { sg.gid_count - t.gid_off, sg.n_particles, sg.lds_f4, tile_index, ids.halo_cap };
#else
	const BndLim lim = { 0u, 0u, 0u, 0u, 0u };
#endif
	// the next pass's tile descriptor: requested here so that its latency passes during the fill
	FusedTile tn = t;
	if (use_ids && have_next) tn = load_tile(sg_next.tiles, tile_index);
	// stream descriptors, from kernel arguments only (wave-uniform by construction)
	TileStreams str;
	str.idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(sg.idx), 0, sg.idx_bytes, 0x00020000);
	str.par = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sg.params), 0, sg.params_bytes, 0x00020000);
	str.lam = __builtin_amdgcn_make_buffer_rsrc(sg.lambda, 0, sg.lambda_bytes, 0x00020000);
#if PBDX_BOUNDS
	str.dbg_n_local = t.n_local; str.dbg_tab_f4 = t.tab_f4; str.dbg_tile = tile_index;
#endif
	const FusedChunk *gchunks = sg.chunks + t.chunk_begin;
	// the tile's table of distinct parameter records (dictionary-form steps): requested now, written into LDS behind the particles once the fill is through
	constexpr uint32_t kTabPerThread = kDictTableF4 / (uint32_t)BLOCK;
	float4 *ltab = lpos + t.n_local;
	static_assert(kTabPerThread >= 1 && kTabPerThread <= 4, "table staging: up to four 16-byte units per thread");
	float4 tabv0 = make_float4(0.f, 0.f, 0.f, 0.f), tabv1 = tabv0, tabv2 = tabv0, tabv3 = tabv0;      // (named, not an array: registers)
	if (t.tab_f4)
	{
		const float4 *gtab = reinterpret_cast<const float4 *>(sg.params) + t.tab_off;
		const uint32_t lastf4 = t.tab_f4 - 1u, i0 = threadIdx.x;
#if PBDX_BOUNDS
		(void)PBDX_BOK(kBndTableSrc, t.tab_off + lastf4, sg.params_bytes / 16u, tile_index);
		(void)PBDX_BOK(kBndTable, lastf4, kDictTableF4, tile_index);
#endif
		tabv0 = gtab[i0 < lastf4 ? i0 : lastf4];
		if constexpr (kTabPerThread > 1) tabv1 = gtab[i0 + BLOCK < lastf4 ? i0 + BLOCK : lastf4];
		if constexpr (kTabPerThread > 2) { tabv2 = gtab[i0 + 2 * BLOCK < lastf4 ? i0 + 2 * BLOCK : lastf4]; tabv3 = gtab[i0 + 3 * BLOCK < lastf4 ? i0 + 3 * BLOCK : lastf4]; }
	}
	const TileFill<BLOCK, COHERENT> fill = { reinterpret_cast<const uint4 *>(gchunks), gid, pos_in, lchunks, lpos, num_chunks, t.n_local,
		keep_owned ? (t.n_owned & ~63u) : 0u, trace, lim, (use_ids && keep_owned && halo_ids_ready) ? ids.halo : nullptr };
	bool staged = false;
	if constexpr (COHERENT)
		if (fold_phase & 1u)
		{
			integrate_fill<BLOCK>(*fold, reinterpret_cast<const uint4 *>(gchunks), gid, pos_in, lchunks, lpos, num_chunks, t.n_local, t.n_owned, trace, lim);
			staged = true;
		}
	// the next pass's halo ids and the dictionary table: inside the fill (TileFill `extra`) where the fill's first batch covers the whole halo -- every id of
	// THIS pass has then been read from ids.halo before the barrier of `wait`, so the list of the next pass may be requested into the same region right behind
	// the position copies -- and otherwise (pass 0, which integrates while it stages; halos of more than eight ids per thread) behind it as before
	const bool single_batch = fill.first + 8u * (uint32_t)BLOCK >= t.n_local;
	const bool ids_in_fill = !staged && single_batch;
	auto request_next_ids = [&]()
	{
		const uint32_t nfirst = tn.n_owned & ~63u, count = tn.n_local - nfirst;      // gid[nfirst .. n_local) of the next pass; 16-byte aligned: gid_off and nfirst are multiples of 4
		const float4 *src = reinterpret_cast<const float4 *>(sg_next.gid + tn.gid_off + nfirst);
		for (uint32_t j = threadIdx.x; 4u * j < count; j += BLOCK)
			if (PBDX_BOK(kBndLdsIds, 4u * j + 3u, ids.halo_cap, tile_index) && PBDX_BOK(kBndGid, tn.gid_off + nfirst + 4u * j + 3u, sg_next.gid_count_dbg(), tile_index))
				lds_dma16<false>(src, j, reinterpret_cast<float4 *>(ids.halo) + (j & ~63u));
	};
	auto store_table = [&]()
	{
		const uint32_t i0 = threadIdx.x;
		if (i0 < t.tab_f4) ltab[i0] = tabv0;
		if constexpr (kTabPerThread > 1) { if (i0 + BLOCK < t.tab_f4) ltab[i0 + BLOCK] = tabv1; }
		if constexpr (kTabPerThread > 2) { if (i0 + 2 * BLOCK < t.tab_f4) ltab[i0 + 2 * BLOCK] = tabv2; if (i0 + 3 * BLOCK < t.tab_f4) ltab[i0 + 3 * BLOCK] = tabv3; }
	};
	const bool table_in_fill = !staged;
	if (!staged)
		fill(wait, [&]() {
			if (use_ids && have_next && ids_in_fill) request_next_ids();
			if (t.tab_f4 && table_in_fill) store_table();
		});
	if (use_ids)
	{
		// (the fill ended with vmcnt(0) + barrier: every id it read from ids.halo has been consumed)
		if (have_next && !ids_in_fill) request_next_ids();
		if (fold_phase & 1u)
		{
			// pass 0 of a launch: the boundary ids, once (read by the write-backs after the wait below)
			const uint32_t b0 = t.wb_begin & ~3u, count = t.n_owned - b0;
			const float4 *src = reinterpret_cast<const float4 *>(gid + b0);
			for (uint32_t j = threadIdx.x; 4u * j < count; j += BLOCK)
				if (PBDX_BOK(kBndLdsIds, 4u * j + 3u, ids.bnd_cap, tile_index))
					lds_dma16<false>(src, j, reinterpret_cast<float4 *>(ids.bnd) + (j & ~63u));
		}
	}
	if (t.tab_f4 && !table_in_fill)
	{
		store_table();
		__syncthreads();
	}
	PBDX_PSTAMP(3);      // table staged
	// chunks are addressed relative to the tile from here on (they sit at lchunks[0 .. num_chunks))
#if PBDX_BOUNDS
	const ChunkSrc csrc = { lchunks, reinterpret_cast<const uint4 *>(gchunks), num_chunks };
#else
	const ChunkSrc csrc = { lchunks, reinterpret_cast<const uint4 *>(gchunks) };
#endif
	uint32_t c = 0, step_counter = 0;
	while (c < num_chunks)
	{
		switch (chunk_type(rfl(chunk_words(csrc, c).x)))
		{
			PBDX_CASE(PBDX_DISTANCE) PBDX_CASE(PBDX_DISTANCE_XPBD) PBDX_CASE(PBDX_DIHEDRAL)
			PBDX_CASE(PBDX_ISOMETRIC_BENDING) PBDX_CASE(PBDX_ISOMETRIC_BENDING_XPBD)
			PBDX_CASE(PBDX_FEM_TRIANGLE) PBDX_CASE(PBDX_STRAIN_TRIANGLE)
			PBDX_CASE(PBDX_VOLUME) PBDX_CASE(PBDX_VOLUME_XPBD)
			PBDX_CASE(PBDX_FEM_TET) PBDX_CASE(PBDX_FEM_TET_XPBD) PBDX_CASE(PBDX_STRAIN_TET)
			PBDX_CASE(PBDX_SHAPE_MATCHING)
			PBDX_CASE_QUAD_STRAIN
			PBDX_CASE_DICT(PBDX_ISOMETRIC_BENDING) PBDX_CASE_DICT(PBDX_ISOMETRIC_BENDING_XPBD) PBDX_CASE_DICT(PBDX_FEM_TET) PBDX_CASE_DICT(PBDX_FEM_TET_XPBD)
			PBDX_CASE_PACKED(PBDX_DISTANCE) PBDX_CASE_PACKED(PBDX_DISTANCE_XPBD) PBDX_CASE_PACKED(PBDX_VOLUME) PBDX_CASE_PACKED(PBDX_VOLUME_XPBD)
		default: c = num_chunks; break;
		}
	}
	PBDX_PSTAMP(4);      // sweep done
	bool written = false;
	if constexpr (COHERENT)
		if (fold_phase & 2u)
		{
			velocity_write_back<BLOCK>(*fold, gid, pos_out, lpos, t.n_owned, lim);
			written = true;
		}
	// write-back of the boundary particles with their ids from LDS (LdsIds): no memory round trip in front of the stores
	if (!written && use_ids && boundary_only)
	{
		if (fold_phase & 1u) { lds_dma_wait(); __syncthreads(); }      // (pass 0: the copy of the boundary ids was requested above)
		const uint32_t b0 = t.wb_begin & ~3u;
		for (uint32_t i = t.wb_begin + threadIdx.x; i < t.n_owned; i += BLOCK)
		{
			const uint32_t g = ids.bnd[PBDX_BCLAMP(kBndLdsIds, i - b0, ids.bnd_cap, tile_index)];
			if (PBDX_BOK(kBndParticle, g, lim.n_particles, lim.tile)) store_pos<COHERENT>(pos_out, g, lpos[i]);
		}
		written = true;
	}
	// write-back of the owned particles, ids batched like the fill
	if (!written)
	{
		constexpr uint32_t kWbBatch = 4;
		const uint32_t last = t.n_owned - 1u;
		for (uint32_t base = (boundary_only ? t.wb_begin : 0u) + threadIdx.x; base < t.n_owned; base += kWbBatch * BLOCK)
		{
			uint32_t g[kWbBatch];
#pragma unroll
			for (uint32_t k = 0; k < kWbBatch; k++) { const uint32_t i = base + k * BLOCK; g[k] = gid[PBDX_BCLAMP(kBndGid, (i < last ? i : last), lim.gid_left, lim.tile)]; }
#pragma unroll
			for (uint32_t k = 0; k < kWbBatch; k++)
			{
				const uint32_t i = base + k * BLOCK;
				if (PBDX_BOK(kBndParticle, g[k], lim.n_particles, lim.tile)) store_pos<COHERENT>(pos_out, g[k], lpos[i < last ? i : last]);
			}
		}
	}
	PBDX_PSTAMP(5);      // write-back stores issued
	if (trace && threadIdx.x == 0)
	{
		__builtin_amdgcn_s_waitcnt(0);
		trace[kTraceStride - 1] = wall_clock64();
	}
}

template <uint32_t MASK, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fused_kernel(FusedArgs a)
{
	extern __shared__ uint4 lds_raw[];
	uint4 *lchunks = lds_raw;
	float4 *lpos = reinterpret_cast<float4 *>(lds_raw + kMaxTileChunks);
	// the chunk descriptors are read through the scalar cache: lines of an earlier plan that lived at the same addresses must not be served
	__builtin_amdgcn_s_dcache_inv();
	const uint32_t tile_index = logical_block(a.seg.num_tiles, a.xcd_remap);
	unsigned long long *trace = a.trace ? a.trace + (size_t)tile_index * kTraceStride : nullptr;
	const RunArgs ra = { a.dt, a.first_iter, a.views };
	process_tile<MASK, BLOCK, false>(a.seg, ra, a.pos_in, a.pos_out, tile_index, trace, lchunks, lpos, false, [] {});
}

// ---- (A') persistent schedule: all launches of a substep's sweeps as ONE launch ---------------------------
// The per-segment launches of (A) are separated by kernel boundaries: ~3.2 us of dead time each plus the wait for
// the slowest of 256 tiles (step traces: 5-10 % of a launch).  Here one workgroup per tile stays resident for all
// `passes` = iterations x segments and a tile starts pass p as soon as the tiles it exchanges particles with have
// finished pass p-1: per tile a completed-pass counter (`epoch`), published after the tile's positions are written
// through (sc1 stores -> s_waitcnt vmcnt(0) -> barrier -> relaxed agent-scope store), polled by one wave of the
// reader (relaxed agent-scope loads), whose fill then reads the positions with sc1 loads.  The dependency list
// of (segment, tile) holds the owners of its halo (read-after-write) and the tiles that had its particles in their
// halo one pass earlier (write-after-read on the double-buffered positions); ensure_plan() derives it from the
// plan.  Arithmetic, order and streams are those of (A): results are bit-identical.
// Residency: gridDim <= number of CUs and one workgroup per CU; HIP guarantees neither, so every wait is bounded
// (PersistArgs::spin_limit) and a timeout raises `*error` instead of hanging -- the host then restores the state it
// saved at the start of the call and repeats the call with schedule (A).

template <uint32_t MASK, int BLOCK>
__global__ __launch_bounds__(BLOCK) void persistent_kernel(PersistArgs a)
{
	extern __shared__ uint4 lds_raw[];
	__shared__ uint32_t s_failed, s_go;
	uint4 *lchunks = lds_raw;
	float4 *lpos = reinterpret_cast<float4 *>(lds_raw + kMaxTileChunks);
	uint32_t sgi = 0;
	__builtin_amdgcn_s_dcache_inv();      // (see fused_kernel)
	// Residency handshake, before anything is modified: every workgroup announces itself; the last one to arrive
	// decides GO, a workgroup that has waited kArriveLimitTicks decides ABORT (one compare-and-swap settles it for
	// everybody).  After GO all gridDim.x workgroups are running and stay until the end, so no later wait can
	// starve.  After ABORT nobody touches the state, the rest of the step's kernels turn into no-ops (ctl) and the
	// host completes the step with one launch per segment.
	if (threadIdx.x == 0)
	{
		s_failed = 0u;
		uint32_t go = 0u;
		if (__hip_atomic_load(a.ctl + kCtlAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
		{
			uint32_t *arrive = a.epoch + a.num_tiles, *decision = arrive + 1;
			if (atomicAdd(arrive, 1u) + 1u == a.expect) atomicCAS(decision, 0u, 1u);
			const unsigned long long t0 = wall_clock64();
			uint32_t d;
			while ((d = __hip_atomic_load(decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u)
			{
				if (wall_clock64() - t0 > kArriveLimitTicks) atomicCAS(decision, 0u, 2u);
				__builtin_amdgcn_s_sleep(1);
			}
			go = d == 1u ? 1u : 0u;
			if (!go && atomicCAS(a.ctl + kCtlAbort, 0u, 1u) == 0u)
			{
				const uint32_t k = __hip_atomic_load(a.ctl + kCtlSubstep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				a.ctl[kCtlFailedSubstep] = k;
				a.error[2] = k;
				__hip_atomic_store(a.error + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
		s_go = go;
	}
	__syncthreads();
	if (!s_go) return;
	for (uint32_t pass = 0; pass < a.passes; pass++)
	{
		// BY VALUE: one wide scalar load of the segment's arguments per pass.  As a reference into the kernel-argument segment every field was fetched where it
		// was first used, each behind a wait of its own -- five dependent scalar-cache round trips between the publish of a pass and the poll of the next
		// (pass probes: 0.56 + 0.76 us per pass; profiles/HISTORY.md [10]).  The values live in SGPRs (spilled to lanes of a vector register under pressure:
		// a v_readlane, not a memory access).
		const SegArgs sg = a.seg[sgi];
		const RunArgs ra = { a.dt, pass < a.first_iter_passes ? 1 : 0, a.views };
		const float4 *pos_in = a.pos[(a.start + pass) & 1u];
		float4 *pos_out = a.pos[(a.start + pass + 1u) & 1u];
		// This workgroup's tiles: blockIdx.x + j gridDim.x, j = 0 .. m - 1, walked forwards in even passes and backwards in odd ones, so that the
		// LAST tile of a pass is the FIRST of the next.  That tile's owned particles are still in LDS when its next pass starts: it stages only
		// its halo then, and it wrote back only its boundary particles; the other tiles of the workgroup are staged and written back in full.
		// One tile per workgroup (m = 1) is the case where every pass is both: the owned particles never leave LDS.  (Asynchronous-execution
		// model of exactly this walk: check_persistent_deps(..., keep_owned, workgroups), pbdx_plan.cpp.)
		const uint32_t m = (a.num_tiles - blockIdx.x + gridDim.x - 1u) / gridDim.x;
		for (uint32_t k = 0; k < m; k++)
		{
			const uint32_t tile = PBDX_BCLAMP(kBndTile, blockIdx.x + ((pass & 1u) ? m - 1u - k : k) * gridDim.x, a.num_tiles, pass);
			const bool first_of_pass = k == 0u, last_of_pass = k + 1u == m;
			// the wait for the neighbouring tiles, run by the fill once its particle ids are in flight: one wave polls
			// the tile's dependencies, one lane each (lists are short: the adjacent tiles)
			unsigned long long *trace = (a.trace[sgi] && pass + a.num_segs >= a.passes) ? a.trace[sgi] + (size_t)tile * kTraceStride : nullptr;
			// the bounds of the tile's dependency list: scalar loads, requested here with the pass's other scalars (not behind the fill's first ids)
			const uint32_t d0 = sload_u32(a.dep_off[sgi], tile), d1 = sload_u32(a.dep_off[sgi], tile + 1u);
			auto wait = [&]()
			{
				if (!pass) return;
				PBDX_PSTAMP(1);      // the fill's first ids are in flight: the wait for the neighbours starts
				if (threadIdx.x < 64)
				{
					const unsigned long long t0 = wall_clock64();
					for (uint32_t d = d0 + threadIdx.x; d < d1; d += 64)
					{
#if PBDX_BOUNDS
						if (!PBDX_BOK(kBndDep, d, a.dep_count[sgi], tile) || !PBDX_BOK(kBndDep, a.dep_tile[sgi][d], a.num_tiles, tile)) continue;
#endif
						const uint32_t *flag = a.epoch + a.dep_tile[sgi][d];
						while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pass)
						{
							if (wall_clock64() - t0 > a.spin_limit) { s_failed = 1u; break; }
							__builtin_amdgcn_s_sleep(1);
						}
					}
				}
				PBDX_PSTAMP(2);      // this wave's dependencies have published
				__syncthreads();
			};
			const uint32_t fold_phase = a.folded ? ((pass == 0 ? 1u : 0u) | (pass + 1u == a.passes ? 2u : 0u)) : 0u;
			// particle ids in LDS (LdsIds): one tile per workgroup, folded launch (pass 0 stages everything and copies the boundary ids)
			const bool use_ids = a.ids_halo_cap != 0u && m == 1u && a.folded;
			LdsIds ids;
			if (use_ids) { ids.halo = reinterpret_cast<uint32_t *>(lds_raw + a.ids_halo_off16); ids.bnd = reinterpret_cast<uint32_t *>(lds_raw + a.ids_bnd_off16); ids.halo_cap = a.ids_halo_cap; ids.bnd_cap = a.ids_bnd_cap; }
			const uint32_t sgi_next = sgi + 1u == a.num_segs ? 0u : sgi + 1u;
			process_tile<MASK, BLOCK, true>(sg, ra, pos_in, pos_out, tile, trace, lchunks, lpos, pass != 0 && first_of_pass, wait, &a.fold, fold_phase,
				last_of_pass && pass + 1u != a.passes, ids, a.seg[sgi_next], use_ids && pass + 1u != a.passes, use_ids && pass != 0u);
			if (s_failed)
			{
				// a neighbour never arrived: the state of this step is garbage.  Say so, turn every later kernel of the call
				// into a no-op (ctl) and leave (uniform: one LDS word); the host restores the snapshot it took at the start
				// of the call and repeats the call with one launch per segment.
				if (threadIdx.x == 0) { atomicOr(a.error, 1u); atomicExch(a.ctl + kCtlAbort, 1u); }
				return;
			}
			// publish: this thread's stores have left the CU, then everybody's, then the counter
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			PBDX_PSTAMP(6);      // this thread's stores have left the CU
			__syncthreads();
			if (threadIdx.x == 0 && !(a.mute_tile0 && tile == 0u)) __hip_atomic_store(a.epoch + tile, pass + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			PBDX_PSTAMP(7);      // published
		}
		sgi = sgi + 1u == a.num_segs ? 0u : sgi + 1u;
	}
	// substeps completed in this call (velocity_kernel counts them when the launch is not folded)
	if (a.folded && blockIdx.x == 0 && threadIdx.x == 0) a.ctl[kCtlSubstep] = a.ctl[kCtlSubstep] + 1u;
}


} // namespace

project_fn project_kernel_for(int type, bool compact) { return kProjectKernels[type][compact ? 1 : 0]; }

fused_fn pick_fused_kernel(uint32_t mask, int block)
{
	if ((mask & ~kMaskClothXpbd) == 0)
		return block == 1024 ? fused_kernel<kMaskClothXpbd, 1024> : block == 512 ? fused_kernel<kMaskClothXpbd, 512> : fused_kernel<kMaskClothXpbd, 256>;
	if ((mask & ~kMaskLight) == 0)
		return block == 1024 ? fused_kernel<kMaskLight, 1024> : block == 512 ? fused_kernel<kMaskLight, 512> : fused_kernel<kMaskLight, 256>;
	// heavy types (FEM / strain / shape matching) need > 128 VGPRs: at most 512 threads per workgroup
	if ((mask & ~kMaskFemTet) == 0)
		return block >= 512 ? fused_kernel<kMaskFemTet, 512> : fused_kernel<kMaskFemTet, 256>;
	if ((mask & ~kMaskStrainTet) == 0)
		return block >= 512 ? fused_kernel<kMaskStrainTet, 512> : fused_kernel<kMaskStrainTet, 256>;
	return block >= 512 ? fused_kernel<kMaskAll, 512> : fused_kernel<kMaskAll, 256>;
}

persist_fn pick_persistent_kernel(uint32_t mask, int block)
{
	if ((mask & ~kMaskClothXpbd) == 0)
		return block == 1024 ? persistent_kernel<kMaskClothXpbd, 1024> : block == 512 ? persistent_kernel<kMaskClothXpbd, 512> : persistent_kernel<kMaskClothXpbd, 256>;
	if ((mask & ~kMaskLight) == 0)
		return block == 1024 ? persistent_kernel<kMaskLight, 1024> : block == 512 ? persistent_kernel<kMaskLight, 512> : persistent_kernel<kMaskLight, 256>;
	if ((mask & ~kMaskFemTet) == 0)
		return block >= 512 ? persistent_kernel<kMaskFemTet, 512> : persistent_kernel<kMaskFemTet, 256>;
	if ((mask & ~kMaskStrainTet) == 0)
		return block >= 512 ? persistent_kernel<kMaskStrainTet, 512> : persistent_kernel<kMaskStrainTet, 256>;
	return block >= 512 ? persistent_kernel<kMaskAll, 512> : persistent_kernel<kMaskAll, 256>;
}

int sweep_bounds_report(uint32_t out[8], int reset)
{
#if PBDX_BOUNDS
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds_rec), 8 * sizeof(uint32_t)) != hipSuccess) return PBDX_ERR_HIP;
	out[7] = 1u;
	if (reset)
	{
		const uint32_t zero[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
		if (hipMemcpyToSymbol(HIP_SYMBOL(g_bounds_rec), zero, sizeof(zero)) != hipSuccess) return PBDX_ERR_HIP;
	}
#else
	(void)out; (void)reset;
#endif
	return PBDX_OK;
}

} // namespace pbdx
