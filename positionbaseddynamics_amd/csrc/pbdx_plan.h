// pbdx_plan.h -- host-side planner of the colour-fused tile schedule.
//
// The reference sweeps the colour groups of SimulationModel::getConstraintGroups()
// one after the other (TimeStepController.cpp:270-286); within a colour every constraint is
// independent.  A per-colour launch therefore moves ~30 MB and is latency bound.  The planner
// turns a run of consecutive colours [c0,c1) (a "segment") into ONE launch:
//
//   * particles are partitioned into spatially compact tiles (recursive coordinate bisection);
//     every particle is owned by exactly one tile;
//   * for a tile the planner walks the colours of the segment BACKWARDS and collects the
//     dependency closure: the constraints whose result can reach an owned particle by the end of
//     the segment, and all particles they touch.  A workgroup loads that particle set into LDS,
//     executes the collected constraints colour by colour (workgroup barrier between colours)
//     and writes back only its owned particles.  Constraints in the halo are executed
//     redundantly by neighbouring tiles on identical inputs in the identical order, so every
//     tile reproduces exactly the values the colour-sequential sweep produces: the result is
//     bit-identical to the per-colour schedule (and hence to the reference).
//   * positions are double buffered (a segment reads buffer A and writes buffer B) because a
//     neighbour still needs the pre-segment value of a halo particle.
//
// The planner is plain C++ (no HIP) so that it can be tested without a GPU
// (pbdx_plan_selftest, tests/test_plan.py).
#ifndef PBDX_PLAN_H
#define PBDX_PLAN_H

#include <stdint.h>
#include <string>
#include <vector>

#define PBDX_MAX_PARAMS 24

namespace pbdx {

struct PlanBatch
{
	int type;
	uint32_t colour;        // ordinal of the colour group (0,1,2,... in execution order)
	uint32_t count;
	const uint32_t *idx;    // count * num_bodies(type)
	const float *params;    // count * param_stride(type)
};

// Parameter layout of a constraint type.  Two layouts, both fixed at compile time so that every
// load in the kernels is unconditional (no data-dependent branch between a prefetch and its use):
//   compact  the parameters that are normally shared by all constraints of a scene (stiffness,
//            material constants, flags) are scalars (`u`), and of the isometric-bending matrix Q
//            only the upper triangle is streamed (init_IsometricBendingConstraint builds Q
//            bitwise symmetric, PositionBasedDynamics.cpp:169-180).  Chosen when the schedule's
//            records really are like that (bitwise checks, compute_type_view).
//   full     every parameter of the record is streamed.
struct TypeView
{
	uint32_t compact;
	uint32_t pad[3];
	float u[PBDX_MAX_PARAMS];
};

// bit k set: parameter k of the host record (include/pbdx.h) is a scalar in the compact layout
constexpr uint32_t kCompactScalars[13] = {
	1u << 1,                                                    // DISTANCE            stiffness
	1u << 1,                                                    // DISTANCE_XPBD
	1u << 1,                                                    // DIHEDRAL
	1u << 0,                                                    // ISOMETRIC_BENDING   stiffness (+ symmetric Q)
	1u << 0,                                                    // ISOMETRIC_BENDING_XPBD
	(1u << 5) | (1u << 6) | (1u << 7) | (1u << 8) | (1u << 9),  // FEM_TRIANGLE        xx, yy, xy, poisson ratios
	(1u << 4) | (1u << 5) | (1u << 6) | (1u << 7) | (1u << 8),  // STRAIN_TRIANGLE     stiffnesses, normalise flags
	1u << 1,                                                    // VOLUME
	1u << 1,                                                    // VOLUME_XPBD
	(1u << 10) | (1u << 11),                                    // FEM_TET             stiffness, poisson ratio
	(1u << 10) | (1u << 11),                                    // FEM_TET_XPBD
	(1u << 9) | (1u << 10) | (1u << 11) | (1u << 12),           // STRAIN_TET
	1u << 0,                                                    // SHAPE_MATCHING      stiffness
};
constexpr int kParamCount[13] = { 2, 2, 2, 17, 17, 10, 9, 2, 2, 12, 12, 13, 24 };

// Projections spread over the four lanes of a quad in the fused tile kernels (pbdx_quad.h); a workgroup-wide chunk in that form holds
// BLOCK / 4 slots.
//  * StrainTetConstraint, PER STEP (PBDX_QUAD_STRAIN, default ON): six sequential sub-projections whose work is per PARTICLE (gradient,
//    correction), so a lane per particle carries little redundant work.  Measured (profiles/r03m_*, r03n_*): a colour step of a tile in quad form
//    takes 1.7 us per chunk of BLOCK / 4 slots against 2.4-2.7 us for the one-lane-per-constraint chain whatever its slot count (<= BLOCK): a win
//    exactly for the steps that fit ONE quad chunk, a loss for larger ones (quad form for every step: 1.55 vs 1.16 ms on the bar).  So the
//    engine picks per step when it expands a plan into chunks: steps with 4 * slots <= BLOCK become one chunk of pseudo-type kQuadStrainChunk,
//    the others stay in the one-lane form; both forms are instantiated in the kernel.
//  * FEMTetConstraint / XPBD_FEMTetConstraint (PBDX_QUAD_FEM, default OFF: an opt-in BUILD, _lib/libpbdx_quad.so, built and GPU-tested next to the
//    product; every step in quad form): bit-identical, but measured SLOWER on MI355X -- the 100 k-tet bar 0.878 ms vs 0.638 ms per substep, 32 bars
//    1.76 vs 1.01 (profiles/r03e_quad_lanes_ab.log).  A lane of the quad still executes ~450 instructions (the per-slot preamble -- volumes, two
//    divisions -- the lane selects and the DPP exchanges eat what the column split saves of the ~390-instruction scalar chain), so even a
//    single quad chunk (1.6 us) loses to the scalar step (1.0 us).
#ifndef PBDX_QUAD_LANES
#define PBDX_QUAD_LANES 0            // 1: the opt-in build (FEM tets in quad form)
#endif
#ifndef PBDX_QUAD_FEM
#define PBDX_QUAD_FEM PBDX_QUAD_LANES
#endif
#ifndef PBDX_QUAD_STRAIN
#define PBDX_QUAD_STRAIN 1
#endif
constexpr bool is_quad_type(int type) { return PBDX_QUAD_FEM != 0 && (type == 9 || type == 10); }      // every step of FEM_TET / FEM_TET_XPBD in quad form
constexpr uint32_t kQuadStrainChunk = 13;    // chunk pseudo-type: a StrainTetConstraint step in quad form (chunk type field: 6 bits, real types 0..12)
constexpr bool quad_strain_step(int type, uint32_t slots, uint32_t block) { return PBDX_QUAD_STRAIN != 0 && type == 11 && slots * 4u <= block; }

constexpr bool is_bending_type(int type) { return type == 3 || type == 4; }
// Q(r,c) is parameter 1 + c*4 + r; the strictly lower triangle (r > c) is mirrored, not streamed
constexpr bool is_mirrored_q(int type, bool compact, int k)
{
	return compact && is_bending_type(type) && k >= 1 && ((k - 1) % 4) > ((k - 1) / 4);
}
constexpr bool is_scalar_param(int type, bool compact, int k) { return compact && ((kCompactScalars[type] >> k) & 1u); }
constexpr bool param_streams(int type, bool compact, int k) { return !is_scalar_param(type, compact, k) && !is_mirrored_q(type, compact, k); }
constexpr int param_plane(int type, bool compact, int k)
{
	int n = 0;
	for (int j = 0; j < k; j++) n += param_streams(type, compact, j) ? 1 : 0;
	return n;
}
constexpr int num_planes(int type, bool compact) { return param_plane(type, compact, kParamCount[type]); }

// Layout of the streamed parameters of one step inside a segment's parameter stream.  The step's slots are taken in groups of 64 (one wave);
// a group's block holds its `np` planes (streamed parameters) x 64 slots in one of two forms (FusedSegment::vector_params):
//  * PLANES: plane p stored [64 slots], one coalesced dword load per plane;
//  * VECTOR SEGMENTS: segment s = planes 4s .. 4s+3 stored [64 slots][4 floats], the last segment = the remaining np % 4 planes stored
//    [64 slots][np % 4 floats]: a lane fetches four consecutive planes of its slot with ONE 16-byte load (a wave: 1 KiB contiguous).
// Issuing a vector-memory instruction costs a wave 15-40 cycles whatever its width (scripts/microbench/vmem_issue.hip: 11 dword loads 290-585
// counts, 3 dwordx4 loads 190-255), and a latency-bound colour step is ONE wave's instruction stream: vector segments make the 100 k-tet
// bar 6.7 % faster (FEM 0.638 -> 0.595 ms), 16 bars 7.4 %, 100x100 / 300x300 cloth 3.5 %.  The 1 024-thread kernels of large cloth
// scenes (16 waves per CU, throughput-bound) measure 2.5 % SLOWER with them (1 M cloth 0.769 -> 0.79 ms, 64-instance block 2.06 -> 2.12;
// profiles/r03r_*) although their code gets smaller (114 instead of 128 VGPRs, no spill): those keep the planes.  The rule, in one place:
constexpr bool vector_params_for_block(int block) { return block <= 512; }
// Returns the float index of (plane, slot) relative to the step's first float.
constexpr size_t param_float_index(bool vec, uint32_t np, uint32_t plane, uint32_t slot)
{
	const uint32_t nfull = np / 4u, tail = np % 4u, l = slot % 64u;
	return (size_t)(slot / 64u) * (np * 64u) + (!vec ? plane * 64u + l : plane < 4u * nfull ? (plane / 4u) * 256u + l * 4u + plane % 4u : nfull * 256u + l * tail + (plane - 4u * nfull));
}

// plane index tables (constant-folded in the kernels once the parameter index is a constant)
struct PlaneTable
{
	signed char plane[2][13][PBDX_MAX_PARAMS];
	constexpr PlaneTable() : plane()
	{
		for (int c = 0; c < 2; c++)
			for (int t = 0; t < 13; t++)
				for (int k = 0; k < PBDX_MAX_PARAMS; k++)
					plane[c][t][k] = (k < kParamCount[t] && param_streams(t, c != 0, k)) ? (signed char)param_plane(t, c != 0, k) : (signed char)-1;
	}
};

// One view for a set of parameter-record arrays of the same type (bitwise comparisons).
struct ParamSpan { const float *params; uint32_t count; };
void compute_type_view(int type, const std::vector<ParamSpan> &spans, TypeView &out);

struct FusedStep      // one (colour, type) run of a tile; 32 bytes, read with scalar loads
{
	uint32_t type;
	uint32_t count;
	uint32_t idx_off;     // into the segment's uint16 index stream (2 or 4 entries per slot)
	uint32_t par_off;     // into the segment's float parameter stream; "wave-tiled" layout: slot q, plane p at
	                      //   par_off + (q / 64) * (nplanes * 64) + p * 64 + (q % 64)
	                      // i.e. one wave reads 256 contiguous bytes per plane and the plane offset is an immediate
	uint32_t dict;        // 1: DICTIONARY form -- the step's part of the parameter stream holds one uint16 per slot, the offset (in 16-byte units) of the slot's
	                      // record in the tile's table of DISTINCT records (FusedTile::tab_off), which the tile stages in LDS with its particles.  A regular
	                      // mesh repeats a few hundred distinct records (the 1000x1000 cloth: 6 M constraints, < 1 000 distinct bending matrices, 100-250 per
	                      // tile), so a bending slot streams 2 bytes instead of 40 (dict_type() says which types take the form)
	uint32_t lam_off;     // into the segment's lambda stream (XPBD types)
	uint32_t barrier;     // workgroup barrier after this step (last step of a colour)
	uint32_t cid_off;     // host only: into slot_cid
};

struct FusedTile      // 44 bytes
{
	uint32_t step_begin, step_end;
	uint32_t n_local;     // particles staged in LDS (owned first; among the owned ones the INTERIOR first: see wb_begin)
	uint32_t n_owned;
	uint32_t gid_off;     // into the segment's global-id stream
	uint32_t slots;       // constraints executed by this tile in this segment
	uint32_t chunk_begin, chunk_end;   // filled by the engine once the workgroup size is chosen (FusedChunk list)
	uint32_t tab_off, tab_f4;          // the tile's table of distinct parameter records (dictionary form): offset into the segment's parameter stream and
	                      // size, both in 16-byte units; records in plane order, padded to whole 16-byte units; 0 / 0 if the tile has none
	uint32_t wb_begin;    // owned particles [0, wb_begin) are INTERIOR: no other tile stages them in any segment, and this tile's own fill does not
	                      // re-read them while its owned particles stay in LDS (persistent schedule) -- a pass that is not the last one of its launch
	                      // writes back only [wb_begin, n_owned).  The same value in every segment (the owned order is shared by all segments).
};

// One workgroup-wide pass over (part of) a step: lanes [0, valid) project slot k*BLOCK + lane of the step.
// 16 bytes, staged in LDS, read as scalars.
struct FusedChunk
{
	uint32_t info;        // type (bits 0-5) | barrier after this chunk (bit 6) | last chunk of its step (bit 7) |
	                      // valid lanes (bits 8-18) | chunks left in this run of equal type, this one included (bits 19-31)
	uint32_t idx_boff;    // BYTE offsets of the chunk's first slot in the three streams
	uint32_t par_boff;
	uint32_t lam_boff;
	// ... and of the chunk whose record the sweep FETCHES while it projects this one: the chunk `ring depth` positions ahead in the run (the run's last
	// chunk beyond its end).  With them in the descriptor the sweep reads ONE descriptor per sub-iteration, a whole sub-iteration before it is used,
	// and the record fetch waits for no scalar load (round 5; until then the fetch read a descriptor of its own and waited for it right after the
	// colour barrier).  The ring depth per type is the kernels' (pbdx_sweep.h Depth<TYPE>); the engine fills these when it expands the steps.
	uint32_t f_idx_boff, f_par_boff, f_lam_boff, pad;
};
static_assert(sizeof(FusedChunk) == 32, "one s_load_dwordx8 per chunk descriptor");

struct FusedSegment
{
	uint32_t colour_begin = 0, colour_end = 0;
	std::vector<FusedTile> tiles;
	std::vector<FusedStep> steps;
	std::vector<uint16_t> idx;
	std::vector<float> params;
	std::vector<uint32_t> gid;
	std::vector<uint32_t> slot_cid;     // host only: slot -> constraint id in execution order
	uint32_t lam_count = 0;
	uint32_t max_local = 0;
	uint32_t type_mask = 0;
	uint64_t slots = 0;                 // executed (incl. redundant halo copies)
	uint64_t constraints = 0;           // distinct constraints of the segment
	uint64_t stream_bytes = 0;          // idx + params + 2*lambda bytes per sweep
	bool vector_params = false;         // form of `params` (param_float_index)
	uint32_t max_tab_f4 = 0;            // largest table of a tile (LDS next to the particles)
};
// rewrites seg.params from one form into the other (a permutation inside every 64-slot block)
void relayout_params(FusedSegment &seg, const TypeView *views, bool vector_params);

struct PlanOptions
{
	uint32_t tile_particles = 0;        // 0 = auto
	uint32_t max_local = 10240;         // LDS capacity in particles (160 KiB / 16 B)
	uint32_t num_cus = 256;
	uint32_t max_segment_colours = 16;
	uint32_t max_tile_steps = 64;       // (colour, type) runs one tile may have in one segment
	double launch_cost_ns = 3000.0;     // cost of one more launch (kernel boundary + tail)
	bool owned_stay_in_lds = false;     // persistent schedule: a pass stages only the halo, and a pass boundary is a tile-to-tile hand-off
	uint32_t threads = 0;               // 0 = auto
	bool vector_params = false;         // form of the parameter streams (the caller predicts the workgroup size: vector_params_for_block)
	bool dict_params = false;           // wide, repetitive records go into per-tile dictionaries (FusedStep::dict); the caller reserves kDictTableF4 of max_local
	bool dict_keep_streams = false;     // (set by build_instanced_plan for its prototype: pbdx_plan.cpp dictionary_pass)
	bool bank_aware = true;             // order the slots of every step so that the lanes of an LDS access group hit different banks (lds_bank_model)
	uint32_t sizing_local = 0;          // LDS capacity the default tile SIZE is derived from (0: max_local); the caller that reserved LDS for the tables passes
	                                    // the unreduced capacity, so that the reservation does not change the tile count where the tiles fit anyway
};
// types whose records may take the dictionary form: wide (ten and more streamed planes) and, on regular meshes, highly repetitive -- the bending
// matrices, and the FEM tets' rest volume + Dm^-1 (100 k-tet bar: 92 % of the slots convert, 0.600 -> 0.581 ms; PBDX_DICT_FEM=0: A/B builds).  Not the
// strain tets: their steps may run in quad form (pbdx_quad.h), where every lane fetches the record itself.
#ifndef PBDX_DICT_FEM
#define PBDX_DICT_FEM 1
#endif
constexpr bool dict_type(int type) { return is_bending_type(type) || (PBDX_DICT_FEM != 0 && (type == 9 || type == 10) && !is_quad_type(type)); }
constexpr uint32_t kDictTableF4 = 1024;                  // LDS reserved for a tile's table: 16 KiB
constexpr uint32_t kDictChunkType = 16;                  // chunk type of a dictionary-form step = kDictChunkType + constraint type
constexpr uint32_t dict_entry_f4(uint32_t planes) { return (planes + 3u) / 4u; }

struct FusedPlan
{
	std::vector<FusedSegment> segs;
	TypeView views[16];
	uint32_t num_tiles = 0;
	uint32_t num_colours = 0;
	uint32_t num_particles = 0;
	uint64_t num_constraints = 0;
	std::vector<uint32_t> tile_of;      // particle -> tile
	std::vector<uint32_t> batch_base;   // execution-order batch -> first constraint id
	double redundancy = 0.0;            // executed slots / distinct constraints
	double build_seconds = 0.0;
};

// `x` = packed xyz (partition quality only; never affects results).  Batches in execution order.
bool build_fused_plan(uint32_t n, const float *x, const std::vector<PlanBatch> &batches,
	const PlanOptions &opt, FusedPlan &out, std::string &why);

// Instanced schedules (K congruent, disjoint copies of a prototype with `n_proto` particles, appended instance after instance:
// every batch holds the prototype's constraints K times, copy k's particle indices = the prototype's + k * n_proto).
// check_instancing verifies exactly that (a few host threads; O(total indices)).  build_instanced_plan plans the PROTOTYPE
// (its share of the tiles) and replicates tiles, steps and streams K times with offset particle ids and every copy's own
// parameter records: same kernels, same per-tile work as a plan of the whole -- set-up cost of one instance.
bool check_instancing(uint32_t n_proto, uint32_t K, const std::vector<PlanBatch> &batches);
bool build_instanced_plan(uint32_t n_proto, uint32_t K, const float *x, const std::vector<PlanBatch> &batches,
	const PlanOptions &opt, FusedPlan &out, std::string &why);

// Symbolic execution: every particle carries a hash of its update history; the fused schedule must
// produce, for every particle, the hash the colour-sequential sweep produces.  Also checks that
// every particle is owned exactly once and that every local index is in range.
bool check_fused_plan(uint32_t n, const std::vector<PlanBatch> &batches, const FusedPlan &plan, std::string &why);

// ---- LDS bank model of the colour sweep ---------------------------------------------------------------------
// The projections gather and scatter their endpoints as 16-byte LDS accesses at tile-local slot h (byte address 16 h).  gfx950 services a wave's
// ds_read_b128 in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63} -- over 64 banks of 4
// bytes, and a ds_write_b128 in eight groups of 8 consecutive lanes over 32 banks (MI355X_MICROARCH.md, LDS table): a group takes one LDS cycle if
// its lanes' slots are pairwise distinct mod 16 (reads) / mod 8 (writes), and one more cycle for every further distinct address on the busiest bank.
// Counters of the 1 M cloth (profiles/r04_sq_counters_persistent_c2.log): SQ_LDS_BANK_CONFLICT = 52 % of SQ_LDS_IDX_ACTIVE.  Within a colour step
// the constraints are independent, so their ORDER inside the step (which lane of which wave projects which one) is free: results do not depend on it.
// lds_bank_model() evaluates a plan under this model; PlanOptions::bank_aware makes the planner order every step's slots to reduce the modelled cycles.
struct LdsBankModel
{
	uint64_t read_groups = 0, read_cycles = 0;      // non-empty 16-lane groups of the endpoint gathers (= conflict-free cycles) and modelled cycles
	uint64_t write_groups = 0, write_cycles = 0;    // the same for the scatter (8-lane groups)
	uint64_t table_groups = 0, table_cycles = 0;    // reads of the dictionary records (broadcast for equal records)
};
void lds_bank_model(const FusedPlan &plan, uint32_t block, LdsBankModel &out);
// the read / write groups of a wave (lane -> group), shared by the model and the planner
constexpr uint32_t lds_read_group(uint32_t lane)      // lane in 0..63
{
	return (lane >> 5) * 2u + ((((lane & 31u) < 4u) || ((lane & 31u) >= 12u && (lane & 31u) < 16u) || ((lane & 31u) >= 20u && (lane & 31u) < 28u)) ? 0u : 1u);
}
constexpr uint32_t lds_write_group(uint32_t lane) { return lane >> 3; }

// ---- persistent schedule (all passes of a substep in one launch, tiles synchronised pairwise) -------------
// Per segment a CSR list: the tiles whose pass p-1 must be complete before tile t may start pass p when that
// pass runs segment s.  = owners of t's halo in s (read-after-write) + tiles that had particles of t in their
// halo in the previous segment (write-after-read on the double-buffered positions); t itself never listed
// (a workgroup runs its own passes in order).
struct PersistentDeps
{
	std::vector<std::vector<uint32_t>> off;    // [segment][tile + 1]
	std::vector<std::vector<uint32_t>> tile;   // [segment][...]
};
void build_persistent_deps(const FusedPlan &plan, PersistentDeps &out);

// Asynchronous-execution check of the lists: a host simulation in which every tile alternates FILL (reads its
// local particles from the pass's input buffer) and WRITE-BACK (writes its owned particles to the other buffer,
// then publishes the pass), tiles advancing in adversarial orders (most advanced first, least advanced first,
// one tile held back for as long as the lists allow, pseudo-random) subject only to the lists.  Every FILL must
// find, for every particle it reads, the version written by pass p-1 (or the initial state for pass 0).
// `passes` = iterations x segments to simulate.
bool check_persistent_deps(const FusedPlan &plan, const PersistentDeps &deps, uint32_t passes, std::string &why, bool keep_owned = false, uint32_t workgroups = 0);

} // namespace pbdx

#endif
