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

// which parameters of a type are uniform over the whole schedule (-> scalar), which are streamed
struct TypeView
{
	uint32_t umask;
	uint32_t nplanes;
	float u[PBDX_MAX_PARAMS];
	uint8_t slot[PBDX_MAX_PARAMS];
};

struct FusedStep      // one (colour, type) run of a tile; 32 bytes, read with scalar loads
{
	uint32_t type;
	uint32_t count;
	uint32_t idx_off;     // into the segment's uint16 index stream (2 or 4 entries per slot)
	uint32_t par_off;     // into the segment's float parameter stream (planar: plane p at par_off + p*par_stride)
	uint32_t par_stride;
	uint32_t lam_off;     // into the segment's lambda stream (XPBD types)
	uint32_t barrier;     // workgroup barrier after this step (last step of a colour)
	uint32_t cid_off;     // host only: into slot_cid
};

struct FusedTile      // 32 bytes
{
	uint32_t step_begin, step_end;
	uint32_t n_local;     // particles staged in LDS (owned first)
	uint32_t n_owned;
	uint32_t gid_off;     // into the segment's global-id stream
	uint32_t slots;       // constraints executed by this tile in this segment
	uint32_t pad0, pad1;
};

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
};

struct PlanOptions
{
	uint32_t tile_particles = 0;        // 0 = auto
	uint32_t max_local = 10240;         // LDS capacity in particles (160 KiB / 16 B)
	uint32_t num_cus = 256;
	uint32_t max_segment_colours = 16;
	double launch_cost_bytes = 12.0e6;  // cost of one more launch expressed in streamed bytes
	uint32_t threads = 0;               // 0 = auto
};

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

// Symbolic execution: every particle carries a hash of its update history; the fused schedule must
// produce, for every particle, the hash the colour-sequential sweep produces.  Also checks that
// every particle is owned exactly once and that every local index is in range.
bool check_fused_plan(uint32_t n, const std::vector<PlanBatch> &batches, const FusedPlan &plan, std::string &why);

} // namespace pbdx

#endif
