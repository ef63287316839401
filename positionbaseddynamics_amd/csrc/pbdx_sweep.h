// pbdx_sweep.h -- what the host engine (pbdx_solver.hip) and the constraint-sweep kernels (pbdx_sweep.hip) share: kernel-argument blocks, the device
// form of a tile descriptor, the constants both sides size things with, and the kernel selectors.
#ifndef PBDX_SWEEP_H
#define PBDX_SWEEP_H

#include <hip/hip_runtime.h>
#include "pbdx_internal.h"
#include "pbdx_bounds.h"
#include "pbdx_plan.h"

namespace pbdx {

// ---- (B) one launch per (colour, type) batch -----------------------------------------------------------------------
struct BatchArgs
{
	float4 *pos;
	const uint32_t *idx;
	float *lambda;
	const float *par;
	uint32_t par_stride;
	uint32_t count;
	float dt;                 // substep size (XPBD compliance)
	int first_iter;           // iteration 0 of a substep: lambda := 0 without reading it
	uint32_t num_blocks;      // grid size (for the XCD-aware remap)
	int xcd_remap;
	TypeView view;
};
typedef void (*project_fn)(BatchArgs);
project_fn project_kernel_for(int type, bool compact);

// control block of the persistent schedule (device words; see persistent_kernel)
enum { kCtlAbort = 0, kCtlFailedSubstep = 1, kCtlSubstep = 2, kCtlWords = 4 };

// ---- (A) colour-fused tiles ----------------------------------------------------------------------------------------
// A tile descriptor as the kernels read it: padded to 64 bytes and read with ONE scalar load (constant address space + uniform index -> s_load_dwordx16
// straight into SGPRs).  As a plain global load it came back in vector registers -- a dependent vector-memory round trip followed by eleven
// v_readfirstlane at the start of every pass, behind the scalar loads of the segment's arguments (pass probes, profiles/HISTORY.md [10]).
struct alignas(64) TileDev { FusedTile t; uint32_t pad[5]; };
static_assert(sizeof(TileDev) == 64 && sizeof(FusedTile) == 44, "one s_load_dwordx16 per tile descriptor");

// the plan image of one segment (read-only except the multiplier stream, which is private per tile)
struct SegArgs
{
	const TileDev *tiles;
	const FusedChunk *chunks;
	const uint16_t *idx;
	const float *params;
	float *lambda;
	const uint32_t *gid;
	uint32_t idx_bytes, params_bytes, lambda_bytes;   // stream sizes (buffer descriptors)
	uint32_t num_tiles;
#if PBDX_BOUNDS
	uint32_t gid_count, chunk_count, n_particles, lds_f4;      // sizes of the raw-pointer streams, particle count, LDS capacity behind the chunk header (16-byte units)
	__device__ __forceinline__ uint32_t gid_count_dbg() const { return gid_count; }
#else
	__device__ __forceinline__ uint32_t gid_count_dbg() const { return 0u; }
#endif
};
struct FusedArgs
{
	const float4 *pos_in;
	float4 *pos_out;
	SegArgs seg;
	float dt;
	int first_iter;
	int xcd_remap;
	// developer trace (PBDX_OPT_TRACE): per tile kTraceStride wall-clock stamps (100 MHz):
	// [0] kernel entry, [1] LDS filled, [2+i] step i done (after its barrier), [last] tile written back
	unsigned long long *trace;
	TypeView views[PBDX_NUM_CONSTRAINT_TYPES];
};
constexpr uint32_t kTraceStride = 80;

#ifndef PBDX_DEPTH_SMALL
#define PBDX_DEPTH_SMALL 4     // ring depth for 2-parameter records (distance, volume, dihedral)
#endif
#ifndef PBDX_DEPTH_BIG
#define PBDX_DEPTH_BIG 2       // ring depth for the wide records (bending 11-17, FEM 10-13, shape matching 24 floats)
#endif
// packed records (run_typed PACKED): chunk type of a packed step of a compact one-plane type = kPackedChunkType + type; dictionary-form steps are packed
// when PBDX_PACK_DICT (A/B switch of the build)
#ifndef PBDX_PACK_DICT
#define PBDX_PACK_DICT 1
#endif
constexpr uint32_t kPackedChunkType = 32;
constexpr bool packed_plain_type(int type) { return type == PBDX_DISTANCE || type == PBDX_DISTANCE_XPBD || type == PBDX_VOLUME || type == PBDX_VOLUME_XPBD; }
constexpr int ring_depth(int type) { return kParamCount[type] <= 2 ? PBDX_DEPTH_SMALL : PBDX_DEPTH_BIG; }      // (also used by the host when it expands a plan into chunks)
template <int TYPE> struct Depth { static constexpr int value = ring_depth(TYPE); };
static_assert((PBDX_DEPTH_SMALL == 2 || PBDX_DEPTH_SMALL == 4) && (PBDX_DEPTH_BIG == 2 || PBDX_DEPTH_BIG == 4), "ring depth must be 2 or 4");

constexpr uint32_t kMaxTileChunks = 256;
constexpr uint32_t kMaxTileSteps = 64;

// Integration and velocity update folded into the persistent launch (first / last pass of a substep): the two
// streaming kernels around the sweeps, their kernel boundaries and one round trip of the positions disappear.
// The arithmetic is that of integrate_kernel / velocity_kernel, operation for operation; a tile integrates its halo
// particles redundantly (same inputs, same operations as their owners) and stores state only for the ones it owns.
struct FoldArgs
{
	float4 *vel, *old, *last;
	uint32_t state_bytes;          // n * 16 (buffer descriptors)
	float h, gx, gy, gz, inv_h;
	float ghx, ghy, ghz;           // g * h, rounded once on the host exactly as the device would (kernel arguments = SGPRs: the products were hoisted out of
	                               // the pass loop into VGPRs and spilled at the 128-VGPR limit of the 1 024-thread kernel)
	int second_order;
};

// ---- (A') persistent schedule --------------------------------------------------------------------------------------
constexpr uint32_t kMaxPersistSegs = 8;
constexpr unsigned long long kTicksPerMs = 100000ull;           // the wall clock runs at 100 MHz
constexpr unsigned long long kArriveLimitTicks = 100000ull;     // 1 ms: all workgroups of a launch must have started by then
struct PersistArgs
{
	float4 *pos[2];
	SegArgs seg[kMaxPersistSegs];
	const uint32_t *dep_off[kMaxPersistSegs];     // per segment: num_tiles + 1 offsets into dep_tile
	const uint32_t *dep_tile[kMaxPersistSegs];
	unsigned long long *trace[kMaxPersistSegs];   // developer trace of the LAST pass of every segment (or null)
	uint32_t *epoch;                              // per tile: passes completed; then [num_tiles] arrivals, [num_tiles + 1] decision (all zeroed before the launch)
	uint32_t *ctl;                                // kCtl* words (device)
	uint32_t *error;                              // page-locked host words: [0] a dependency wait timed out, [1] launch refused, [2] at which substep
	uint32_t num_segs, passes, num_tiles;
	uint32_t first_iter_passes;                   // passes [0, first_iter_passes) belong to iteration 0 of a substep (multipliers := 0 unread): num_segs for a launch
	                                              // that starts a substep's sweeps, 0 for a later iteration launched on its own (contacts between the iterations)
	uint32_t expect;                              // arrivals that mean "everybody is here" (gridDim.x; one more in the self-test)
	unsigned long long spin_limit;                // bound of a dependency wait in wall-clock ticks (PBDX_OPT_PERSISTENT_TIMEOUT_MS)
	int mute_tile0;                               // self-test of the timeout path: tile 0 never publishes its first pass
	int folded;                                   // pass 0 integrates, the last pass updates the velocities (FoldArgs)
	FoldArgs fold;
	int start;                                    // position buffer pass 0 reads
	float dt;
	// particle ids resident in LDS (LdsIds; one tile per workgroup only): offsets of the two regions from the start of the dynamic LDS in 16-byte units
	// and their capacities in ids; ids_halo_cap == 0: not in use
	uint32_t ids_halo_off16, ids_bnd_off16, ids_halo_cap, ids_bnd_cap;
#if PBDX_BOUNDS
	uint32_t dep_count[kMaxPersistSegs];          // entries of dep_tile
#endif
	TypeView views[PBDX_NUM_CONSTRAINT_TYPES];
};

typedef void (*fused_fn)(FusedArgs);
typedef void (*persist_fn)(PersistArgs);
constexpr uint32_t kMaskClothXpbd = (1u << PBDX_DISTANCE_XPBD) | (1u << PBDX_ISOMETRIC_BENDING_XPBD);
constexpr uint32_t kMaskLight = (1u << PBDX_DISTANCE) | (1u << PBDX_DISTANCE_XPBD) | (1u << PBDX_ISOMETRIC_BENDING) |
	(1u << PBDX_ISOMETRIC_BENDING_XPBD) | (1u << PBDX_VOLUME) | (1u << PBDX_VOLUME_XPBD) | (1u << PBDX_DIHEDRAL);
constexpr uint32_t kMaskAll = (1u << PBDX_NUM_CONSTRAINT_TYPES) - 1u;
// the solid workloads get kernels of their own: the everything-kernel carries the register demand of its heaviest type (shape matching, strain
// tets: 256 VGPRs and spills) into every run
constexpr uint32_t kMaskFemTet = kMaskLight | (1u << PBDX_FEM_TET) | (1u << PBDX_FEM_TET_XPBD);
constexpr uint32_t kMaskStrainTet = kMaskLight | (1u << PBDX_STRAIN_TET);
fused_fn pick_fused_kernel(uint32_t mask, int block);
persist_fn pick_persistent_kernel(uint32_t mask, int block);
// record of the range checks of a PBDX_BOUNDS build (pbdx_bounds.h; the checks live in the sweep kernels): copies it out, optionally clears it
int sweep_bounds_report(uint32_t out[8], int reset);

} // namespace pbdx

#endif
