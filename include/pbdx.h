/* pbdx.h -- C ABI of the MI355X-native PBD/XPBD constraint-projection engine.
 *
 * This is the drop-in boundary for ONE path of
 * InteractiveComputerGraphics/PositionBasedDynamics: the inner solver loop of
 * PBD::TimeStepController::step (reference Simulation/TimeStepController.cpp:75-241,
 * positionConstraintProjection :251-295) plus the SimulationModel / ParticleData
 * state it reads (Simulation/SimulationModel.h:138-263, Simulation/ParticleData.h:86-311).
 *
 * Plain pointers and sizes only; every function returns an int status
 * (PBDX_OK == 0) unless documented otherwise; no exceptions cross the ABI; all
 * host buffers are caller-owned; no global state except the thread-local
 * last-error string.  All compute entry points run hand-written HIP kernels on
 * gfx950 -- there is NO CPU fallback: without a GPU they return
 * PBDX_ERR_NO_DEVICE.
 *
 * Three object kinds:
 *   pbdx_solver   device engine: particle SoA + colour-batched constraint
 *                 schedule + the substep loop.  This is what a reference-side
 *                 TimeStep plug-in binds to.
 *   pbdx_model    host-side mirror of PBD::SimulationModel for particle
 *                 scenes (mesh builders, add*Constraint, greedy colouring) so
 *                 that scenes can be built without the reference.
 *   pbdx_timestep host-side mirror of PBD::TimeStepController (same parameter
 *                 names and defaults) driving a pbdx_solver from a pbdx_model.
 */
#ifndef PBDX_H
#define PBDX_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBDX_VERSION 100

/* ---- status codes ------------------------------------------------------ */
enum {
	PBDX_OK = 0,
	PBDX_ERR_INVALID = 1,      /* bad argument / bad state */
	PBDX_ERR_NO_DEVICE = 2,    /* no HIP device (engine never falls back to CPU) */
	PBDX_ERR_HIP = 3,          /* a HIP runtime call failed; see pbdx_last_error() */
	PBDX_ERR_UNSUPPORTED = 4,  /* constraint type not handled by the engine */
	PBDX_ERR_ALLOC = 5
};

/* ---- constraint types on the path --------------------------------------
 * One enumerator per particle constraint class of the reference
 * (Simulation/Constraints.h:255-491).  The reference's own TYPE_IDs are
 * run-time counters (Simulation/Constraints.cpp:17-49), so a binding maps
 * `X::TYPE_ID` -> these values; never hard-code the reference ids.
 *
 * Host parameter record (floats per constraint, `param_stride`), matrices in
 * Eigen's default column-major order M(r,c) -> [c*rows + r]:
 *   DISTANCE, DISTANCE_XPBD        [restLength, stiffness]                              2
 *   DIHEDRAL                       [restAngle, stiffness]                               2
 *   ISOMETRIC_BENDING(_XPBD)       [stiffness, Q(4x4)]                                  17
 *   FEM_TRIANGLE                   [area, invRestMat(2x2), xx, yy, xy, xyPoisson, yxPoisson]  10
 *   STRAIN_TRIANGLE                [invRestMat(2x2), xx, yy, xy, normStretch, normShear]      9
 *   VOLUME, VOLUME_XPBD            [restVolume, stiffness]                              2
 *   FEM_TET, FEM_TET_XPBD          [volume, invRestMat(3x3), stiffness, poissonRatio]   12
 *   STRAIN_TET                     [invRestMat(3x3), stretch, shear, normStretch, normShear]  13
 *   SHAPE_MATCHING (4 particles)   [stiffness, restCm(3), x0(4x3), w(4), numClusters(4)]      24
 * XPBD multipliers (m_lambda) are engine-owned and reset at iteration 0 of
 * every substep (Simulation/Constraints.cpp:1241,1448,1725,1877).
 */
typedef enum pbdx_constraint_type {
	PBDX_DISTANCE = 0,               /* DistanceConstraint               Constraints.cpp:1166-1206 */
	PBDX_DISTANCE_XPBD = 1,          /* DistanceConstraint_XPBD          Constraints.cpp:1211-1258 */
	PBDX_DIHEDRAL = 2,               /* DihedralConstraint               Constraints.cpp:1264-1339 */
	PBDX_ISOMETRIC_BENDING = 3,      /* IsometricBendingConstraint       Constraints.cpp:1345-1402 */
	PBDX_ISOMETRIC_BENDING_XPBD = 4, /* IsometricBendingConstraint_XPBD  Constraints.cpp:1407-1471 */
	PBDX_FEM_TRIANGLE = 5,           /* FEMTriangleConstraint            Constraints.cpp:1476-1538 */
	PBDX_STRAIN_TRIANGLE = 6,        /* StrainTriangleConstraint         Constraints.cpp:1544-1610 */
	PBDX_VOLUME = 7,                 /* VolumeConstraint                 Constraints.cpp:1617-1680 */
	PBDX_VOLUME_XPBD = 8,            /* VolumeConstraint_XPBD            Constraints.cpp:1686-1750 */
	PBDX_FEM_TET = 9,                /* FEMTetConstraint                 Constraints.cpp:1755-1825 */
	PBDX_FEM_TET_XPBD = 10,          /* XPBD_FEMTetConstraint            Constraints.cpp:1830-1906 */
	PBDX_STRAIN_TET = 11,            /* StrainTetConstraint              Constraints.cpp:1912-1980 */
	PBDX_SHAPE_MATCHING = 12,        /* ShapeMatchingConstraint (4 pts)  Constraints.cpp:1985-2028 */
	PBDX_NUM_CONSTRAINT_TYPES = 13
} pbdx_constraint_type;

/* number of particles / floats per host parameter record of a type (0 if invalid) */
uint32_t pbdx_type_num_bodies(int type);
uint32_t pbdx_type_param_stride(int type);
const char *pbdx_type_name(int type);

/* velocity update (TimeStepController::ENUM_VUPDATE_*, TimeStepController.cpp:67-72) */
enum { PBDX_VUPDATE_FIRST_ORDER = 0, PBDX_VUPDATE_SECOND_ORDER = 1 };

/* thread-local description of the last failure ("" if none) */
const char *pbdx_last_error(void);
int pbdx_version(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int pbdx_device_count(void);
/* Ensemble sharding: the one thing that shards is a set of independent scene instances -- contiguous blocks, one process (or one
 * pbdx_solver on its own device: every entry point selects its solver's device and restores the caller's) per GPU, no data-path collective.
 * [*begin, *end) = the instances of `rank` out of `world`; block sizes differ by at most one.  The python ensemble helper and bench.py use
 * this definition (positionbaseddynamics_amd/ensemble.py); a C++ host with several solvers in one process uses it the same way. */
int pbdx_ensemble_shard(uint64_t total, uint32_t world, uint32_t rank, uint64_t *begin, uint64_t *end);

/* ======================================================================== */
/* pbdx_solver -- the device engine                                         */
/* ======================================================================== */
typedef struct pbdx_solver pbdx_solver;

/* Create an engine on HIP device `device` with its own stream.
 * Replaces: construction of PBD::TimeStepController (TimeStepController.cpp:23-32). */
int pbdx_solver_create(pbdx_solver **out, int device);
void pbdx_solver_destroy(pbdx_solver *s);

/* Upload particle state.  Arrays are packed xyz triples exactly as
 * ParticleData's std::vector<Vector3r> (Simulation/ParticleData.h:91-100) in a
 * float build; `mass`/`inv_mass` as m_masses/m_invMasses.  `v`, `old_x`,
 * `last_x` may be NULL (=> 0, x, x).  Accelerations are not uploaded: the
 * reference overwrites them with gravity for every dynamic particle at the
 * start of each step (TimeStep::clearAccelerations, TimeStep.cpp:28-62).
 * A particle is static iff mass == 0 (TimeIntegration.cpp:14).
 * Changing the particle COUNT drops the constraint schedule (its indices refer to the old image): re-add it.
 * A pbdx_solver is not thread-safe; use one per host thread / device stream. */
int pbdx_solver_set_particles(pbdx_solver *s, uint32_t n,
	const float *x, const float *v, const float *old_x, const float *last_x,
	const float *mass, const float *inv_mass);
/* The same for a host whose Real is double (the reference's default build, Common/Common.h:10-28): the arrays are copied
 * as they are and converted to the device's fp32 by one kernel (one rounding per value, like a (float) cast on the host). */
int pbdx_solver_set_particles_f64(pbdx_solver *s, uint32_t n,
	const double *x, const double *v, const double *old_x, const double *last_x,
	const double *mass, const double *inv_mass);
int pbdx_solver_get_particles_f64(pbdx_solver *s, uint32_t n,
	double *x, double *v, double *old_x, double *last_x);
/* Overwrite positions only (n must match); used for teacher-forced parity. */
int pbdx_solver_set_positions(pbdx_solver *s, uint32_t n, const float *x);

/* Download state; any pointer may be NULL.  Synchronises the engine stream. */
int pbdx_solver_get_particles(pbdx_solver *s, uint32_t n,
	float *x, float *v, float *old_x, float *last_x);

/* ---- Dirty tracking of a host mirror ------------------------------------------------------------
 * A host application that cannot be instrumented (the reference's ParticleData hands out plain references) is tracked by
 * FULL-COVERAGE block hashes of its arrays: one 64-bit hash per PBDX_HASH_BLOCK consecutive elements, the XOR over the
 * block's 64-bit words of pbdx_hash_word(word, index of the word in the array) -- order-free, so the host evaluates it with
 * as many threads as it likes and the device with one workgroup per block.  A single changed word always changes its
 * block's hash.  pbdx_solver_get_particles_hashed downloads like pbdx_solver_get_particles and ALSO returns the block hashes
 * of exactly the bytes it delivered (computed on the device's staging copy: the host does not read 48 MB again to know what
 * it has); hash arrays hold pbdx_hash_num_blocks(n) values, any of them may be NULL. */
#define PBDX_HASH_BLOCK 1024u
#if defined(__HIPCC__)
#define PBDX_HOST_DEVICE __host__ __device__
#else
#define PBDX_HOST_DEVICE
#endif
PBDX_HOST_DEVICE static inline uint64_t pbdx_hash_word(uint64_t word, uint32_t index)
{
	uint64_t m = (word ^ ((uint64_t)index * 0x9E3779B97F4A7C15ull)) * 0xD1B54A32D192ED03ull;
	return m ^ (m >> 29);
}
static inline uint32_t pbdx_hash_num_blocks(uint32_t n) { return (n + PBDX_HASH_BLOCK - 1u) / PBDX_HASH_BLOCK; }
/* hash of block `block` of an array of n elements of elem_bytes bytes each (a multiple of 4; host side).  The array is read as 64-bit words
 * (a block of 1024 elements is a whole number of them); an odd 32-bit word at the very end of the array counts as a 64-bit word with a zero
 * upper half. */
static inline uint64_t pbdx_hash_block(const void *base, uint32_t n, uint32_t elem_bytes, uint32_t block)
{
	const unsigned char *b = (const unsigned char *)base;
	const uint64_t total = (uint64_t)n * elem_bytes;
	const uint64_t first = (uint64_t)block * PBDX_HASH_BLOCK * elem_bytes;
	uint64_t last = first + (uint64_t)PBDX_HASH_BLOCK * elem_bytes, h = 0, o;
	if (last > total) last = total;
	/* (index * constant advances by the constant from word to word: one multiplication per word is left; four independent accumulators) */
	{
		const uint64_t K = 0x9E3779B97F4A7C15ull, M = 0xD1B54A32D192ED03ull;
		uint64_t k = (uint64_t)(uint32_t)(first >> 3) * K, h1 = 0, h2 = 0, h3 = 0;
		for (o = first; o + 32 <= last; o += 32)
		{
			uint64_t w[4], m0, m1, m2, m3;
			memcpy(w, b + o, 32);
			m0 = (w[0] ^ k) * M; m1 = (w[1] ^ (k + K)) * M; m2 = (w[2] ^ (k + 2 * K)) * M; m3 = (w[3] ^ (k + 3 * K)) * M;
			h ^= m0 ^ (m0 >> 29); h1 ^= m1 ^ (m1 >> 29); h2 ^= m2 ^ (m2 >> 29); h3 ^= m3 ^ (m3 >> 29);
			k += 4 * K;
		}
		h ^= h1 ^ h2 ^ h3;
	}
	for (; o + 8 <= last; o += 8)
	{
		uint64_t w;
		memcpy(&w, b + o, 8);
		h ^= pbdx_hash_word(w, (uint32_t)(o >> 3));
	}
	if (o < last)
	{
		uint32_t t;
		memcpy(&t, b + o, 4);
		h ^= pbdx_hash_word((uint64_t)t, (uint32_t)(o >> 3));
	}
	return h;
}
int pbdx_solver_get_particles_hashed(pbdx_solver *s, uint32_t n, float *x, float *v, float *old_x, float *last_x,
	uint64_t *hash_x, uint64_t *hash_v, uint64_t *hash_old, uint64_t *hash_last);
int pbdx_solver_get_particles_hashed_f64(pbdx_solver *s, uint32_t n, double *x, double *v, double *old_x, double *last_x,
	uint64_t *hash_x, uint64_t *hash_v, uint64_t *hash_old, uint64_t *hash_last);
/* Partial upload: element ranges (num_ranges pairs first, count) of ONE of the caller's arrays replace the device's values;
 * everything else on the device stays.  `base` is the start of the WHOLE array (element 0), as for pbdx_solver_set_particles.
 * The particle count must be the uploaded one (a first upload is pbdx_solver_set_particles). */
enum { PBDX_ARRAY_X = 0, PBDX_ARRAY_V = 1, PBDX_ARRAY_OLD_X = 2, PBDX_ARRAY_LAST_X = 3, PBDX_ARRAY_MASS = 4, PBDX_ARRAY_INV_MASS = 5 };
int pbdx_solver_update_particle_ranges(pbdx_solver *s, int array, const float *base, uint32_t num_ranges, const uint32_t *ranges);
int pbdx_solver_update_particle_ranges_f64(pbdx_solver *s, int array, const double *base, uint32_t num_ranges, const uint32_t *ranges);

/* Constraint schedule = the reference's colour groups
 * (SimulationModel::getConstraintGroups(), SimulationModel.cpp:1033-1094).
 * begin -> add_batch* -> end.  `group` is the colour index: groups execute in
 * increasing order each iteration (Gauss-Seidel across colours,
 * TimeStepController.cpp:270-286); batches of the same group touch disjoint
 * particles and may run in any order.  `indices` holds count*num_bodies(type)
 * particle indices (Constraint::m_bodies order), `params` count*param_stride
 * floats laid out as documented above.  Order inside a batch is kept. */
int pbdx_solver_begin_schedule(pbdx_solver *s);
int pbdx_solver_add_batch(pbdx_solver *s, uint32_t group, int type, uint32_t count,
	const uint32_t *indices, const float *params, uint32_t param_stride);
int pbdx_solver_end_schedule(pbdx_solver *s);
/* Run-time parameter edits (the reference's setClothStiffness / m_stiffness / m_restLength edits between steps,
 * SimulationModel.h setConstraintValue<>): replace the parameter records of batch `batch_index` (order of add_batch calls;
 * same count and stride) and commit.  The committed schedule keeps its colouring, tiles and launch plan -- only the
 * parameter streams and scalar kernel arguments are refreshed (no replanning, no re-measurement) unless the edit changes
 * whether a type's shared parameters are uniform (then the plan is rebuilt by the next step). */
int pbdx_solver_update_batch_params(pbdx_solver *s, uint32_t batch_index, uint32_t count, const float *params, uint32_t param_stride);
int pbdx_solver_commit_params(pbdx_solver *s);
/* Hint for ensemble schedules: the particles are `instances` blocks of `particles_per_instance`, and every batch holds the
 * constraints of instance 0, then of instance 1, ... with the same block-local particle indices (what K rounds of the same
 * builder calls produce).  The engine verifies the hint when it plans; if it holds, ONE instance is planned and the tile
 * schedule replicated (set-up time of one instance); if not, the whole schedule is planned as usual.  instances <= 1 clears
 * the hint.  Never changes a result. */
int pbdx_solver_set_instancing(pbdx_solver *s, uint32_t particles_per_instance, uint32_t instances);
/* Debug validator: every group's batches touch pairwise-disjoint particles
 * (the invariant data-race freedom rests on).  Returns PBDX_OK or PBDX_ERR_INVALID. */
int pbdx_solver_validate_schedule(pbdx_solver *s);

/* ---- One substep in pieces: mixed models ------------------------------------------------------------------
 * A SimulationModel may hold constraint classes the engine does not know (the reference's PositionBasedGenericConstraints.h:31-218
 * templates, user subclasses of PBD::Constraint).  The reference runs them through the virtual solvePositionConstraint inside the
 * same colour groups (TimeStepController.cpp:270-286).  A host that wants the engine for the groups' KNOWN batches drives the substep
 * itself with the three calls below and runs its own constraints of group g between project_groups(.., g, g + 1) and the next group,
 * on positions it fetched with pbdx_solver_get_particles and returns with pbdx_solver_update_particle_ranges(PBDX_ARRAY_X):
 *     integrate;  for iteration: for group: project_groups / host constraints;  update_velocities
 * -- exactly TimeStepController.cpp:112-160.  Per-colour launches, synchronous (each call returns when the device is done); every
 * transfer in between is the caller's: correct, and slow by construction (reference-side plug-in: numMixedGroups()). */
int pbdx_solver_integrate(pbdx_solver *s, float h_sub, const float gravity[3]);      /* TimeStepController.cpp:112-129, TimeIntegration.cpp:7-19 */
/* the engine's batches of colour groups [group_begin, group_end) for iteration `iteration` of the substep (0: XPBD multipliers := 0) */
int pbdx_solver_project_groups(pbdx_solver *s, float h_sub, uint32_t iteration, uint32_t group_begin, uint32_t group_end);
int pbdx_solver_update_velocities(pbdx_solver *s, float h_sub, int velocity_update_method);   /* TimeStepController.cpp:142-160 */

/* Advance `num_steps` full time steps of size h, device-resident (no host
 * transfers).  One step == TimeStepController::step for a particle scene:
 *   a_i = gravity (dynamic particles); hs = h/sub_steps;
 *   repeat sub_steps: integrate -> max_iterations x (groups in order) -> velocity update.
 * Replaces TimeStepController.cpp:75-176 + positionConstraintProjection :251-295. */
int pbdx_solver_step(pbdx_solver *s, float h, uint32_t sub_steps, uint32_t max_iterations,
	int velocity_update_method, const float gravity[3], uint32_t num_steps);
/* Run only the projection loop of ONE substep on the current positions (no
 * integrate / velocity update): `iterations` Gauss-Seidel sweeps, lambda reset
 * at sweep 0, XPBD dt = h_sub.  Used by known-answer tests. */
int pbdx_solver_project(pbdx_solver *s, float h_sub, uint32_t iterations);
int pbdx_solver_synchronize(pbdx_solver *s);
/* Checkpoint of the device-resident particle state (x, v, oldX, lastX) and its restoration: stream-ordered device-to-device copies, no host
 * synchronisation.  The reference-side plug-in steps speculatively while its exact parameter scan still runs on the host; an edit found by the scan
 * undoes the step with restore_state (plugin/TimeStepControllerHIP.cpp).  The checkpoint is dropped when the particle set changes. */
int pbdx_solver_save_state(pbdx_solver *s);
int pbdx_solver_restore_state(pbdx_solver *s);

/* ---- contacts with static rigid bodies ------------------------------------------
 * Particle vs static rigid body contacts with analytic distance fields, i.e. the part of
 * DistanceFieldCollisionDetection::collisionDetection (DistanceFieldCollisionDetection.cpp:26-197,281-358)
 * and of TimeStepController::velocityConstraintProjection (TimeStepController.cpp:298-355) that concerns
 * ParticleRigidBodyContactConstraints (Constraints.cpp:2115-2189) when every rigid body is static
 * (mass 0): collision detection once per step after the substeps, then `max_iterations_v` velocity
 * sweeps.  With static bodies the contact list decomposes into independent per-particle chains, so the
 * device result equals the reference's sequential sweep.  Assumption (true for the reference's demos):
 * the collision object's AABB covers the region where its distance field is negative, so the
 * reference's bounding-volume culling never removes a penetrating particle. */
enum {
	PBDX_SHAPE_BOX = 0,            /* params: half extents (m_box = 0.5 * box)          DistanceFieldCollisionBox */
	PBDX_SHAPE_SPHERE = 1,         /* params[0] radius                                   DistanceFieldCollisionSphere */
	PBDX_SHAPE_TORUS = 2,          /* params[0..1] radii                                 DistanceFieldCollisionTorus */
	PBDX_SHAPE_CYLINDER = 3,       /* params[0] radius, [1] half height (m_dim)          DistanceFieldCollisionCylinder */
	PBDX_SHAPE_HOLLOW_SPHERE = 4,  /* params[0] radius, [1] thickness                    DistanceFieldCollisionHollowSphere */
	PBDX_SHAPE_HOLLOW_BOX = 5      /* params[0..2] half extents, [3] thickness           DistanceFieldCollisionHollowBox */
};
typedef struct pbdx_collider {
	int shape;
	int invert;                    /* m_invertSDF == -1 */
	float params[4];
	float com[3];                  /* RigidBody::getPosition() */
	float R[9];                    /* RigidBody::getTransformationR(), row-major: x_local = R (x_world - com) + v1 */
	float v1[3], v2[3];            /* getTransformationV1 / V2:  x_world = R^T x_local + v2 */
	float restitution, friction;   /* of the rigid body */
	float body_v[3], body_omega[3];/* velocity / angular velocity of the (static or kinematic) body */
	uint32_t body_index;
} pbdx_collider;
/* particles [first, first+count) belong to a triangle / tet model registered as a collision object */
typedef struct pbdx_collision_range { uint32_t first, count; float restitution, friction; } pbdx_collision_range;
int pbdx_solver_set_colliders(pbdx_solver *s, uint32_t n, const pbdx_collider *colliders);
int pbdx_solver_set_collision_ranges(pbdx_solver *s, uint32_t n, const pbdx_collision_range *ranges);
/* tolerance = CollisionDetection::m_tolerance (default 0.01), contact_stiffness =
 * SimulationModel::m_contactStiffnessParticleRigidBody (default 100), max_iterations_v = "maxIterationsV" (default 5) */
int pbdx_solver_set_contact_params(pbdx_solver *s, float tolerance, float contact_stiffness, uint32_t max_iterations_v);
/* contacts found by the last step (sum over particles).  A particle with more than 8 simultaneous contacts is an error:
 * pbdx_solver_step itself returns PBDX_ERR_UNSUPPORTED for such a call (the reference has no per-particle limit). */
int pbdx_solver_get_num_contacts(pbdx_solver *s, uint32_t *out);

/* ---- dynamic rigid bodies as impulse sinks (the remainder) ------------------------------------------------
 * A rigid body of finite mass keeps its state and its time integration on the host (rigid-body dynamics are outside this path:
 * TimeStepController.cpp:94-104,137-152 stay the host's).  What the engine takes over is the body's part in
 * ParticleRigidBodyContactConstraint::solveVelocityConstraint (Constraints.cpp:2148-2189): the contact impulses change the body's
 * velocity and angular velocity, which the next contact of the same body reads.  The contacts of particles that touch a dynamic
 * body (all their contacts, also the ones with static bodies) are therefore collected into one list in the REFERENCE'S ORDER and
 * solved sequentially on the device; every other particle keeps its independent chain.  The reference's order
 * (DistanceFieldCollisionDetection.cpp:34-47,102-177,197-213): collision-object pairs (i, k) lexicographically, and inside a pair the
 * depth-first, left-to-right walk of object i's point hierarchy, i.e. ascending position in its entity list (kdTree.inl:84-105).
 * The host hands over that position per particle (`rank`) and the object indices of the ranges and colliders.
 * Before every step: pbdx_solver_set_colliders with the bodies' poses and velocities at the END of the step's substeps (the host has
 * integrated them), pbdx_solver_set_collider_dynamics; after the step: pbdx_solver_get_body_velocities = the bodies' velocities after
 * the contact solve (the host writes them into its bodies, Constraints.cpp:2180-2187).
 * Not combined with contacts between deformable solids (pbdx_solver_set_tet_colliders): PBDX_ERR_UNSUPPORTED. */
typedef struct pbdx_collider_dynamics {
	float inv_mass;                /* RigidBody::getInvMass(); 0: static */
	float inertia_inv_w[9];        /* RigidBody::getInertiaTensorInverseW(), row-major */
	uint32_t object_index;         /* position of the body's collision object in CollisionDetection::getCollisionObjects() */
	uint32_t pad;
} pbdx_collider_dynamics;
/* n = number of colliders (same order as pbdx_solver_set_colliders), or 0: every body static again */
int pbdx_solver_set_collider_dynamics(pbdx_solver *s, uint32_t n, const pbdx_collider_dynamics *dyn);
/* range_object_index[r]: position of collision range r's object in the collision-object list; rank[i] for every particle i of the
 * engine (n_particles entries): its position in its collision object's point hierarchy (ignored for particles outside the ranges) */
int pbdx_solver_set_contact_order(pbdx_solver *s, uint32_t n_ranges, const uint32_t *range_object_index, uint32_t n_particles, const uint32_t *rank);
/* v, omega: 3 floats per collider */
int pbdx_solver_get_body_velocities(pbdx_solver *s, uint32_t n, float *v, float *omega);

/* ---- contacts between deformable solids (second half) -----------------------------------------------------
 * Tet models that carry an analytic distance field in their rest frame (DistanceFieldCollisionDetection::addCollisionBox / ...
 * on a TetModelCollisionObjectType body + initTetBVH, DistanceFieldCollisionDetection.cpp:496-509,730-742) colliding with the
 * particles of the other tet models: detection = collisionDetectionSolidSolid (:361-483: dual bounding-sphere-hierarchy traversal,
 * point-in-tet, rest-frame distance field, findRefTetAt :744-812) once per step after the substeps; the resulting
 * ParticleTetContactConstraints are solved sequentially inside the iteration loop of the NEXT step
 * (TimeStepController.cpp:288-291, Constraints.cpp:2236-2277, PositionBasedDynamics.cpp:1220-1265).
 * The bounding-sphere hierarchies are built by the host application (the reference constructs them with std::sort on tied
 * coordinates when the collision object is registered): the engine takes their structure -- entity order, nodes = (child0, child1,
 * begin, count), -1 = no child -- and refreshes the spheres every step like KDTree::update; the rest-pose tet hierarchy also needs its
 * (static) spheres.  Contacts are produced, and solved, in the reference's list order for ONE OpenMP thread (the reference's own
 * order is thread-count dependent, DistanceFieldCollisionDetection.cpp:176-196).  Friction of these contacts must be 0: the
 * reference's friction impulse reads an uninitialised multiplier (see pbdx_tetcontact.h); with 0 the velocity solve is a no-op.
 * While tet colliders are set the sweeps of a substep run as one launch per segment / colour (the contact solve sits between
 * the iterations). */
typedef struct pbdx_bvh
{
	uint32_t num_nodes, num_entities;
	const uint32_t *entities;      /* kd-tree entity order (m_lst) */
	const int32_t *nodes;          /* 4 per node: child0, child1, begin, count */
	const float *hulls;            /* 4 per node (centre, radius): required for the rest-pose hierarchy, ignored for the others */
} pbdx_bvh;
typedef struct pbdx_tet_collider
{
	int shape, invert;             /* PBDX_SHAPE_*, m_invertSDF == -1 */
	float params[4];               /* as pbdx_collider */
	uint32_t first_particle, num_vertices, num_tets;
	const uint32_t *tets;          /* 4 model-local vertex indices per tet (IndexedTetMesh::getTets) */
	float initial_x[3], initial_R[9]; /* TetModel::getInitialX / getInitialR (row-major): rest frame of the distance field */
	float restitution, friction;   /* of the tet model (friction must be 0: the reference's friction impulse for these contacts reads an unset multiplier;
	                                * with friction 0 its velocity solve is defined and implemented, including the pMax < 0 branch) */
	int test_mesh;                 /* m_testMesh: the model's particles are tested against the other solids */
	uint32_t body_index;           /* tet model index (reported in the contacts) */
	pbdx_bvh points, tets_bvh, tets_rest; /* m_bvh, m_bvhTets, m_bvhTets0 */
} pbdx_tet_collider;
/* tolerance = CollisionDetection::m_tolerance.  Needs rest positions: pbdx_solver_set_rest_positions (default: the positions of
 * the first pbdx_solver_set_particles call).  Replaces the previous set and empties the contact list (n = 0: no deformable colliders); after a
 * failed call (invalid records, friction != 0, out of memory) NO deformable colliders are set.  The detection's scratch buffers grow on demand. */
int pbdx_solver_set_tet_colliders(pbdx_solver *s, uint32_t n, const pbdx_tet_collider *colliders, float tolerance);
int pbdx_solver_set_rest_positions(pbdx_solver *s, uint32_t n, const float *x0);   /* ParticleData::m_x0, packed xyz */
/* The contact list of the last detection, 34 floats per contact: particle, solid, tet, bary[3], normal[3], 1/(J M^-1 J^T),
 * m_x[4][3], m_invMasses[4], tet vertex particle ids[4] (indices as floats: colliders reaching indices above 2^24 are refused),
 * tangent[3], maximal tangent impulse (m_constraintInfo.col(1), (1, 2)).  *count = number of contacts (may exceed capacity). */
#define PBDX_TET_CONTACT_FLOATS 34
int pbdx_solver_get_tet_contacts(pbdx_solver *s, uint32_t capacity, uint32_t *count, float *out);
/* XPBD multipliers of batch `batch_index` (order of add_batch calls). */
int pbdx_solver_get_lambdas(pbdx_solver *s, uint32_t batch_index, uint32_t count, float *out);

/* Launch options.  None of them changes a result bit: they select how the same colour-ordered
 * Gauss-Seidel sweep (TimeStepController.cpp:270-286) is mapped onto launches. */
enum {
	PBDX_OPT_USE_GRAPH = 1,        /* capture one substep into a hipGraph (default 1) */
	PBDX_OPT_BLOCK_SIZE = 2,       /* threads per workgroup of the per-colour kernels: 64/128/256 (default 256) */
	PBDX_OPT_XCD_REMAP = 3,        /* XCD-aware blockIdx -> constraint-range / tile mapping (default 1) */
	PBDX_OPT_FUSE = 4,             /* 1 = colour-fused LDS tile schedule, 0 = one launch per (colour, type), 2 = auto (default):
	                                * fused, except that schedules with compute-heavy types (FEM, strain, shape matching) are timed
	                                * once both ways on scratch positions and the faster one is kept (results are identical) */
	PBDX_OPT_TILE_PARTICLES = 5,   /* particles owned by one tile; 0 = auto (default) */
	PBDX_OPT_FUSE_BLOCK = 6,       /* threads per workgroup of the fused kernel: 0 = auto, 256, 512, 1024 */
	PBDX_OPT_MAX_SEGMENT_COLOURS = 7, /* upper bound on colours fused into one launch (default 16) */
	PBDX_OPT_LDS_PARTICLES = 8,    /* LDS capacity of a tile in particles (default 10240 = 160 KiB / 16 B) */
	PBDX_OPT_TRACE = 9,            /* developer aid: fused kernels stamp wall_clock64() per tile and colour step */
	PBDX_OPT_PIN_HOST = 11,        /* particle transfers (set/get_particles, update_particle_ranges) go through a page-locked mirror the ENGINE owns
	                                * (56 B per particle of page-locked host memory; 112 B for a double host) and run at the PCIe rate: the device
	                                * copies to / from the mirror, host threads copy between the mirror and the caller's arrays, array by array,
	                                * overlapped.  The caller's memory is never registered with the GPU (until round 5 it was: hipHostRegister of
	                                * heap arrays; HISTORY [9] has why that went).  Default 0 */
	PBDX_OPT_PAIRS = 10,           /* removed (round 2): projected two chunks of a colour step jointly with packed fp32 arithmetic, measured 10-30 % slower; accepted, ignored */
	PBDX_OPT_PERSISTENT = 12,      /* fused schedule only: all sweeps of a substep as ONE launch; a tile starts its next pass as soon as its
	                                * neighbouring tiles have published theirs (no kernel boundary, no chip-wide wait for the slowest tile).
	                                * 1 (default) = used unless a one-off measurement on scratch positions finds it clearly slower than one launch per
	                                * segment, 0 = never, 2 = always (if the plan is eligible), 3 = self-test (the launch is made to refuse),
	                                * 4 = self-test (a tile is made to time out waiting for its neighbour).  Needs every workgroup co-resident: the
	                                * launch first checks that (bounded handshake); if not, it modifies nothing, the engine completes the step
	                                * with one launch per segment and stops using the schedule (pbdx_solver_describe: persistent_refusals).
	                                * A wait inside the launch is bounded as well (PBDX_OPT_PERSISTENT_TIMEOUT_MS): if it expires, the engine restores the
	                                * particle state it saved at the start of the call, repeats the call with one launch per segment and stops
	                                * using the schedule (pbdx_persistent_info::timeouts) -- the caller sees the result of an undisturbed run. */
	PBDX_OPT_PERSISTENT_TIMEOUT_MS = 13, /* bound of a tile-to-tile wait inside the persistent launch in milliseconds, 1 .. 10000 (default 250) */
	PBDX_OPT_PERSISTENT_WGS_PER_CU = 14, /* tiles (workgroups) the persistent launch keeps resident per CU, 1 .. 4 (default 1): with k > 1 the LDS is split k ways
	                                * (smaller tiles, more halo) and one tile's fill / hand-off overlaps another tile's colour sweep on the same CU */
	PBDX_OPT_TET_CONTACTS_SERIAL = 15,   /* developer cross-check: 1 = detect and solve the contacts between deformable solids in ONE thread, in the reference's own
	                                      * control flow (default 0: the parallel, order-preserving form of pbdx_tetcontact_dev.h; both give the same bits) */
	PBDX_OPT_TET_FORCE_IMPULSES = 16,    /* developer aid (test of the velocity-impulse application path of particle-tet contacts): 1 = contacts with pMax > 0
	                                      * are given the impulse the reference gives contacts with pMax < 0 -- NOT the reference's result; default 0 */
	PBDX_OPT_SUBSTEP_EVENTS = 17,        /* measurement (SURVEY 8d "hipEvents around the device-resident substep loop ... median"): 1 = pbdx_solver_step records one
	                                      * HIP event after every substep on the engine's stream; pbdx_solver_get_substep_times returns the device time of
	                                      * each substep of the last call.  n > 1: the same, and n events are created right away (the first
	                                      * measured call then creates none).  Default 0 */
};
int pbdx_solver_set_option(pbdx_solver *s, int option, int64_t value);

/* The colour-fused schedule the engine planned for the current constraint schedule (built lazily by
 * the first pbdx_solver_step / pbdx_solver_project).  active == 0: the per-colour schedule runs. */
typedef struct pbdx_plan_info {
	int built, active;
	uint32_t num_segments;         /* launches per Gauss-Seidel sweep */
	uint32_t num_tiles, num_colours, max_local;
	uint64_t slots_per_sweep;      /* constraint executions incl. redundant halo copies */
	uint64_t stream_bytes_per_sweep; /* index + parameter + multiplier bytes streamed from HBM per sweep */
	double redundancy;             /* slots / distinct constraints */
	double build_seconds;
	uint64_t compulsory_stream_bytes_per_sweep; /* the same streams without halo redundancy: every distinct constraint's record (16-bit
	                                * indices, streamed parameter planes, multiplier read + write) exactly once per sweep */
} pbdx_plan_info;
int pbdx_solver_get_plan_info(pbdx_solver *s, pbdx_plan_info *out);
typedef struct pbdx_segment_info {
	uint32_t colour_begin, colour_end, num_tiles, block, lds_bytes, type_mask;
	uint64_t constraints, slots, stream_bytes;
	uint64_t algorithmic_bytes;    /* SURVEY 8d bytes of the distinct constraints of the segment (one launch) */
	double profiled_ms;            /* last profiled pbdx_solver_step: summed launch time, launches */
	uint64_t profiled_launches;
} pbdx_segment_info;
int pbdx_solver_get_segment_info(pbdx_solver *s, uint32_t segment, pbdx_segment_info *out);
/* The one-launch form of the fused schedule (PBDX_OPT_PERSISTENT). */
typedef struct pbdx_persistent_info {
	int eligible, active;          /* the plan can run as one launch per substep / the schedule in use: 1 = one launch per substep, 2 = one launch per ITERATION
	                                * (scenes with contacts between deformable solids: the contact list is solved between the iterations) */
	uint32_t grid, block, lds_bytes; /* launch geometry */
	uint32_t refusals;             /* launches that found their workgroups not co-resident (the engine then fell back for good) */
	uint32_t timeouts;             /* calls in which a tile-to-tile wait expired (state restored, call repeated with one launch per segment) */
	int last_folded;               /* the substeps enqueued last ran integration and velocity update inside the persistent launch (one launch per substep) */
	double autotune_fused_ms, autotune_persistent_ms; /* the one-off measurement: 12 sweeps on scratch positions, 0 = not measured */
	double profiled_ms;            /* last profiled pbdx_solver_step: summed duration and number of persistent launches */
	uint64_t profiled_launches;
	uint64_t algorithmic_bytes_per_sweep; /* SURVEY 8d bytes of one Gauss-Seidel sweep (a launch runs `iterations` sweeps) */
} pbdx_persistent_info;
int pbdx_solver_get_persistent_info(pbdx_solver *s, pbdx_persistent_info *out);
/* Developer trace of the LAST launch of `segment` (PBDX_OPT_TRACE): num_tiles * stride stamps of the
 * 100 MHz wall clock; per tile [0] kernel entry, [1] LDS filled, [2+i] colour step i finished,
 * [stride-1] tile written back. */
int pbdx_solver_get_trace(pbdx_solver *s, uint32_t segment, uint64_t *out, uint32_t capacity, uint32_t *stride);

/* Timing of the last pbdx_solver_step call measured with HIP events on the
 * engine's own stream: total milliseconds, and (if profile_kernels was set)
 * accumulated milliseconds + launch count of the projection kernels only. */
typedef struct pbdx_step_stats {
	double total_ms;            /* whole pbdx_solver_step call on the stream */
	double projection_ms;       /* sum over projection launches (profiled mode only, else 0) */
	uint64_t projection_launches;
	uint64_t projections;       /* constraint projections executed */
	uint64_t kernel_launches;   /* all launches (integrate + projection + velocity) */
	uint64_t algorithmic_bytes; /* SURVEY 8d bytes: sum_type count*bytes_per_projection*iters + particles*140 */
} pbdx_step_stats;
int pbdx_solver_get_stats(pbdx_solver *s, pbdx_step_stats *out);
/* Device time of every substep of the last pbdx_solver_step call, in milliseconds (PBDX_OPT_SUBSTEP_EVENTS; replaces nothing in the
 * reference -- Utils/Timing.h is the nearest): substep k = from the event after substep k - 1 (the call's start event for k = 0) to the
 * event after substep k.  *count = substeps measured (0: option off, per-launch profiling on, or the call recovered from a refused /
 * timed-out persistent launch); at most `capacity` values are written to out_ms (may be NULL). */
int pbdx_solver_get_substep_times(pbdx_solver *s, float *out_ms, uint32_t capacity, uint32_t *count);
/* profile_kernels != 0: bracket every projection launch with HIP events (no graph). */
int pbdx_solver_set_profiling(pbdx_solver *s, int profile_kernels);
/* Per-constraint-type totals of the last profiled pbdx_solver_step: milliseconds summed over
 * that type's launches (event before the launch -> event before the next launch), number of
 * launches, number of projections. */
int pbdx_solver_get_type_stats(pbdx_solver *s, int type, double *ms, uint64_t *launches, uint64_t *projections);
/* SURVEY 8d algorithmic bytes per projection of a type. */
uint32_t pbdx_type_algorithmic_bytes(int type);
/* Device / engine description for logs (device name, CU count, schedule size). */
int pbdx_solver_describe(pbdx_solver *s, char *buf, size_t buf_size);

/* ======================================================================== */
/* pbdx_model -- host mirror of PBD::SimulationModel (particle scenes)      */
/* ======================================================================== */
typedef struct pbdx_model pbdx_model;

int pbdx_model_create(pbdx_model **out);   /* SimulationModel::SimulationModel + init */
void pbdx_model_destroy(pbdx_model *m);
int pbdx_model_cleanup(pbdx_model *m);     /* SimulationModel::cleanup  SimulationModel.cpp:105-126 */
int pbdx_model_reset(pbdx_model *m);       /* SimulationModel::reset    SimulationModel.cpp:270-304 */

/* Mesh builders.  Return the model index (>=0) or -1.
 * addRegularTriangleModel SimulationModel.cpp:831-901; rotation is a row-major 3x3. */
int pbdx_model_add_regular_triangle_model(pbdx_model *m, int width, int height,
	const float translation[3], const float rotation[9], const float scale[2]);
/* addTriangleModel SimulationModel.cpp:806-829 (no UVs: rendering only). */
int pbdx_model_add_triangle_model(pbdx_model *m, uint32_t n_points, uint32_t n_faces,
	const float *points, const uint32_t *indices);
/* addRegularTetModel SimulationModel.cpp:921-1005 */
int pbdx_model_add_regular_tet_model(pbdx_model *m, int width, int height, int depth,
	const float translation[3], const float rotation[9], const float scale[3]);
/* addTetModel SimulationModel.cpp:903-919 */
int pbdx_model_add_tet_model(pbdx_model *m, uint32_t n_points, uint32_t n_tets,
	const float *points, const uint32_t *indices);

/* Instances (the ensemble workloads are K calls of the same builders with K translations).
 * Appends `count` congruent copies of everything the model holds; copy k (1..count) is what the reference builds when the
 * same mesh and constraint builders are called again with translation T + offsets[3(k-1)..] (regular models: positions
 * re-evaluated as R p + T_k; explicit-point models and loose particles: prototype position + offset).  Particles, mesh
 * models, constraints and colour groups are numbered exactly as those K+1 rounds of builder calls would number them, and
 * the colouring is the one the reference computes for them (first-fit colouring of disjoint congruent copies appended in
 * order repeats the prototype's colouring) -- but topology, constraints and groups are stored once and expanded on demand,
 * so set-up cost no longer grows with the instance count.  The model is sealed afterwards (no further add*). */
int pbdx_model_add_instances(pbdx_model *m, uint32_t count, const float *offsets);
uint32_t pbdx_model_num_instances(const pbdx_model *m);   /* 1 for a model without instances */

/* Mesh topology queries (Utils/IndexedFaceMesh.cpp:118-226, IndexedTetMesh.cpp:55-182). */
uint32_t pbdx_model_num_triangle_models(const pbdx_model *m);
uint32_t pbdx_model_num_tet_models(const pbdx_model *m);
uint32_t pbdx_model_triangle_model_index_offset(const pbdx_model *m, uint32_t tm);
uint32_t pbdx_model_tet_model_index_offset(const pbdx_model *m, uint32_t tm);
uint32_t pbdx_model_triangle_model_num_edges(const pbdx_model *m, uint32_t tm);
/* out: num_edges * 4 values (vert0, vert1, face0, face1), 0xffffffff = no face */
int pbdx_model_triangle_model_get_edges(const pbdx_model *m, uint32_t tm, uint32_t *out);
uint32_t pbdx_model_triangle_model_num_vertices(const pbdx_model *m, uint32_t tm);
uint32_t pbdx_model_triangle_model_num_faces(const pbdx_model *m, uint32_t tm);
int pbdx_model_triangle_model_get_faces(const pbdx_model *m, uint32_t tm, uint32_t *out); /* num_faces*3, model-local vertex ids */
uint32_t pbdx_model_tet_model_num_vertices(const pbdx_model *m, uint32_t tm);
uint32_t pbdx_model_tet_model_num_tets(const pbdx_model *m, uint32_t tm);
int pbdx_model_tet_model_get_tets(const pbdx_model *m, uint32_t tm, uint32_t *out);       /* num_tets*4 */
uint32_t pbdx_model_tet_model_num_edges(const pbdx_model *m, uint32_t tm);
int pbdx_model_tet_model_get_edges(const pbdx_model *m, uint32_t tm, uint32_t *out); /* num_edges*2 */

/* ParticleData accessors (Simulation/ParticleData.h:139-260). */
uint32_t pbdx_model_num_particles(const pbdx_model *m);
int pbdx_model_add_vertex(pbdx_model *m, const float x[3]);            /* ParticleData::addVertex */
int pbdx_model_set_mass(pbdx_model *m, uint32_t i, float mass);        /* setMass keeps invMass in sync :239-246 */
/* which: 0=x 1=x0 2=v 3=a 4=oldX 5=lastX (xyz triples); 6=mass 7=invMass (scalars) */
int pbdx_model_get_array(const pbdx_model *m, int which, float *out);
int pbdx_model_set_array(pbdx_model *m, int which, const float *in);
/* zero-copy pointer to the packed position array (pypbd getVertices analogue).  Writes through this
 * pointer cannot be seen by the engine: call pbdx_model_mark_state_dirty afterwards. */
float *pbdx_model_positions_ptr(pbdx_model *m);
/* Dirty tracking of the host particle state: every pbdx_model_set_array bumps an internal version;
 * pbdx_timestep_step_resident re-uploads the host state when it is newer than the device image. */
int pbdx_model_mark_state_dirty(pbdx_model *m);

/* Per-constraint builders (SimulationModel.cpp:565-806); return 1 on success, 0 if
 * the reference's initConstraint would have returned false. */
int pbdx_model_add_distance_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, float stiffness);
int pbdx_model_add_distance_constraint_xpbd(pbdx_model *m, uint32_t p1, uint32_t p2, float stiffness);
int pbdx_model_add_dihedral_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, float stiffness);
int pbdx_model_add_isometric_bending_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, float stiffness);
int pbdx_model_add_isometric_bending_constraint_xpbd(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, float stiffness);
int pbdx_model_add_fem_triangle_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3,
	float xx, float yy, float xy, float xy_poisson, float yx_poisson);
int pbdx_model_add_strain_triangle_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3,
	float xx, float yy, float xy, int normalize_stretch, int normalize_shear);
int pbdx_model_add_volume_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, float stiffness);
int pbdx_model_add_volume_constraint_xpbd(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, float stiffness);
int pbdx_model_add_fem_tet_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, float stiffness, float poisson);
int pbdx_model_add_fem_tet_constraint_xpbd(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, float stiffness, float poisson);
int pbdx_model_add_strain_tet_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4,
	float stretch, float shear, int normalize_stretch, int normalize_shear);
int pbdx_model_add_shape_matching_constraint(pbdx_model *m, uint32_t n, const uint32_t *particles,
	const uint32_t *num_clusters, float stiffness);   /* n must be 4 */

/* Bulk builders: addClothConstraints :1125-1184 (method 1 distance, 2 FEM tri,
 * 3 strain tri, 4 XPBD distance), addBendingConstraints :1186-1240 (1 dihedral,
 * 2 isometric, 3 isometric XPBD), addSolidConstraints :1242-1349 (1 distance+
 * volume, 2 FEM tet, 3 XPBD FEM tet, 4 strain tet, 5 shape matching, 6 XPBD
 * distance+volume). */
int pbdx_model_add_cloth_constraints(pbdx_model *m, uint32_t tri_model, uint32_t method,
	float distance_stiffness, float xx, float yy, float xy, float xy_poisson, float yx_poisson,
	int normalize_stretch, int normalize_shear);
int pbdx_model_add_bending_constraints(pbdx_model *m, uint32_t tri_model, uint32_t method, float stiffness);
int pbdx_model_add_solid_constraints(pbdx_model *m, uint32_t tet_model, uint32_t method,
	float stiffness, float poisson, float volume_stiffness, int normalize_stretch, int normalize_shear);

/* Constraint inspection. */
uint32_t pbdx_model_num_constraints(const pbdx_model *m);
int pbdx_model_constraint_type(const pbdx_model *m, uint32_t c);            /* pbdx_constraint_type */
int pbdx_model_constraint_bodies(const pbdx_model *m, uint32_t c, uint32_t *out); /* num_bodies values */
int pbdx_model_constraint_params(const pbdx_model *m, uint32_t c, float *out);    /* param_stride values */
int pbdx_model_set_constraint_params(pbdx_model *m, uint32_t c, const float *in); /* python-mutable fields */

/* Greedy first-fit colouring in creation order; cached until the next add*
 * (SimulationModel::initConstraintGroups, SimulationModel.cpp:1033-1094). */
int pbdx_model_init_constraint_groups(pbdx_model *m);
/* The same colouring computed on the device, group for group identical (pbdx_colour.hip says why the recurrence
 * parallelises exactly).  PBDX_ERR_NO_DEVICE without a GPU, PBDX_ERR_UNSUPPORTED for what the device form does not take (more than
 * 128 groups, a constraint naming a body twice): the caller then uses the host form above -- there is no silent fallback. */
int pbdx_model_init_constraint_groups_device(pbdx_model *m, int device);
/* ... and on raw arrays, for a reference-side binding (what SimulationModel::initConstraintGroups, SimulationModel.cpp:1033-1094,
 * computes from m_constraints[i]->m_bodies): constraint i has bodies[body_off[i] .. body_off[i+1]) (1..4 of them, indices
 * < num_bodies = particles + rigid bodies).  group_of[i] receives its group; the reference's m_constraintGroups[g] is the ascending
 * list of the constraints with group_of == g.  rounds (optional): dependency rounds the propagation took. */
int pbdx_colour_constraints(int device, uint32_t num_bodies, uint32_t num_constraints, const uint32_t *body_off, const uint32_t *bodies,
	uint32_t *group_of, uint32_t *num_groups, uint32_t *rounds);
/* The same on the host (what pbdx_model_init_constraint_groups runs: one bit mask of used groups per body instead of the reference's
 * one byte map per group; identical groups, 0.05 s instead of 0.33 s for the 6 M constraints of configs[1]) -- the reference-side plug-in
 * colours with it when the model's groups are not initialised yet.  Any number of groups and of bodies per constraint. */
int pbdx_colour_constraints_host(uint32_t num_bodies, uint32_t num_constraints, const uint32_t *body_off, const uint32_t *bodies,
	uint32_t *group_of, uint32_t *num_groups);
int pbdx_model_groups_initialized(const pbdx_model *m);   /* m_groupsInitialized */
uint32_t pbdx_model_num_groups(const pbdx_model *m);
uint32_t pbdx_model_group_size(const pbdx_model *m, uint32_t g);
int pbdx_model_get_group(const pbdx_model *m, uint32_t g, uint32_t *out);

/* Planner self-test (host only, no GPU needed): packs the model's colour groups exactly as
 * pbdx_timestep does, plans the colour-fused tile schedule and proves by symbolic execution that
 * every particle receives the update history of the colour-sequential sweep.  Returns PBDX_OK or
 * PBDX_ERR_INVALID (pbdx_last_error() says why).  tile_particles / lds_particles / max_segment_colours
 * as in PBDX_OPT_*; 0 = defaults. */
int pbdx_model_plan_check(pbdx_model *m, uint32_t tile_particles, uint32_t lds_particles,
	uint32_t max_segment_colours, pbdx_plan_info *out);

/* ======================================================================== */
/* pbdx_timestep -- host mirror of PBD::TimeStepController                  */
/* ======================================================================== */
typedef struct pbdx_timestep pbdx_timestep;

/* Parameter ids; names mirror TimeStepController.cpp:47-72 so json scenes /
 * python setValueUInt keep their meaning.  Defaults subSteps=5, maxIterations=1,
 * maxIterationsV=5, velocityUpdateMethod=0 (TimeStepController.cpp:23-32). */
enum {
	PBDX_TS_NUM_SUB_STEPS = 0,          /* "subSteps" */
	PBDX_TS_MAX_ITERATIONS = 1,         /* "maxIterations" */
	PBDX_TS_MAX_ITERATIONS_V = 2,       /* "maxIterationsV" (no particle velocity constraints on this path) */
	PBDX_TS_VELOCITY_UPDATE_METHOD = 3  /* "velocityUpdateMethod" */
};

/* Creating a time step and setting its parameters needs no GPU; the engine on HIP device `device` is created
 * by the first call that steps / uploads (PBDX_ERR_NO_DEVICE there if none is visible: there is no CPU path). */
int pbdx_timestep_create(pbdx_timestep **out, int device);
void pbdx_timestep_destroy(pbdx_timestep *ts);
int pbdx_timestep_set_param(pbdx_timestep *ts, int id, int64_t value);
int64_t pbdx_timestep_get_param(const pbdx_timestep *ts, int id);
const char *pbdx_timestep_param_name(int id);
int pbdx_timestep_set_gravity(pbdx_timestep *ts, const float g[3]);     /* Simulation::GRAVITATION, Simulation.cpp:16 */
int pbdx_timestep_set_time_step_size(pbdx_timestep *ts, float h);       /* TimeManager::setTimeStepSize, default 0.005 */
float pbdx_timestep_get_time_step_size(const pbdx_timestep *ts);
float pbdx_timestep_get_time(const pbdx_timestep *ts);                  /* TimeManager::getTime */
int pbdx_timestep_reset(pbdx_timestep *ts);                             /* TimeStepController::reset + time=0 */

/* TimeStep::step(model) semantics: host ParticleData in -> one full step on
 * the device -> host ParticleData out (x, v, oldX, lastX, a).  The device image
 * (particles + packed colour schedule) is rebuilt when the model's topology
 * changed (groups re-initialised, particle/constraint count changed) or after
 * pbdx_timestep_invalidate(). */
int pbdx_timestep_step(pbdx_timestep *ts, pbdx_model *m);
/* Device-resident variant: uploads only if the image is stale, runs
 * `num_steps` steps without touching host memory; pbdx_timestep_sync_to_host
 * writes the state back into the model. */
int pbdx_timestep_step_resident(pbdx_timestep *ts, pbdx_model *m, uint32_t num_steps);
int pbdx_timestep_sync_to_host(pbdx_timestep *ts, pbdx_model *m);
/* explicit upload of the model's host particle state (and of a stale schedule) without stepping */
int pbdx_timestep_sync_from_host(pbdx_timestep *ts, pbdx_model *m);
int pbdx_timestep_invalidate(pbdx_timestep *ts);
/* Known-answer / teacher-forced entry: upload the model's host state, run only the projection
 * loop of one substep (`iterations` Gauss-Seidel sweeps over the colour groups, lambda reset at
 * sweep 0, XPBD dt = h/subSteps), download positions.  No integration, no velocity update. */
int pbdx_timestep_project(pbdx_timestep *ts, pbdx_model *m, uint32_t iterations);
/* the engine underneath (owned by the timestep; created on demand, NULL without a HIP device) */
pbdx_solver *pbdx_timestep_solver(pbdx_timestep *ts);

/* ======================================================================== */
/* pbdx_ensemble -- independent instances over several devices, one process  */
/* ======================================================================== */
/* a single scene does not shard (one connected colour-sequential Gauss-Seidel problem; the reference itself is one process on one model,
 * Simulation/TimeStepController.cpp:75-241); a model of K congruent independent instances (pbdx_model_add_instances) does -- contiguous blocks of
 * instances (pbdx_ensemble_shard), one per device, no exchange on the data path.  One engine (time step, solver, stream, device image) per entry of
 * `devices`; a device may be listed more than once (its blocks then share it).  A step runs every device's block concurrently (one RESIDENT host thread
 * per device, created with the ensemble and woken per call; every entry point selects its device and restores the caller's).  A single process needs no collective: what RCCL
 * reduces between the ranks of `bench.py --gpus N` is available on the host here. */
typedef struct pbdx_ensemble pbdx_ensemble;
int pbdx_ensemble_create(pbdx_ensemble **out, const int *devices, uint32_t n);
void pbdx_ensemble_destroy(pbdx_ensemble *e);
uint32_t pbdx_ensemble_num_shards(const pbdx_ensemble *e);
/* parameters of every engine (ids, names and defaults of pbdx_timestep_set_param / TimeStepController.cpp:47-72) */
int pbdx_ensemble_set_param(pbdx_ensemble *e, int id, int64_t value);
int pbdx_ensemble_set_gravity(pbdx_ensemble *e, const float g[3]);
int pbdx_ensemble_set_time_step_size(pbdx_ensemble *e, float h);
/* Splits `m` into one block of instances per device (fewer instances than devices: the surplus devices stay idle; a model without instances is one
 * block).  The blocks are COPIES: `m` is not referenced after the call (it may be destroyed); after editing `m` call this again -- pbdx_ensemble_gather
 * refuses a model that was edited since.  Every record of a block is bit for bit the whole model's.  A failure leaves the ensemble without a model. */
int pbdx_ensemble_set_model(pbdx_ensemble *e, const pbdx_model *m);
/* `num_steps` steps of every block, device-resident, all devices at once; returns when all are done. */
int pbdx_ensemble_step(pbdx_ensemble *e, uint32_t num_steps);
/* The blocks' particle state (x, v, oldX, lastX) into the arrays of the model given to pbdx_ensemble_set_model. */
int pbdx_ensemble_gather(pbdx_ensemble *e, pbdx_model *m);
/* Block `shard`: its device, its instances [*begin, *end) and the host wall time of its last step call; the engine and the block model behind it
 * (plan / schedule / timing queries: pbdx_timestep_solver + pbdx_solver_get_*); wall time of the last pbdx_ensemble_step over all devices. */
int pbdx_ensemble_get_shard(const pbdx_ensemble *e, uint32_t shard, int *device, uint64_t *begin, uint64_t *end, double *last_step_ms);
pbdx_timestep *pbdx_ensemble_timestep(pbdx_ensemble *e, uint32_t shard);
pbdx_model *pbdx_ensemble_shard_model(pbdx_ensemble *e, uint32_t shard);
double pbdx_ensemble_last_step_ms(const pbdx_ensemble *e);

/* ======================================================================== */
/* pbdx_comm -- the collective of a multi-process host: RCCL, loaded at run time */
/* ======================================================================== */
/* with one PROCESS per GPU (the form `bench.py --gpus N` runs under torch.distributed) the ranks exchange control data only -- a barrier,
 * the maximum of their times, the sum of their projection counts, their checksums.  A C / C++ host has no torch: these entry points give it the same
 * operations on the same library.  librccl.so is opened with dlopen by the first call (libpbdx.so does not link it; PBDX_RCCL_LIB names another file);
 * rank 0 calls pbdx_comm_unique_id and hands the PBDX_COMM_ID_BYTES bytes to the other ranks by the host's own channel (a file, MPI, a socket); every rank
 * calls pbdx_comm_create with its HIP device.  Values cross by value: host arrays in, host arrays out.  No reference counterpart. */
#define PBDX_COMM_ID_BYTES 128
typedef struct pbdx_comm pbdx_comm;
int pbdx_comm_available(void);                                   /* 1: librccl.so and the symbols used here were found */
int pbdx_comm_unique_id(void *id, size_t bytes);                 /* bytes >= PBDX_COMM_ID_BYTES */
int pbdx_comm_create(pbdx_comm **out, const void *id, size_t bytes, int world, int rank, int device);
void pbdx_comm_destroy(pbdx_comm *c);
int pbdx_comm_world(const pbdx_comm *c);
int pbdx_comm_rank(const pbdx_comm *c);
int pbdx_comm_all_reduce_sum_u64(pbdx_comm *c, uint64_t *values, uint32_t n);      /* in place */
int pbdx_comm_all_reduce_max_f64(pbdx_comm *c, double *values, uint32_t n);        /* in place */
int pbdx_comm_all_gather_u64(pbdx_comm *c, const uint64_t *mine, uint32_t n, uint64_t *all);   /* all: world x n, rank-major */
int pbdx_comm_barrier(pbdx_comm *c);

#ifdef __cplusplus
}
#endif
#endif /* PBDX_H */
