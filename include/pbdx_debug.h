/* pbdx_debug.h -- developer / test aids of libpbdx.so.  NOT part of the drop-in boundary (include/pbdx.h): nothing a
 * reference-side binding needs is declared here.  These entry points evaluate pieces of the engine in isolation (the
 * contact detection and the contact solve on the host with the device's own header code, counters and capacities of the
 * last detection, counter-calibration streaming kernels for rocprofv3 --pmc) for the test-suite and the profiling scripts. */
#ifndef PBDX_DEBUG_H
#define PBDX_DEBUG_H

#include "pbdx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Developer aid: the position solve of contact records (as returned by pbdx_solver_get_tet_contacts) applied on the host to pos4 (n x (x, y, z, invMass),
 * in place) in the given order (order == NULL: list order): the sequential loop of TimeStepController.cpp:288-291 with the engine's arithmetic. */
int pbdx_debug_tet_solve_host(uint32_t n_particles, float *pos4, uint32_t n_contacts, const float *records, const uint32_t *order);
/* Developer aid: the bounding spheres (centre, radius) of hierarchy `which` (0 points, 1 tets, 2 tets at rest) of a collider as the last
 * detection left them; *count = number of nodes. */
int pbdx_debug_tet_hulls(pbdx_solver *s, uint32_t collider, int which, uint32_t capacity, uint32_t *count, float *out);
/* Developer aid: counters of the last detection: contacts, (flags: 1, 2), overlapping leaf pairs, 64-candidate chunks, dependency levels of the
 * solve, generations and node pairs of the traversal's recursion tree. */
int pbdx_debug_tet_counters(pbdx_solver *s, uint32_t out[8]);
/* Developer aid: capacities of the detection's scratch (node pairs, overlapping leaf pairs, contacts) and how often it was enlarged: the scratch
 * grows on demand -- an overflow is detected after the detection that caused it, the buffer is made four times as large and the detection repeated. */
int pbdx_debug_tet_capacity(pbdx_solver *s, uint32_t out[4]);
/* The same detection evaluated on the HOST by the same code (pbdx_tetcontact.h is host + device): developer / test aid, no GPU
 * needed.  pos4 / rest4: n x (x, y, z, invMass) records. */
int pbdx_debug_tet_contacts(uint32_t n_particles, const float *pos4, const float *rest4, const float *vel4 /* (vx, vy, vz, mass) records or NULL: at rest */,
	uint32_t n, const pbdx_tet_collider *colliders, float tolerance, uint32_t capacity, uint32_t *count, float *out);
/* Developer aid: the velocity part of ONE particle-tet contact on the host (the engine's arithmetic; friction 0).  in (26 floats): invMass0, v0[3],
 * invMass[4], v[4][3], bary[3], normal[3]; out (20 floats): tangent[3], pMax, impulse applied (1 / 0), corr_v0[3], corr_v[4][3]. */
int pbdx_debug_tet_velocity_kat(const float *in, float *out);
/* The contact of a particle with a rigid body of any mass on the host (csrc/pbdx_contact.h dyn_contact_*): in 38 floats, out 20 floats, see csrc/pbdx_tetcontact.cpp. */
int pbdx_debug_dyn_contact_kat(const float *in, float *out);
/* Developer aid: how many contacts of the last detection carried a non-zero velocity impulse (pMax < 0), and the total since the colliders were set. */
int pbdx_debug_tet_impulses(pbdx_solver *s, uint32_t *last, uint64_t *total);

/* Counter calibration (developer aid for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): one streaming
 * kernel over `nbytes` of HBM in one of the engine's access widths.  mode 0: 4-byte reads
 * (kernel calib_read_b32), 1: 16-byte reads (calib_read_b128), 2: 4-byte writes, 3: 16-byte writes, 4: the buffer read EIGHT times in
 * one launch with 16-byte loads (calib_reread_b128: with a buffer between the L2 and the Infinity-Cache size this tells whether a
 * counter sees Infinity-Cache hits). */
int pbdx_debug_stream(int device, uint64_t nbytes, int mode);
/* Developer / test aid (host only): the two forms of a step's parameter stream (pbdx_plan.h param_float_index: planes for 1 024-thread workgroups,
 * vector segments up to 512).  `block` holds ceil(slots / 64) * planes * 64 floats of ONE step in the form `from_vector`; it is rewritten into the
 * other form by the planner's own conversion (relayout_params).  `type` / `compact` select the constraint type whose plane count applies;
 * *planes_out (optional) receives that count. */
int pbdx_debug_relayout_params(int type, int compact, uint32_t slots, int from_vector, float *block, uint32_t *planes_out);
/* ... and the index function itself: float index of (plane, slot) relative to the step's first float. */
uint64_t pbdx_debug_param_float_index(int vector_params, uint32_t planes, uint32_t plane, uint32_t slot);

/* The practical HBM roof ("use the measured copy bandwidth as the practical roof and report both"): device-to-device float4 copy of `nbytes`,
 * best of `reps` launches, read + written bytes per second in GB/s. */
int pbdx_debug_copy_bandwidth(int device, uint64_t nbytes, int reps, double *gbs);
/* The library's host-side copy (csrc/pbdx_hostio.hip: memcpy by a resident team of threads above 1 MiB; the copy between caller memory and
 * the library's page-locked buffers).  Needs no GPU: tests/test_hostio.py, scripts/dev/hostio_bench.py. */
void pbdx_debug_host_copy(void *dst, const void *src, uint64_t bytes);
/* The vector-ALU issue interval the SQ_INSTS_VALU counter is to be priced with: shader cycles per wave64 v_mul_f32 / v_add_f32 and SIMD, measured with
 * `threads` (256, 512, 1024) threads per workgroup and one workgroup per CU, i.e. at the occupancy of the sweep kernels. */
int pbdx_debug_valu_issue(int device, int threads, double *cycles_per_instruction);
/* Developer aid (host only, no GPU): the colour-fused plan the engine builds for `m` on a 256-CU device, with (bank_aware != 0) or without the bank-aware
 * order of the slots inside every colour step, evaluated under the LDS bank model of csrc/pbdx_plan.h (lds_bank_model) and, if `check`, executed
 * symbolically against the colour-sequential sweep.  out[0..5] = 16-lane read groups of the endpoint gathers (= their cycles if conflict-free), modelled
 * read cycles, 8-lane write groups, modelled write cycles, groups and cycles of the dictionary-record reads -- all per sweep; out[6] = slots per sweep,
 * out[7] = plan build time (microseconds). */
int pbdx_debug_plan_lds_model(pbdx_model *m, int bank_aware, int check, uint64_t out[8]);

/* Sanitizer-grade debug build (libpbdx built with -DPBDX_BOUNDS=1, scripts/build_variant.sh): every raw address of the fused / persistent sweep is
 * range-checked before the access (csrc/pbdx_bounds.h).  out[0] = violations since the last reset, out[1..6] = the first one (kind, workgroup, thread,
 * index, limit, tile), out[7] = 1 if the loaded library is such a build (the product build checks nothing and reports zeros). */
int pbdx_debug_bounds_report(int device, uint32_t out[8], int reset);

#ifdef __cplusplus
}
#endif

#endif
