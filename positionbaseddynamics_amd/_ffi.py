"""ctypes binding of libpbdx.so (the C ABI declared in include/pbdx.h).

The library is built in-tree by `__graft_entry__.build()` /
`make -C positionbaseddynamics_amd/csrc`.  There is deliberately no Python or
CPU fallback: if the shared library is missing, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PBDX_LIB: developer override to A/B a differently tuned build of the SAME library (never a fallback)
LIB_PATH = os.environ.get("PBDX_LIB") or os.path.join(_HERE, "_lib", "libpbdx.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libpbdx.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C positionbaseddynamics_amd/csrc` (hipcc --offload-arch=gfx950)" % LIB_PATH)

lib = C.CDLL(LIB_PATH)

u32 = C.c_uint32
i64 = C.c_int64
f32 = C.c_float
vp = C.c_void_p
pf = C.POINTER(C.c_float)
pu = C.POINTER(C.c_uint32)
pd_ = C.POINTER(C.c_double)


class StepStats(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("projection_ms", C.c_double), ("projection_launches", C.c_uint64),
                ("projections", C.c_uint64), ("kernel_launches", C.c_uint64), ("algorithmic_bytes", C.c_uint64)]


class Collider(C.Structure):
    _fields_ = [("shape", C.c_int), ("invert", C.c_int), ("params", C.c_float * 4), ("com", C.c_float * 3), ("R", C.c_float * 9),
                ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("restitution", C.c_float), ("friction", C.c_float),
                ("body_v", C.c_float * 3), ("body_omega", C.c_float * 3), ("body_index", C.c_uint32)]


class ColliderDynamics(C.Structure):
    _fields_ = [("inv_mass", C.c_float), ("inertia_inv_w", C.c_float * 9), ("object_index", C.c_uint32), ("pad", C.c_uint32)]


class CollisionRange(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32), ("restitution", C.c_float), ("friction", C.c_float)]


class Bvh(C.Structure):
    _fields_ = [("num_nodes", C.c_uint32), ("num_entities", C.c_uint32), ("entities", C.POINTER(C.c_uint32)),
                ("nodes", C.POINTER(C.c_int32)), ("hulls", C.POINTER(C.c_float))]


class TetCollider(C.Structure):
    _fields_ = [("shape", C.c_int), ("invert", C.c_int), ("params", C.c_float * 4),
                ("first_particle", C.c_uint32), ("num_vertices", C.c_uint32), ("num_tets", C.c_uint32), ("tets", C.POINTER(C.c_uint32)),
                ("initial_x", C.c_float * 3), ("initial_R", C.c_float * 9), ("restitution", C.c_float), ("friction", C.c_float),
                ("test_mesh", C.c_int), ("body_index", C.c_uint32), ("points", Bvh), ("tets_bvh", Bvh), ("tets_rest", Bvh)]


TET_CONTACT_FLOATS = 34


class PlanInfo(C.Structure):
    _fields_ = [("built", C.c_int), ("active", C.c_int), ("num_segments", C.c_uint32), ("num_tiles", C.c_uint32),
                ("num_colours", C.c_uint32), ("max_local", C.c_uint32), ("slots_per_sweep", C.c_uint64),
                ("stream_bytes_per_sweep", C.c_uint64), ("redundancy", C.c_double), ("build_seconds", C.c_double),
                ("compulsory_stream_bytes_per_sweep", C.c_uint64)]


class SegmentInfo(C.Structure):
    _fields_ = [("colour_begin", C.c_uint32), ("colour_end", C.c_uint32), ("num_tiles", C.c_uint32), ("block", C.c_uint32),
                ("lds_bytes", C.c_uint32), ("type_mask", C.c_uint32), ("constraints", C.c_uint64), ("slots", C.c_uint64),
                ("stream_bytes", C.c_uint64), ("algorithmic_bytes", C.c_uint64), ("profiled_ms", C.c_double),
                ("profiled_launches", C.c_uint64)]


class PersistentInfo(C.Structure):
    _fields_ = [("eligible", C.c_int), ("active", C.c_int), ("grid", C.c_uint32), ("block", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("refusals", C.c_uint32), ("timeouts", C.c_uint32), ("last_folded", C.c_int), ("autotune_fused_ms", C.c_double), ("autotune_persistent_ms", C.c_double),
                ("profiled_ms", C.c_double), ("profiled_launches", C.c_uint64), ("algorithmic_bytes_per_sweep", C.c_uint64)]


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# every symbol include/pbdx.h declares: (name, restype, argtypes...)
SIGNATURES = [
    ("pbdx_type_num_bodies", u32, C.c_int), ("pbdx_type_param_stride", u32, C.c_int), ("pbdx_type_name", C.c_char_p, C.c_int),
    ("pbdx_type_algorithmic_bytes", u32, C.c_int),
    ("pbdx_last_error", C.c_char_p), ("pbdx_version", C.c_int), ("pbdx_device_count", C.c_int),
    ("pbdx_ensemble_shard", C.c_int, C.c_uint64, u32, u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)),
    ("pbdx_solver_create", C.c_int, C.POINTER(vp), C.c_int), ("pbdx_solver_destroy", None, vp),
    ("pbdx_solver_set_particles", C.c_int, vp, u32, pf, pf, pf, pf, pf, pf),
    ("pbdx_solver_set_particles_f64", C.c_int, vp, u32, pd_, pd_, pd_, pd_, pd_, pd_),
    ("pbdx_solver_get_particles_f64", C.c_int, vp, u32, pd_, pd_, pd_, pd_),
    ("pbdx_solver_update_batch_params", C.c_int, vp, u32, u32, pf, u32), ("pbdx_solver_commit_params", C.c_int, vp),
    ("pbdx_solver_set_positions", C.c_int, vp, u32, pf),
    ("pbdx_solver_get_particles", C.c_int, vp, u32, pf, pf, pf, pf),
    ("pbdx_solver_begin_schedule", C.c_int, vp),
    ("pbdx_solver_add_batch", C.c_int, vp, u32, C.c_int, u32, pu, pf, u32),
    ("pbdx_solver_set_instancing", C.c_int, vp, u32, u32),
    ("pbdx_solver_end_schedule", C.c_int, vp), ("pbdx_solver_validate_schedule", C.c_int, vp),
    ("pbdx_solver_step", C.c_int, vp, f32, u32, u32, C.c_int, pf, u32),
    ("pbdx_solver_project", C.c_int, vp, f32, u32), ("pbdx_solver_synchronize", C.c_int, vp),
    ("pbdx_solver_save_state", C.c_int, vp), ("pbdx_solver_restore_state", C.c_int, vp),
    ("pbdx_solver_integrate", C.c_int, vp, f32, pf), ("pbdx_solver_project_groups", C.c_int, vp, f32, u32, u32, u32),
    ("pbdx_solver_update_velocities", C.c_int, vp, f32, C.c_int),
    ("pbdx_solver_get_lambdas", C.c_int, vp, u32, u32, pf),
    ("pbdx_solver_set_option", C.c_int, vp, C.c_int, i64),
    ("pbdx_solver_get_stats", C.c_int, vp, C.POINTER(StepStats)),
    ("pbdx_solver_get_substep_times", C.c_int, vp, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)),
    ("pbdx_solver_set_profiling", C.c_int, vp, C.c_int),
    ("pbdx_solver_get_type_stats", C.c_int, vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)),
    ("pbdx_solver_describe", C.c_int, vp, C.c_char_p, C.c_size_t),
    ("pbdx_solver_get_plan_info", C.c_int, vp, C.POINTER(PlanInfo)),
    ("pbdx_solver_get_segment_info", C.c_int, vp, u32, C.POINTER(SegmentInfo)),
    ("pbdx_solver_get_persistent_info", C.c_int, vp, C.POINTER(PersistentInfo)),
    ("pbdx_solver_get_trace", C.c_int, vp, u32, C.POINTER(C.c_uint64), u32, C.POINTER(u32)),
    ("pbdx_solver_set_colliders", C.c_int, vp, u32, C.POINTER(Collider)),
    ("pbdx_solver_set_collision_ranges", C.c_int, vp, u32, C.POINTER(CollisionRange)),
    ("pbdx_solver_set_contact_params", C.c_int, vp, f32, f32, u32),
    ("pbdx_solver_get_num_contacts", C.c_int, vp, C.POINTER(u32)),
    ("pbdx_solver_set_collider_dynamics", C.c_int, vp, u32, C.POINTER(ColliderDynamics)),
    ("pbdx_solver_set_contact_order", C.c_int, vp, u32, C.POINTER(u32), u32, C.POINTER(u32)),
    ("pbdx_solver_get_body_velocities", C.c_int, vp, u32, pf, pf),
    ("pbdx_debug_stream", C.c_int, C.c_int, C.c_uint64, C.c_int),
    ("pbdx_debug_bounds_report", C.c_int, C.c_int, C.POINTER(u32), C.c_int),
    ("pbdx_debug_copy_bandwidth", C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)),
    ("pbdx_debug_host_copy", None, C.c_void_p, C.c_void_p, C.c_uint64),
    ("pbdx_debug_valu_issue", C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)),
    ("pbdx_debug_plan_lds_model", C.c_int, vp, C.c_int, C.c_int, C.POINTER(C.c_uint64)),
    ("pbdx_debug_relayout_params", C.c_int, C.c_int, C.c_int, u32, C.c_int, pf, C.POINTER(u32)),
    ("pbdx_debug_param_float_index", C.c_uint64, C.c_int, u32, u32, u32),
    ("pbdx_solver_get_particles_hashed", C.c_int, vp, u32, pf, pf, pf, pf, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)),
    ("pbdx_solver_get_particles_hashed_f64", C.c_int, vp, u32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)),
    ("pbdx_solver_update_particle_ranges", C.c_int, vp, C.c_int, pf, u32, pu),
    ("pbdx_solver_update_particle_ranges_f64", C.c_int, vp, C.c_int, C.POINTER(C.c_double), u32, pu),
    ("pbdx_solver_set_tet_colliders", C.c_int, vp, u32, C.POINTER(TetCollider), f32),
    ("pbdx_solver_set_rest_positions", C.c_int, vp, u32, pf),
    ("pbdx_solver_get_tet_contacts", C.c_int, vp, u32, C.POINTER(u32), pf),
    ("pbdx_debug_tet_counters", C.c_int, vp, C.POINTER(u32)),
    ("pbdx_debug_tet_capacity", C.c_int, vp, C.POINTER(u32)),
    ("pbdx_debug_tet_hulls", C.c_int, vp, u32, C.c_int, u32, C.POINTER(u32), pf),
    ("pbdx_debug_tet_solve_host", C.c_int, u32, pf, u32, pf, C.POINTER(u32)),
    ("pbdx_debug_tet_contacts", C.c_int, u32, pf, pf, pf, u32, C.POINTER(TetCollider), f32, u32, C.POINTER(u32), pf),
    ("pbdx_debug_tet_velocity_kat", C.c_int, pf, pf),
    ("pbdx_debug_dyn_contact_kat", C.c_int, pf, pf),
    ("pbdx_debug_tet_impulses", C.c_int, vp, C.POINTER(u32), C.POINTER(C.c_uint64)),
    ("pbdx_model_plan_check", C.c_int, vp, u32, u32, u32, C.POINTER(PlanInfo)),
    ("pbdx_model_create", C.c_int, C.POINTER(vp)), ("pbdx_model_destroy", None, vp),
    ("pbdx_model_cleanup", C.c_int, vp), ("pbdx_model_reset", C.c_int, vp),
    ("pbdx_model_add_regular_triangle_model", C.c_int, vp, C.c_int, C.c_int, pf, pf, pf),
    ("pbdx_model_add_triangle_model", C.c_int, vp, u32, u32, pf, pu),
    ("pbdx_model_add_regular_tet_model", C.c_int, vp, C.c_int, C.c_int, C.c_int, pf, pf, pf),
    ("pbdx_model_add_tet_model", C.c_int, vp, u32, u32, pf, pu),
    ("pbdx_model_add_instances", C.c_int, vp, u32, pf), ("pbdx_model_num_instances", u32, vp),
    ("pbdx_model_num_triangle_models", u32, vp), ("pbdx_model_num_tet_models", u32, vp),
    ("pbdx_model_triangle_model_index_offset", u32, vp, u32), ("pbdx_model_tet_model_index_offset", u32, vp, u32),
    ("pbdx_model_triangle_model_num_edges", u32, vp, u32), ("pbdx_model_triangle_model_get_edges", C.c_int, vp, u32, pu),
    ("pbdx_model_triangle_model_num_vertices", u32, vp, u32), ("pbdx_model_triangle_model_num_faces", u32, vp, u32),
    ("pbdx_model_triangle_model_get_faces", C.c_int, vp, u32, pu),
    ("pbdx_model_tet_model_num_vertices", u32, vp, u32), ("pbdx_model_tet_model_num_tets", u32, vp, u32),
    ("pbdx_model_tet_model_get_tets", C.c_int, vp, u32, pu),
    ("pbdx_model_tet_model_num_edges", u32, vp, u32), ("pbdx_model_tet_model_get_edges", C.c_int, vp, u32, pu),
    ("pbdx_model_num_particles", u32, vp), ("pbdx_model_add_vertex", C.c_int, vp, pf),
    ("pbdx_model_set_mass", C.c_int, vp, u32, f32),
    ("pbdx_model_get_array", C.c_int, vp, C.c_int, pf), ("pbdx_model_set_array", C.c_int, vp, C.c_int, pf),
    ("pbdx_model_positions_ptr", pf, vp),
    ("pbdx_model_add_distance_constraint", C.c_int, vp, u32, u32, f32),
    ("pbdx_model_add_distance_constraint_xpbd", C.c_int, vp, u32, u32, f32),
    ("pbdx_model_add_dihedral_constraint", C.c_int, vp, u32, u32, u32, u32, f32),
    ("pbdx_model_add_isometric_bending_constraint", C.c_int, vp, u32, u32, u32, u32, f32),
    ("pbdx_model_add_isometric_bending_constraint_xpbd", C.c_int, vp, u32, u32, u32, u32, f32),
    ("pbdx_model_add_fem_triangle_constraint", C.c_int, vp, u32, u32, u32, f32, f32, f32, f32, f32),
    ("pbdx_model_add_strain_triangle_constraint", C.c_int, vp, u32, u32, u32, f32, f32, f32, C.c_int, C.c_int),
    ("pbdx_model_add_volume_constraint", C.c_int, vp, u32, u32, u32, u32, f32),
    ("pbdx_model_add_volume_constraint_xpbd", C.c_int, vp, u32, u32, u32, u32, f32),
    ("pbdx_model_add_fem_tet_constraint", C.c_int, vp, u32, u32, u32, u32, f32, f32),
    ("pbdx_model_add_fem_tet_constraint_xpbd", C.c_int, vp, u32, u32, u32, u32, f32, f32),
    ("pbdx_model_add_strain_tet_constraint", C.c_int, vp, u32, u32, u32, u32, f32, f32, C.c_int, C.c_int),
    ("pbdx_model_add_shape_matching_constraint", C.c_int, vp, u32, pu, pu, f32),
    ("pbdx_model_add_cloth_constraints", C.c_int, vp, u32, u32, f32, f32, f32, f32, f32, f32, C.c_int, C.c_int),
    ("pbdx_model_add_bending_constraints", C.c_int, vp, u32, u32, f32),
    ("pbdx_model_add_solid_constraints", C.c_int, vp, u32, u32, f32, f32, f32, C.c_int, C.c_int),
    ("pbdx_model_num_constraints", u32, vp), ("pbdx_model_constraint_type", C.c_int, vp, u32),
    ("pbdx_model_constraint_bodies", C.c_int, vp, u32, pu), ("pbdx_model_constraint_params", C.c_int, vp, u32, pf),
    ("pbdx_model_set_constraint_params", C.c_int, vp, u32, pf),
    ("pbdx_model_init_constraint_groups", C.c_int, vp), ("pbdx_model_groups_initialized", C.c_int, vp),
    ("pbdx_model_init_constraint_groups_device", C.c_int, vp, C.c_int),
    ("pbdx_colour_constraints", C.c_int, C.c_int, u32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)),
    ("pbdx_colour_constraints_host", C.c_int, u32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)),
    ("pbdx_model_num_groups", u32, vp), ("pbdx_model_group_size", u32, vp, u32), ("pbdx_model_get_group", C.c_int, vp, u32, pu),
    ("pbdx_timestep_create", C.c_int, C.POINTER(vp), C.c_int), ("pbdx_timestep_destroy", None, vp),
    ("pbdx_timestep_set_param", C.c_int, vp, C.c_int, i64), ("pbdx_timestep_get_param", i64, vp, C.c_int),
    ("pbdx_timestep_param_name", C.c_char_p, C.c_int),
    ("pbdx_timestep_set_gravity", C.c_int, vp, pf), ("pbdx_timestep_set_time_step_size", C.c_int, vp, f32),
    ("pbdx_timestep_get_time_step_size", f32, vp), ("pbdx_timestep_get_time", f32, vp),
    ("pbdx_timestep_reset", C.c_int, vp), ("pbdx_timestep_step", C.c_int, vp, vp),
    ("pbdx_timestep_step_resident", C.c_int, vp, vp, u32), ("pbdx_timestep_sync_to_host", C.c_int, vp, vp),
    ("pbdx_timestep_sync_from_host", C.c_int, vp, vp), ("pbdx_model_mark_state_dirty", C.c_int, vp),
    ("pbdx_timestep_invalidate", C.c_int, vp), ("pbdx_timestep_project", C.c_int, vp, vp, u32), ("pbdx_timestep_solver", vp, vp),
    ("pbdx_ensemble_create", C.c_int, C.POINTER(vp), C.POINTER(C.c_int), u32), ("pbdx_ensemble_destroy", None, vp), ("pbdx_ensemble_num_shards", u32, vp),
    ("pbdx_ensemble_set_param", C.c_int, vp, C.c_int, i64), ("pbdx_ensemble_set_gravity", C.c_int, vp, pf), ("pbdx_ensemble_set_time_step_size", C.c_int, vp, f32),
    ("pbdx_ensemble_set_model", C.c_int, vp, vp), ("pbdx_ensemble_step", C.c_int, vp, u32), ("pbdx_ensemble_gather", C.c_int, vp, vp),
    ("pbdx_ensemble_get_shard", C.c_int, vp, u32, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)),
    ("pbdx_ensemble_timestep", vp, vp, u32), ("pbdx_ensemble_shard_model", vp, vp, u32), ("pbdx_ensemble_last_step_ms", C.c_double, vp),
    # the collective of a multi-process host (RCCL loaded at run time, csrc/pbdx_comm.cpp)
    ("pbdx_comm_available", C.c_int), ("pbdx_comm_unique_id", C.c_int, vp, C.c_size_t),
    ("pbdx_comm_create", C.c_int, C.POINTER(vp), vp, C.c_size_t, C.c_int, C.c_int, C.c_int), ("pbdx_comm_destroy", None, vp),
    ("pbdx_comm_world", C.c_int, vp), ("pbdx_comm_rank", C.c_int, vp),
    ("pbdx_comm_all_reduce_sum_u64", C.c_int, vp, C.POINTER(C.c_uint64), u32), ("pbdx_comm_all_reduce_max_f64", C.c_int, vp, C.POINTER(C.c_double), u32),
    ("pbdx_comm_all_gather_u64", C.c_int, vp, C.POINTER(C.c_uint64), u32, C.POINTER(C.c_uint64)), ("pbdx_comm_barrier", C.c_int, vp),
]

for _s in SIGNATURES:
    _sig(*_s)


class PbdxError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.pbdx_last_error().decode(errors="replace")
        super().__init__("%s failed with status %d: %s" % (where, code, msg))


def check(code, where):
    if code != 0:
        raise PbdxError(code, where)
    return code
