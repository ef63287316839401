"""Developer aid: wall time per kernel of the deformable-contact path on the two-bar scene (PBDX_TET_PROFILE=1, hipGraph off)."""
import sys, os, time, ctypes as C
os.environ["PBDX_TET_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refdrv
from tests import tetcontact_util as tcu, test_tetcontact as tt
from positionbaseddynamics_amd import _ffi
dims = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,16,16").split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ref = refdrv.Ref("f32")
tcu.two_bar_scene(ref, dims=dims, t_upper=(0.3, 0.5 * (1.0 + 1.0 / (dims[1] - 1)) + 0.02, 0.05))
assert ref.install_timestep_plugin(tt.PLUGIN) == 0
ref.lib.refdrv_attach_collision_detection()
ref.set_params(1, 5, 0)
lib, ts = tt._plugin_handles(ref)
sol = C.c_void_p(lib.pbdx_timestep_hip_solver(ts))
_ffi.check(_ffi.lib.pbdx_solver_set_option(sol, 1, 0), "graph off")
lib.pbdx_timestep_hip_step_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
model = ref.model_ptr()
assert lib.pbdx_timestep_hip_step_resident(ts, model, 1) == 0
t0 = time.perf_counter()
if os.environ.get("PBDX_TET_STEPWISE"):
    c = (C.c_uint32 * 8)()
    for k in range(steps - 1):
        rc = lib.pbdx_timestep_hip_step_resident(ts, model, 1)
        _ffi.check(_ffi.lib.pbdx_debug_tet_counters(sol, c), "counters")
        print("step %d rc %d: contacts %d flags %d %d leaf pairs %d chunks %d levels %d generations %d tree nodes %d" % (k + 2, rc, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]))
        if rc:
            break
else:
    assert lib.pbdx_timestep_hip_step_resident(ts, model, steps - 1) == 0
print("ms/step (profiling syncs included): %.3f" % ((time.perf_counter() - t0) * 1e3 / (steps - 1)))
c = (C.c_uint32 * 8)()
_ffi.check(_ffi.lib.pbdx_debug_tet_counters(sol, c), "counters")
print("last detection: contacts %d, leaf pairs %d, chunks %d, levels %d, generations %d, tree nodes %d" % (c[0], c[3], c[4], c[5], c[6], c[7]))
print("contacts at the end:", len(tt._device_contacts(lib, ts, 1 << 16)))
ref.reset_all()      # destroys the plug-in and its solver: the profile is printed
