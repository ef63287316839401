import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import refdrv
from tests import tetcontact_util as tcu, test_tetcontact as tt
from positionbaseddynamics_amd import _ffi
ref = refdrv.Ref("f32")
objs = tcu.two_bar_scene(ref)
ref.set_params(1, 5, 0)
ref.step(60)
cpu = [[ref.bvh(co, which)["hulls"].astype(np.float32) for which in (0, 1)] for co in objs]
nodes = [[ref.bvh(co, which)["nodes"] for which in (0, 1)] for co in objs]
tcu.two_bar_scene(ref)
assert ref.install_timestep_plugin(tt.PLUGIN) == 0
ref.lib.refdrv_attach_collision_detection()
ref.set_params(1, 5, 0)
lib, ts = tt._plugin_handles(ref)
_ffi.check(_ffi.lib.pbdx_solver_set_option(C.c_void_p(lib.pbdx_timestep_hip_solver(ts)), 15, int(sys.argv[1]) if len(sys.argv) > 1 else 0), "opt")
ref.step(60)
for q in (0, 1):
    for which in (0, 1):
        d = tt._device_hulls(lib, ts, q, which)
        bad = np.where((d.view(np.uint32) != cpu[q][which].view(np.uint32)).any(axis=1))[0]
        print("solid", q, "hier", which, "nodes", len(d), "mismatching", len(bad))
        for i in bad[:6]:
            print("   node", i, nodes[q][which][i], d[i], cpu[q][which][i], (d[i] - cpu[q][which][i]))
