"""Scene specifications of the BASELINE workloads (and of the parity scenes), in the package so that bench.py
and the examples do not depend on the test tree.

A scene is a list of operations (the reference's own builder calls, Demos/ClothDemo/main.cpp:117-162,
Demos/BarDemo/main.cpp:116-166, with their arguments), so that the same list can drive this package's
SimulationModel mirror (`build_model`) and -- in the tests and in bench.py's cpu_baseline leg -- the reference
itself.  All scenes are deterministic (no RNG except the seeded irregular meshes)."""
import os

import numpy as np

GRAVITY = (0.0, -9.81, 0.0)


def rot_x_half_pi():
    """AngleAxisr(M_PI*0.5, (1,0,0)).matrix() in a float build (Demos/ClothDemo/main.cpp:136):
    cos(float(pi/2)) = -4.371139e-08, sin = 1."""
    a = np.float32(np.pi * 0.5)
    c = np.float32(np.cos(a))
    s = np.float32(np.sin(a))
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float32)


def cloth_spec(n_cols, n_rows, cloth_method, bending_method, cloth_k=None, bending_k=None,
               width=10.0, height=10.0, T=(0, 1, 0), pin=True, instances=1, instance_offset=(0, 0, 0), instanced=False, first_instance=0):
    """Demos/ClothDemo/main.cpp:132-162 generalised to n_cols x n_rows and K instances.  instanced=True: the K - 1 copies
    are one ("instances", offsets) operation (SimulationModel.addInstances: built, coloured and stored once) instead of
    K - 1 more rounds of builder calls; both describe the same model.
    first_instance = g0: the K instances are the GLOBAL instances g0 .. g0+K-1 of a larger ensemble (instance g sits at
    T + g * instance_offset, float arithmetic): what a rank of a sharded job builds (bench.py --workload c4)."""
    if first_instance:
        f = np.float32
        Tg = [tuple(f(T[i]) + f(g) * f(instance_offset[i]) for i in range(3)) for g in range(first_instance, first_instance + instances)]
        # the shard is built as instance g0 + copies at k * instance_offset: that must land every copy exactly where the
        # whole job's builder calls put instance g0 + k (true for offsets / translations exactly representable in float)
        for k in range(instances):
            if any(f(Tg[0][i]) + f(k) * f(instance_offset[i]) != Tg[k][i] for i in range(3)):
                raise ValueError("cloth_spec: instance %d of a shard starting at %d would not sit where the unsharded ensemble puts it" % (k, first_instance))
        return cloth_spec(n_cols, n_rows, cloth_method, bending_method, cloth_k, bending_k, width, height, Tg[0], pin, instances, instance_offset, instanced)
    if instanced and instances > 1:
        ops = cloth_spec(n_cols, n_rows, cloth_method, bending_method, cloth_k, bending_k, width, height, T, pin)
        offs = [tuple(np.float32(k) * np.float32(instance_offset[i]) for i in range(3)) for k in range(1, instances)]
        return ops + [("instances", offs)]
    if cloth_k is None:
        cloth_k = 100000.0 if cloth_method == 4 else 1.0
    if bending_k is None:
        bending_k = 100.0 if bending_method == 3 else 0.01
    ops = []
    R = rot_x_half_pi()
    for k in range(instances):
        off = k * n_cols * n_rows
        Tk = tuple(np.float32(T[i]) + np.float32(k) * np.float32(instance_offset[i]) for i in range(3))
        ops.append(("tri", n_cols, n_rows, Tk, R, (width, height)))
        if pin:
            ops.append(("mass", off, 0.0))
            ops.append(("mass", off + n_rows - 1, 0.0))
        if cloth_method:
            ops.append(("cloth", k, cloth_method, cloth_k, 1.0, 1.0, 1.0, 0.3, 0.3, False, False))
        if bending_method:
            ops.append(("bending", k, bending_method, bending_k))
    return ops


def bar_spec(width, height, depth, solid_method, k=None, kv=None, poisson=0.3, T=(5, 0, 0),
             scale=(10.0, 1.5, 1.5), ns=False, nsh=False, instances=1, instance_offset=(0.0, 0.0, 3.0), instanced=False):
    """Demos/BarDemo/main.cpp:130-166 generalised (and K independent bars for ensemble runs; instanced: see cloth_spec)."""
    if instanced and instances > 1:
        # (the plain K-instance form takes T as given for instance 0 and float32 arithmetic for the others: the same here)
        ops = bar_spec(width, height, depth, solid_method, k, kv, poisson, tuple(np.float32(t) for t in T), scale, ns, nsh)
        offs = [tuple(np.float32(q) * np.float32(instance_offset[i]) for i in range(3)) for q in range(1, instances)]
        return ops + [("instances", offs)]
    if k is None:
        k = {3: 1000000.0, 6: 100000.0}.get(solid_method, 1.0)
    if kv is None:
        kv = 100000.0 if solid_method == 6 else 1.0
    ops = []
    n_per = width * height * depth
    for inst in range(instances):
        Tk = tuple(np.float32(T[i]) + np.float32(inst) * np.float32(instance_offset[i]) for i in range(3)) if instances > 1 else T
        ops.append(("tet", width, height, depth, Tk, None, scale))
        for j in range(height):
            for q in range(depth):
                ops.append(("mass", inst * n_per + j * depth + q, 0.0))
        ops.append(("solid", inst, solid_method, k, poisson, kv, ns, nsh))
    return ops


def delaunay_cloth_spec(n_points=900, seed=3, cloth_method=4, bending_method=3):
    """An IRREGULAR triangle mesh (2-D Delaunay triangulation of random points, lifted to a wavy sheet):
    vertex valences 3..12, no grid structure, boundary edges.  Deterministic."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    uv = rng.random((n_points, 2)) * 8.0
    tri = Delaunay(uv)
    pts = np.stack([uv[:, 0], 0.3 * np.sin(uv[:, 0]) * np.cos(uv[:, 1]) + 1.0, uv[:, 1]], axis=1).astype(np.float32)
    faces = tri.simplices.astype(np.uint32)
    ops = [("trimesh", pts, faces), ("mass", 0, 0.0), ("mass", 1, 0.0)]
    ops.append(("cloth", 0, cloth_method, 100000.0 if cloth_method == 4 else 1.0, 1.0, 1.0, 1.0, 0.3, 0.3, False, False))
    if bending_method:
        ops.append(("bending", 0, bending_method, 100.0 if bending_method == 3 else 0.01))
    return ops


def delaunay_solid_spec(n_points=400, seed=5, solid_method=2):
    """An irregular tetrahedral mesh (3-D Delaunay of random points in a box), slivers removed."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = (rng.random((n_points, 3)) * np.array([4.0, 1.5, 1.5])).astype(np.float32)
    tets = Delaunay(pts.astype(np.float64)).simplices
    p = pts.astype(np.float64)
    vol = np.abs(np.einsum("ij,ij->i", p[tets[:, 3]] - p[tets[:, 0]], np.cross(p[tets[:, 2]] - p[tets[:, 0]], p[tets[:, 1]] - p[tets[:, 0]]))) / 6.0
    tets = tets[vol > 1e-3].astype(np.uint32)
    k = {3: 1000000.0, 6: 100000.0}.get(solid_method, 1.0)
    ops = [("tetmesh", pts, tets)]
    for i in np.nonzero(pts[:, 0] < 0.3)[0]:
        ops.append(("mass", int(i), 0.0))
    ops.append(("solid", 0, solid_method, k, 0.3, 100000.0 if solid_method == 6 else 1.0, False, False))
    return ops


def config5_like_spec(n_points=220):
    """A small stand-in for BASELINE configs[4] with floor contacts only: three irregular tet solids (FEM tets, method 2, Poisson 0.2) stacked above
    a static floor, submitted as ONE tet model with three components (so that no tet-tet pairs exist).  The scene file's own shape -- three
    armadillo_4k tet models, floor contacts AND contacts between the solids -- is `armadillo_collision_scene` below."""
    from scipy.spatial import Delaunay
    all_pts, all_tets = [], []
    offset = 0
    for k, (ty, seed) in enumerate(((1.6, 11), (3.4, 12), (5.2, 13))):
        rng = np.random.default_rng(seed)
        pts = (rng.random((n_points, 3)) * np.array([1.6, 1.2, 1.4]) + np.array([-0.8 + 0.3 * k, ty, -0.7])).astype(np.float32)
        tets = Delaunay(pts.astype(np.float64)).simplices
        p = pts.astype(np.float64)
        vol = np.abs(np.einsum("ij,ij->i", p[tets[:, 3]] - p[tets[:, 0]], np.cross(p[tets[:, 2]] - p[tets[:, 0]], p[tets[:, 1]] - p[tets[:, 0]]))) / 6.0
        all_pts.append(pts)
        all_tets.append(tets[vol > 2e-4].astype(np.uint32) + np.uint32(offset))
        offset += n_points
    return [("tetmesh", np.concatenate(all_pts), np.concatenate(all_tets)), ("solid", 0, 2, 1.0, 0.2, 1.0, False, False)]


# ---- BASELINE configs[4]: data/scenes/ArmadilloCollisionScene.json in the shape that can be pinned -----------------------------------------
ARMADILLO_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "armadillo_collision_scene.npz")


def armadillo_collision_scene():
    """(ops, fixture).  The scene file's three armadillo_4k tet models (1180 vertices, 3717 tets each; scale 2; translations (0,10,0), (0,4,0),
    (0,18,0); rotations 1.57 about y, 0.1 and 0.5 about z) as the reference's own TetGenLoader + SceneLoaderDemo placement produce them, FEM tets
    (tetModelSimulationMethod 2, stiffness 1, Poisson ratio 0.2), above the static floor box (100 x 1 x 100, restitution 0.6, friction 0):
    maxIterations 1, maxIterationsV 5, time step 0.01, contact tolerance 0, contact stiffness 100; subSteps 5 in the file, 8 in BASELINE.json.
    Two forced deviations: the solids carry an analytic box (3 x 4 x 2.6) in their rest frame instead of the cubic SDF of armadillo.obj (Discregrid
    and that surface mesh are not in the reference tree), and their friction coefficient is 0 instead of 0.3 (the reference's friction impulse between
    deformables reads a multiplier nothing has written).  The fixture (tests/golden/make_golden.py armadillo) holds the placed meshes, the
    bounding-sphere hierarchies the reference built, the collision objects as the reference stores them, and the reference's results."""
    g = np.load(ARMADILLO_FIXTURE)
    ops = []
    for q in range(3):
        _, _, offset, nv, nt, _ = (int(v) for v in g["c%d_meta" % q])
        ops.append(("tetmesh", g["x0"][offset:offset + nv], g["c%d_tets" % q].reshape(-1, 4)))
    method, k, nu, kv = g["solid"]
    for q in range(3):
        ops.append(("solid", q, int(method), float(k), float(nu), float(kv), False, False))
    return ops, g


class FixtureTetColliders:
    """pbdx_tet_collider records from a fixture's arrays (keys c<q>_meta / _params / _tets / _initial_x / _initial_R / _<hierarchy>_{lst,nodes,hulls});
    keeps the arrays alive."""

    def __init__(self, g, n):
        import ctypes as C
        from . import _ffi
        self.keep = []
        self.n = n
        self.arr = (_ffi.TetCollider * n)()
        self.tolerance = float(g["tolerance"])
        for q in range(n):
            shape, invert, offset, nv, nt, body = (int(v) for v in g["c%d_meta" % q])
            c = self.arr[q]
            c.shape, c.invert = shape, invert
            for k in range(4):
                c.params[k] = g["c%d_params" % q][k]
            c.first_particle, c.num_vertices, c.num_tets = offset, nv, nt
            tets = np.ascontiguousarray(g["c%d_tets" % q], dtype=np.uint32)
            self.keep.append(tets)
            c.tets = tets.ctypes.data_as(C.POINTER(C.c_uint32))
            for k in range(3):
                c.initial_x[k] = g["c%d_initial_x" % q][k]
            for k in range(9):
                c.initial_R[k] = g["c%d_initial_R" % q].reshape(-1)[k]
            c.restitution = float(g["c%d_restitution" % q]) if ("c%d_restitution" % q) in g else 0.6
            c.friction, c.test_mesh, c.body_index = 0.0, 1, body
            for name, field in (("points", "points"), ("tets", "tets_bvh"), ("rest", "tets_rest")):
                lst = np.ascontiguousarray(g["c%d_%s_lst" % (q, name)], dtype=np.uint32)
                nodes = np.ascontiguousarray(g["c%d_%s_nodes" % (q, name)], dtype=np.int32)
                hulls = np.ascontiguousarray(g["c%d_%s_hulls" % (q, name)], dtype=np.float32)
                self.keep += [lst, nodes, hulls]
                f = getattr(c, field)
                f.num_nodes, f.num_entities = len(nodes), len(lst)
                f.entities = lst.ctypes.data_as(C.POINTER(C.c_uint32))
                f.nodes = nodes.ctypes.data_as(C.POINTER(C.c_int32))
                f.hulls = hulls.ctypes.data_as(C.POINTER(C.c_float))


def install_armadillo_colliders(sol, g):
    """floor (static rigid body with an analytic box), the three solids' collision ranges, contact parameters, rest positions and the deformable
    colliders of the armadillo scene on a Solver whose particles are uploaded.  Returns the collider records (keep them alive while they are in use)."""
    sol.set_colliders([dict(shape=int(g["rb_shape"]), invert=bool(g["rb_invert"]), params=list(g["rb_params"]), com=g["rb_com"], R=g["rb_R"], v1=g["rb_v1"], v2=g["rb_v2"],
                            restitution=float(g["rb_restitution"]), friction=float(g["rb_friction"]), body_index=int(g["rb_body_index"]))])
    sol.set_collision_ranges([(int(r[0]), int(r[1]), float(r[2]), float(r[3])) for r in g["ranges"]])
    sol.set_contact_params(float(g["tolerance"]), float(g["contact_stiffness"]), int(g["max_iterations_v"]))
    sol.set_rest_positions(g["x0"])
    cols = FixtureTetColliders(g, 3)
    sol.set_tet_colliders(cols.arr, cols.n, cols.tolerance)
    return cols


def kitchen_sink_spec():
    """One model with EVERY particle constraint type, so that colour groups mix types (a colour becomes
    several (colour, type) batches / tile steps): a cloth with FEM-triangle + distance + dihedral + PBD
    isometric bending, a second cloth with strain-triangle + XPBD distance + XPBD bending, a tet bar with
    FEM + volume + distance, a second bar with XPBD FEM + XPBD volume + strain + shape matching, plus
    extra XPBD distance constraints stitching the two cloths together."""
    ops = []
    R = rot_x_half_pi()
    ops.append(("tri", 14, 12, (0.0, 2.0, 0.0), R, (4.0, 3.0)))
    ops.append(("tri", 10, 11, (0.5, 2.6, 0.2), R, (3.0, 3.0)))
    ops.append(("tet", 7, 4, 3, (5.0, 0.0, 0.0), None, (3.0, 1.0, 0.8)))
    ops.append(("tet", 5, 3, 3, (5.0, 2.0, 0.0), None, (2.0, 0.8, 0.8)))
    n0, n1 = 14 * 12, 10 * 11
    ops += [("mass", 0, 0.0), ("mass", 13, 0.0), ("mass", n0, 0.0)]
    for j in range(4 * 3):
        ops.append(("mass", n0 + n1 + j, 0.0))
    ops.append(("cloth", 0, 2, 1.0, 1.0, 1.0, 1.0, 0.3, 0.3, False, False))      # FEM triangle
    ops.append(("cloth", 0, 1, 0.8, 1.0, 1.0, 1.0, 0.3, 0.3, False, False))      # distance
    ops.append(("bending", 0, 1, 0.02))                                          # dihedral
    ops.append(("bending", 0, 2, 0.01))                                          # isometric bending
    ops.append(("cloth", 1, 3, 1.0, 1.0, 1.0, 1.0, 0.3, 0.3, True, False))       # strain triangle
    ops.append(("cloth", 1, 4, 50000.0, 1.0, 1.0, 1.0, 0.3, 0.3, False, False))  # XPBD distance
    ops.append(("bending", 1, 3, 50.0))                                          # XPBD isometric bending
    ops.append(("solid", 0, 2, 1.0, 0.3, 1.0, False, False))                     # FEM tet
    ops.append(("solid", 0, 1, 0.9, 0.3, 0.7, False, False))                     # distance + volume
    ops.append(("solid", 1, 3, 100000.0, 0.3, 1.0, False, False))                # XPBD FEM tet
    ops.append(("solid", 1, 6, 50000.0, 0.3, 50000.0, False, False))             # XPBD distance + volume
    ops.append(("solid", 1, 4, 1.0, 0.3, 1.0, False, True))                      # strain tet
    ops.append(("solid", 1, 5, 0.5, 0.3, 1.0, False, False))                     # shape matching
    for k in range(8):
        ops.append(("constraint", "distance_xpbd", [20 + 14 * (k % 3) + k, n0 + 10 * (k % 4) + k], 2000.0))
    return ops



_CONSTRAINT_ADD = {
    "distance": "addDistanceConstraint", "distance_xpbd": "addDistanceConstraint_XPBD",
    "dihedral": "addDihedralConstraint", "isometric_bending": "addIsometricBendingConstraint",
    "isometric_bending_xpbd": "addIsometricBendingConstraint_XPBD", "fem_triangle": "addFEMTriangleConstraint",
    "strain_triangle": "addStrainTriangleConstraint", "volume": "addVolumeConstraint",
    "volume_xpbd": "addVolumeConstraint_XPBD", "fem_tet": "addFEMTetConstraint", "fem_tet_xpbd": "addFEMTetConstraint_XPBD",
    "strain_tet": "addStrainTetConstraint",
}


def build_model(ops):
    """Build the same scene with the product's SimulationModel mirror."""
    import positionbaseddynamics_amd as pbd  # noqa: PLC0415 (the package imports this module lazily)
    m = pbd.SimulationModel()
    for op in ops:
        k = op[0]
        if k == "tri":
            m.addRegularTriangleModel(op[1], op[2], op[3], op[4], op[5])
        elif k == "tet":
            m.addRegularTetModel(op[1], op[2], op[3], op[4], op[5], op[6])
        elif k == "trimesh":
            m.addTriangleModel(op[1], op[2])
        elif k == "tetmesh":
            m.addTetModel(op[1], op[2])
        elif k == "vertex":
            m.getParticles().addVertex(op[1])
        elif k == "mass":
            m.getParticles().setMass(op[1], op[2])
        elif k == "cloth":
            m.addClothConstraints(*op[1:])
        elif k == "bending":
            m.addBendingConstraints(*op[1:])
        elif k == "solid":
            m.addSolidConstraints(*op[1:])
        elif k == "instances":
            m.addInstances(op[1])
        elif k == "constraint":
            if op[1] == "shape_matching":
                ok = m.addShapeMatchingConstraint(len(op[2]), op[2], op[3], op[4])
            else:
                ok = getattr(m, _CONSTRAINT_ADD[op[1]])(*[int(b) for b in op[2]], *op[3:])
            assert ok, "model rejected constraint %r" % (op,)
        else:
            raise ValueError(k)
    return m


