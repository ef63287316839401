"""Known-answer-test inputs: seeded random, mutually disjoint constraints of one type.
`kat_arrays` produces plain numpy arrays (stored in the golden fixtures); `kat_ops` turns them
into a scene operation list for tests.util.apply_ref / build_mine."""
import numpy as np

KAT_TYPES = ["distance", "distance_xpbd", "dihedral", "isometric_bending", "isometric_bending_xpbd", "fem_triangle",
             "strain_triangle", "volume", "volume_xpbd", "fem_tet", "fem_tet_xpbd", "strain_tet", "shape_matching"]
NUM_BODIES = {"distance": 2, "distance_xpbd": 2, "fem_triangle": 3, "strain_triangle": 3}


def _rp(rng, n, spread=1.0):
    return (rng.standard_normal((n, 3)) * spread).astype(np.float32)


def kat_arrays(type_name, n_constraints, seed, static_fraction=0.2, perturb=0.08):
    rng = np.random.default_rng(seed)
    nb = NUM_BODIES.get(type_name, 4)
    pts = []
    for c in range(n_constraints):
        if type_name in ("dihedral", "isometric_bending", "isometric_bending_xpbd"):
            # two triangles sharing the edge (p2,p3), opened by 2..3 rad
            e0 = _rp(rng, 1)[0]
            d = _rp(rng, 1)[0]
            d /= np.linalg.norm(d)
            e1 = e0 + d * np.float32(0.5 + rng.random())
            a = _rp(rng, 1)[0]
            a -= d * np.dot(a, d)
            a /= np.linalg.norm(a)
            th = np.float32(2.0 + rng.random())
            b = a * np.cos(th) + np.cross(d, a) * np.sin(th)
            mid = 0.5 * (e0 + e1)
            p = [mid + a * np.float32(0.4 + 0.3 * rng.random()), mid + b * np.float32(0.4 + 0.3 * rng.random()), e0, e1]
        elif nb == 4:
            base = _rp(rng, 1, 3.0)[0]
            p = [base, base + [1, 0, 0], base + [0, 1, 0], base + [0, 0, 1]]
            p = [np.asarray(q, dtype=np.float32) + _rp(rng, 1, 0.15)[0] for q in p]
        elif nb == 3:
            base = _rp(rng, 1, 3.0)[0]
            p = [base, base + [1, 0, 0.2], base + [0.1, 0, 1]]
            p = [np.asarray(q, dtype=np.float32) + _rp(rng, 1, 0.1)[0] for q in p]
        else:
            base = _rp(rng, 1, 3.0)[0]
            p = [base, base + _rp(rng, 1, 0.5)[0]]
        pts.extend(np.asarray(q, dtype=np.float32) for q in p)
    verts = np.array(pts, dtype=np.float32)
    n = len(verts)
    masses = np.where(rng.random(n) < static_fraction, 0.0, 0.5 + rng.random(n)).astype(np.float32)
    bodies = np.arange(n_constraints * nb, dtype=np.uint32).reshape(n_constraints, nb)
    args = []
    nclusters = np.zeros((n_constraints, 4), dtype=np.uint32)
    for c in range(n_constraints):
        if type_name in ("distance", "dihedral", "isometric_bending", "volume"):
            a = [np.float32(0.1 + 0.9 * rng.random())]
        elif type_name in ("distance_xpbd", "volume_xpbd"):
            a = [np.float32(10 ** rng.uniform(2, 5))]
        elif type_name == "isometric_bending_xpbd":
            a = [np.float32(10 ** rng.uniform(0, 3))]
        elif type_name == "fem_triangle":
            a = [1.0, 1.0, 1.0, 0.3, 0.3]
        elif type_name == "strain_triangle":
            a = [1.0, 1.0, 1.0, c % 2, (c // 2) % 2]
        elif type_name == "fem_tet":
            a = [np.float32(0.5 + rng.random()), 0.3]
        elif type_name == "fem_tet_xpbd":
            a = [np.float32(10 ** rng.uniform(3, 6)), 0.3]
        elif type_name == "strain_tet":
            a = [1.0, 1.0, c % 2, (c // 2) % 2]
        elif type_name == "shape_matching":
            a = [np.float32(0.2 + 0.8 * rng.random())]
            nclusters[c] = [1 + (c + k) % 4 for k in range(4)]
        args.append([float(v) for v in a])
    x_start = (verts + rng.standard_normal(verts.shape).astype(np.float32) * np.float32(perturb)).astype(np.float32)
    return {"verts": verts, "masses": masses, "bodies": bodies, "args": np.array(args, dtype=np.float64),
            "nclusters": nclusters, "x_start": x_start}


def kat_ops(type_name, arrs):
    ops = [("vertex", v) for v in arrs["verts"]]
    # shape matching captures the inverse masses at init: set masses before creating constraints
    ops += [("mass", i, float(m)) for i, m in enumerate(arrs["masses"])]
    for c, b in enumerate(arrs["bodies"]):
        a = list(arrs["args"][c])
        if type_name == "strain_triangle":
            a = a[:3] + [bool(a[3]), bool(a[4])]
        elif type_name == "strain_tet":
            a = a[:2] + [bool(a[2]), bool(a[3])]
        if type_name == "shape_matching":
            ops.append(("constraint", type_name, [int(v) for v in b], [int(v) for v in arrs["nclusters"][c]], a[0]))
        else:
            ops.append(("constraint", type_name, [int(v) for v in b]) + tuple(a))
    return ops


def invert_tets(arrs, seed):
    """Push vertex 3 of every tet through / close to the opposite face so FEM tets take the
    inversion-handling (SVD) branch: volume ratio < 0.2 or negative."""
    rng = np.random.default_rng(seed)
    x = arrs["verts"].copy()
    for c in range(len(arrs["bodies"])):
        p = x[4 * c:4 * c + 4]
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n /= np.linalg.norm(n)
        h = np.dot(p[3] - p[0], n)
        p[3] -= n * np.float32(h * (1.0 + 0.6 * rng.random()) if c % 2 else h * 0.95)
    return x.astype(np.float32)
