"""oracle/scene_ref.py -- TEST INFRASTRUCTURE ONLY: applies a scene specification
(positionbaseddynamics_amd/scenes.py: a list of the reference's own builder calls) to an oracle object
(oracle.refdrv.Ref = the reference itself, or oracle.port.Port = the plain-C restatement).  Imported by tests/ and
by bench.py's cpu_baseline leg only."""


def expand_instances(ops):
    """("instances", offsets): what SimulationModel.addInstances means for the reference -- every operation before it is
    issued again per copy, mesh builders with translation T + offset (float32), particle / model indices shifted."""
    import numpy as np
    out = []
    for op in ops:
        if op[0] != "instances":
            out.append(op)
            continue
        proto = list(out)
        n_particles = 0
        n_tri = sum(1 for o in proto if o[0] in ("tri", "trimesh"))
        n_tet = sum(1 for o in proto if o[0] in ("tet", "tetmesh"))
        for o in proto:
            if o[0] == "tri":
                n_particles += o[1] * o[2]
            elif o[0] == "tet":
                n_particles += o[1] * o[2] * o[3]
            elif o[0] in ("trimesh", "tetmesh"):
                n_particles += len(o[1])
            elif o[0] == "vertex":
                n_particles += 1
        for k, off in enumerate(op[1], start=1):
            off32 = [np.float32(v) for v in off]
            for o in proto:
                if o[0] == "tri":
                    out.append(("tri", o[1], o[2], tuple(np.float32(o[3][i]) + off32[i] for i in range(3)), o[4], o[5]))
                elif o[0] == "tet":
                    out.append(("tet", o[1], o[2], o[3], tuple(np.float32(o[4][i]) + off32[i] for i in range(3)), o[5], o[6]))
                elif o[0] in ("trimesh", "tetmesh"):
                    out.append((o[0], (np.asarray(o[1], dtype=np.float32) + np.array(off32, dtype=np.float32)).astype(np.float32), o[2]))
                elif o[0] == "vertex":
                    out.append(("vertex", tuple(np.float32(o[1][i]) + off32[i] for i in range(3))))
                elif o[0] == "mass":
                    out.append(("mass", o[1] + k * n_particles, o[2]))
                elif o[0] in ("cloth", "bending"):
                    out.append((o[0], o[1] + k * n_tri) + tuple(o[2:]))
                elif o[0] == "solid":
                    out.append((o[0], o[1] + k * n_tet) + tuple(o[2:]))
                elif o[0] == "constraint":
                    out.append(("constraint", o[1], [int(b) + k * n_particles for b in o[2]]) + tuple(o[3:]))
                else:
                    raise ValueError(o[0])
    return out


def apply_ref(ref, ops):
    """Apply a scene to an oracle.refdrv.Ref / oracle port object (same builder API)."""
    ref.reset_all()
    for op in expand_instances(ops):
        k = op[0]
        if k == "tri":
            ref.add_regular_triangle_model(op[1], op[2], op[3], op[4], op[5])
        elif k == "tet":
            ref.add_regular_tet_model(op[1], op[2], op[3], op[4], op[5], op[6])
        elif k == "trimesh":
            ref.add_triangle_model(op[1], op[2])
        elif k == "tetmesh":
            ref.add_tet_model(op[1], op[2])
        elif k == "vertex":
            ref.add_vertex(op[1])
        elif k == "mass":
            ref.set_mass(op[1], op[2])
        elif k == "cloth":
            ref.add_cloth_constraints(*op[1:])
        elif k == "bending":
            ref.add_bending_constraints(*op[1:])
        elif k == "solid":
            ref.add_solid_constraints(*op[1:])
        elif k == "constraint":
            ok = ref.add_constraint(op[1], op[2], *op[3:])
            assert ok, "oracle rejected constraint %r" % (op,)
        else:
            raise ValueError(k)
    return ref


