"""oracle/scene_ref.py -- TEST INFRASTRUCTURE ONLY: applies a scene specification
(positionbaseddynamics_amd/scenes.py: a list of the reference's own builder calls) to an oracle object
(oracle.refdrv.Ref = the reference itself, or oracle.port.Port = the plain-C restatement).  Imported by tests/ and
by bench.py's cpu_baseline leg only."""


def apply_ref(ref, ops):
    """Apply a scene to an oracle.refdrv.Ref / oracle port object (same builder API)."""
    ref.reset_all()
    for op in ops:
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


