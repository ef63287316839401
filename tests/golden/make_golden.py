#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (oracle/_ref, built from the
unmodified sources under /root/reference by oracle/Makefile; float build with -ffp-contract=off and
the default double build).  The reference ships no tests or golden vectors of its own
(SURVEY.md section 4), so these outputs of the reference's own code are what pins the oracle port
and the GPU path.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refdrv  # noqa: E402
from tests import util  # noqa: E402
from tests.kat import KAT_TYPES, kat_arrays, kat_ops  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

SCENES = util.GOLDEN_SCENES


def main():
    for t in KAT_TYPES:
        arrs = kat_arrays(t, 97, seed=4321 + KAT_TYPES.index(t))
        ops = kat_ops(t, arrs)
        out = dict(arrs)
        for variant in ("f32", "f64"):
            r = refdrv.Ref(variant)
            util.apply_ref(r, ops)
            r.set_time_step_size(0.005)
            for sweeps in (1, 2):
                r.set_array(0, arrs["x_start"])
                for it in range(sweeps):
                    r.solve_position_constraints(it)
                x = r.positions()
                out["x%d_%s" % (sweeps, variant)] = x.astype(np.float32) if variant == "f32" else x
            if variant == "f32":
                out["params_f32"] = np.array([np.pad(r.constraint_params(c), (0, 24 - len(r.constraint_params(c)))) for c in range(r.num_constraints())], dtype=np.float32)
        np.savez_compressed(os.path.join(OUT, "kat_%s.npz" % t), **out)
        print("kat", t, "done")
    for name, (ops, sub, iters, horizons) in SCENES.items():
        out = {"sub_steps": sub, "iters": iters, "horizons": np.array(horizons)}
        for variant in ("f32", "f64"):
            for steps in horizons:
                o = util.oracle_run(ops, steps, sub, iters, variant)
                x = o.positions()
                out["x_%s_%d" % (variant, steps)] = x.astype(np.float32) if variant == "f32" else x
                if variant == "f32" and steps == horizons[0]:
                    out["groups_sizes"] = np.array([len(g) for g in o.groups()], dtype=np.uint32)
                    out["v_f32_%d" % steps] = o.get_array(2).astype(np.float32)
        np.savez_compressed(os.path.join(OUT, "scene_%s.npz" % name), **out)
        print("scene", name, "done")


def tet_contact_golden():
    """Two colliding bars with box distance fields (tests/tetcontact_util.two_bar_scene): the hierarchies the reference built, and the
    reference's state and contact list after 71 / 76 steps -- everything a parity test of the engine's detection and of the
    contact solve needs without the reference at hand."""
    from tests import tetcontact_util as tcu
    ref = refdrv.Ref("f32")
    objs = tcu.two_bar_scene(ref)
    ref.set_params(1, 5, 0)
    out = {"x0": ref.get_array(1).astype(np.float32), "w": ref.get_array(7).astype(np.float32), "tolerance": np.float32(0.01), "steps": np.array([71, 76])}
    for q, co in enumerate(objs):
        info = ref.tet_model_info(q)
        shape, invert, params = ref.collision_object_shape(co)
        out["c%d_meta" % q] = np.array([shape, invert, info["offset"], info["num_vertices"], info["num_tets"], q], dtype=np.int64)
        out["c%d_params" % q] = np.asarray(params, dtype=np.float32)
        out["c%d_tets" % q] = np.asarray(info["tets"], dtype=np.uint32)
        out["c%d_initial_x" % q] = np.asarray(info["initial_x"], dtype=np.float32)
        out["c%d_initial_R" % q] = np.asarray(info["initial_R"], dtype=np.float32)
        for which, name in ((0, "points"), (1, "tets"), (2, "rest")):
            b = ref.bvh(co, which)
            out["c%d_%s_lst" % (q, name)] = np.asarray(b["lst"], dtype=np.uint32)
            out["c%d_%s_nodes" % (q, name)] = np.asarray(b["nodes"], dtype=np.int32)
            out["c%d_%s_hulls" % (q, name)] = np.asarray(b["hulls"], dtype=np.float32)
    done = 0
    for steps in (71, 76):
        ref.step(steps - done)
        done = steps
        out["x_%d" % steps] = ref.positions().astype(np.float32)
        out["v_%d" % steps] = ref.get_array(2).astype(np.float32)
        out["contacts_%d" % steps] = tcu.oracle_contacts_as_engine_records(ref)
    ref.reset_all()
    np.savez_compressed(os.path.join(OUT, "tetcontact_two_bars.npz"), **out)
    print("tet contact scene done:", len(out["contacts_71"]), len(out["contacts_76"]), "contacts")


def tet_contact_velocity_golden():
    """The velocity part of a particle-tet contact through the reference's own init_ / velocitySolve_ParticleTetContactConstraint (friction 0) on
    the adversarial inputs of tests/tetcontact_util.velocity_kat_inputs."""
    from tests import tetcontact_util as tcu
    ref = refdrv.Ref("f32")
    inp = tcu.velocity_kat_inputs()
    out = np.array([ref.kat_tet_contact_velocity(row.astype(np.float64)) for row in inp], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "tetcontact_velocity_kat.npz"), inputs=inp, outputs=out)
    print("tet contact velocity KAT done:", len(inp), "cases,", int((out[:, 3] < 0).sum()), "with pMax < 0")


def armadillo_scene_golden():
    """BASELINE configs[4] in the shape that can be pinned (tests/tetcontact_util.armadillo_scene): what the reference's own TetGenLoader and the
    scene file's placement make of data/models/armadillo_4k.node/.ele (rest positions, tets, initial transforms), the three bounding-sphere hierarchies
    the reference built per solid, the floor and the collision ranges as the reference's collision objects store them, and the reference's state and
    deformable-contact list after 120 / 260 steps at 8 substeps (BASELINE.json) and at 5 (the scene file).  Written into the PACKAGE
    (positionbaseddynamics_amd/data): bench.py's configs[4] line runs where neither /root/reference nor its data files exist."""
    from tests import tetcontact_util as tcu
    ref = refdrv.Ref("f32")
    out = {}
    for sub in (8, 5):
        objs = tcu.armadillo_scene(ref, sub)
        if sub == 8:
            out["x0"] = ref.get_array(1).astype(np.float32)
            out["mass"] = ref.get_array(6).astype(np.float32)
            out["w"] = ref.get_array(7).astype(np.float32)
            out["box"] = np.asarray(tcu.ARMADILLO_BOX, dtype=np.float32)
            cols, ranges, tol, stiff = ref.collision_objects()
            assert len(cols) == 1 and len(ranges) == 3
            rb = cols[0]
            out["rb_shape"] = np.int64(rb["shape"]); out["rb_invert"] = np.int64(rb["invert"]); out["rb_params"] = np.asarray(rb["params"], dtype=np.float32)
            for k in ("com", "R", "v1", "v2"):
                out["rb_" + k] = np.asarray(rb[k], dtype=np.float32)
            out["rb_restitution"] = np.float32(rb["restitution"]); out["rb_friction"] = np.float32(rb["friction"]); out["rb_body_index"] = np.int64(rb["body_index"])
            out["ranges"] = np.asarray(ranges, dtype=np.float64)
            out["tolerance"] = np.float32(tol); out["contact_stiffness"] = np.float32(stiff); out["max_iterations_v"] = np.int64(5)
            out["time_step"] = np.float32(0.01); out["iterations"] = np.int64(1)
            out["solid"] = np.asarray([2, 1.0, 0.2, 1.0], dtype=np.float64)          # addSolidConstraints: method, stiffness, Poisson ratio, volume stiffness
            for q, co in enumerate(objs):
                info = ref.tet_model_info(q)
                shape, invert, params = ref.collision_object_shape(co)
                out["c%d_meta" % q] = np.array([shape, invert, info["offset"], info["num_vertices"], info["num_tets"], q], dtype=np.int64)
                out["c%d_params" % q] = np.asarray(params, dtype=np.float32)
                out["c%d_tets" % q] = np.asarray(info["tets"], dtype=np.uint32)
                out["c%d_initial_x" % q] = np.asarray(info["initial_x"], dtype=np.float32)
                out["c%d_initial_R" % q] = np.asarray(info["initial_R"], dtype=np.float32)
                out["c%d_restitution" % q] = np.float32(tcu.ARMADILLO_PLACEMENT[q][3])
                for which, name in ((0, "points"), (1, "tets"), (2, "rest")):
                    b = ref.bvh(co, which)
                    out["c%d_%s_lst" % (q, name)] = np.asarray(b["lst"], dtype=np.uint32)
                    out["c%d_%s_nodes" % (q, name)] = np.asarray(b["nodes"], dtype=np.int32)
                    out["c%d_%s_hulls" % (q, name)] = np.asarray(b["hulls"], dtype=np.float32)
        done, tet_total, rb_total = 0, 0, 0
        for steps in (120, 260):
            for _ in range(steps - done):
                ref.step(1)
                tet_total += ref.num_particle_solid_contacts()
                rb_total += len(ref.contacts())
            done = steps
            out["x_sub%d_%d" % (sub, steps)] = ref.positions().astype(np.float32)
            out["v_sub%d_%d" % (sub, steps)] = ref.get_array(2).astype(np.float32)
            out["contacts_sub%d_%d" % (sub, steps)] = tcu.oracle_contacts_as_engine_records(ref)
        out["contact_totals_sub%d" % sub] = np.array([tet_total, rb_total], dtype=np.int64)
        print("armadillo scene, %d substeps: %d deformable contacts, %d floor contacts over 260 steps" % (sub, tet_total, rb_total))
    out["steps"] = np.array([120, 260])
    ref.reset_all()
    dst = os.path.join(ROOT, "positionbaseddynamics_amd", "data")
    os.makedirs(dst, exist_ok=True)
    np.savez_compressed(os.path.join(dst, "armadillo_collision_scene.npz"), **out)
    print("armadillo scene fixture: %d particles, %d bytes" % (len(out["x0"]), os.path.getsize(os.path.join(dst, "armadillo_collision_scene.npz"))))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "armadillo":
        armadillo_scene_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "tetcontact":
        tet_contact_golden()
        tet_contact_velocity_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "tetcontact_velocity":
        tet_contact_velocity_golden()
    else:
        main()
        tet_contact_golden()
        tet_contact_velocity_golden()
        armadillo_scene_golden()
