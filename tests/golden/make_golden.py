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


if __name__ == "__main__":
    main()
