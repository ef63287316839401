"""Pins the oracle: (1) the plain-C port (oracle/pbd_oracle.c) against the golden vectors that
tests/golden/make_golden.py produced by running the unmodified reference (float build compiled
with -ffp-contract=off, and the default double build) -- bit-exact in float; (2) where the
reference-as-oracle library oracle/_ref is present, the port and the golden files against it
live.  The reference itself ships no tests or golden vectors."""
import glob
import os

import numpy as np
import pytest

from oracle import port, refdrv
from tests import kat, util

GOLD = os.path.join(util.ROOT, "tests", "golden")


def test_golden_files_present():
    assert len(glob.glob(os.path.join(GOLD, "kat_*.npz"))) == len(kat.KAT_TYPES)
    assert len(glob.glob(os.path.join(GOLD, "scene_*.npz"))) == len(util.GOLDEN_SCENES)


@pytest.mark.parametrize("type_name", kat.KAT_TYPES)
def test_port_matches_golden_known_answers(type_name):
    g = np.load(os.path.join(GOLD, "kat_%s.npz" % type_name))
    # the stored inputs must be what the generator produces today (fixtures stay reproducible)
    arrs = kat.kat_arrays(type_name, 97, seed=4321 + kat.KAT_TYPES.index(type_name))
    for k in ("verts", "masses", "bodies", "args", "x_start"):
        assert np.array_equal(arrs[k], g[k]), k
    ops = kat.kat_ops(type_name, g)
    for variant in ("f32", "f64"):
        p = port.Port(variant)
        util.apply_ref(p, ops)
        p.set_time_step_size(0.005)
        if variant == "f32":
            params = np.array([np.pad(p.constraint_params(c), (0, 24 - len(p.constraint_params(c)))) for c in range(p.num_constraints())], dtype=np.float32)
            assert util.bitwise_equal(params, g["params_f32"]), "rest data differs from the reference"
        for sweeps in (1, 2):
            p.set_array(0, g["x_start"])
            for it in range(sweeps):
                p.solve_position_constraints(it)
            x = p.positions()
            gold = g["x%d_%s" % (sweeps, variant)]
            if variant == "f32":
                assert util.bitwise_equal(x.astype(np.float32), gold), "%s sweeps=%d: %d ulp" % (
                    type_name, sweeps, util.ulp_diff(x.astype(np.float32), gold))
            else:
                # Eigen vectorises 3-vectors of double with 2-wide packets ((c0+c1)+c2) -- the port keeps
                # the float build's association, so the double instantiation agrees to rounding only
                assert util.max_err(x, gold) <= 1e-11


@pytest.mark.parametrize("name", list(util.GOLDEN_SCENES))
def test_port_matches_golden_scenes(name):
    ops, sub, iters, horizons = util.GOLDEN_SCENES[name]
    g = np.load(os.path.join(GOLD, "scene_%s.npz" % name))
    p = port.Port("f32")
    util.apply_ref(p, ops)
    p.set_params(sub, iters, 0)
    assert np.array_equal(np.array([len(x) for x in p.groups()], dtype=np.uint32), g["groups_sizes"])
    done = 0
    for steps in horizons:
        if steps > 10 and p.num_constraints() > 10000:
            continue       # keep the CPU suite short; the long horizon is covered on the GPU
        p.step(steps - done)
        done = steps
        x = p.positions().astype(np.float32)
        assert util.bitwise_equal(x, g["x_f32_%d" % steps]), "%s after %d steps: max err %.3e" % (
            name, steps, util.max_err(x, g["x_f32_%d" % steps]))
    p64 = port.Port("f64")
    util.apply_ref(p64, ops)
    p64.set_params(sub, iters, 0)
    p64.step(horizons[0])
    e = util.max_err(p64.positions(), g["x_f64_%d" % horizons[0]])
    # PBD isometric bending (C1) amplifies rounding differences; everything else is tight
    assert e <= (1e-6 if name.startswith("c1_") else 1e-9), e


@pytest.mark.skipif(not refdrv.available("f32"), reason="oracle/_ref not built (needs /root/reference)")
def test_golden_files_reproduce_from_the_live_reference():
    name = "c2_cloth50_xpbd_dist_isobend_10it"
    ops, sub, iters, horizons = util.GOLDEN_SCENES[name]
    g = np.load(os.path.join(GOLD, "scene_%s.npz" % name))
    x = util.oracle_positions(ops, 10, sub, iters, "f32").astype(np.float32)
    assert util.bitwise_equal(x, g["x_f32_10"])
    x8 = util.oracle_positions(ops, 10, sub, iters, "f32", threads=8).astype(np.float32)
    assert util.bitwise_equal(x8, g["x_f32_10"]), "the reference is thread-count independent (colouring)"


@pytest.mark.skipif(not refdrv.available("f32"), reason="oracle/_ref not built (needs /root/reference)")
def test_port_matches_live_reference_on_inverted_tets():
    for tname in ("fem_tet", "fem_tet_xpbd"):
        arrs = kat.kat_arrays(tname, 64, seed=77, static_fraction=0.1)
        xs = kat.invert_tets(arrs, 5)
        ops = kat.kat_ops(tname, arrs)
        r = refdrv.Ref("f32")
        util.apply_ref(r, ops)
        r.set_time_step_size(0.005)
        r.set_array(0, xs)
        r.solve_position_constraints(0)
        p = port.Port("f32")
        util.apply_ref(p, ops)
        p.set_time_step_size(0.005)
        p.set_array(0, xs)
        p.solve_position_constraints(0)
        assert util.max_err(r.positions(), xs) > 1e-3
        assert util.bitwise_equal(p.positions().astype(np.float32), r.positions().astype(np.float32)), tname


def test_fast_build_is_only_a_timing_baseline():
    """The -O3 -march=x86-64-v3 build contracts FMAs: it is used for cpu_baseline timing, never for parity."""
    if not (refdrv.available("fast") and refdrv.available("f32")):
        pytest.skip("oracle/_ref not built")
    ops = util.cloth_spec(20, 20, 4, 3)
    a = util.oracle_positions(ops, 3, 1, 5, "f32")
    b = util.oracle_positions(ops, 3, 1, 5, "fast")
    assert util.max_err(a, b) < 1e-4
