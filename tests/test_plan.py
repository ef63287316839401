"""Planner of the colour-fused tile schedule (positionbaseddynamics_amd/csrc/pbdx_plan.cpp), host only.

pbdx_model_plan_check packs the model's colour groups exactly as the time step does, plans tiles +
segments and then EXECUTES the plan symbolically (every particle carries a hash of its update
history): the fused schedule must hand every particle the history of the reference's
colour-sequential sweep (TimeStepController.cpp:270-286), every particle must be owned by exactly
one tile per segment and every tile-local index must map to the right particle.

It also checks the tile-to-tile dependency lists of the persistent schedule (all passes of a substep in one
launch): a host simulation lets the tiles run asynchronously in adversarial orders (most advanced first, least
advanced first, pseudo-random, one tile held back) subject only to the lists and requires every LDS fill to find
the particle versions of the previous pass in the double-buffered positions; and it checks the checker -- with
one list entry removed the simulation must report the stale read."""
import numpy as np
import pytest

from tests import util

CASES = {
    "cloth 50x50 XPBD dist+bend": util.cloth_spec(50, 50, 4, 3),
    "cloth 37x91 PBD dist + isometric": util.cloth_spec(37, 91, 1, 2),
    "cloth 40x40 FEM tri + dihedral": util.cloth_spec(40, 40, 2, 1),
    "3 cloth instances": util.cloth_spec(24, 24, 4, 3, instances=3, instance_offset=(12, 0, 0)),
    "bar 30x5x5 FEM tet": util.bar_spec(30, 5, 5, 2),
    "bar 20x6x6 XPBD dist+vol": util.bar_spec(20, 6, 6, 6),
    "bar 12x5x5 shape matching": util.bar_spec(12, 5, 5, 5),
    "irregular Delaunay cloth": util.delaunay_cloth_spec(600),
    "kitchen sink (all 13 types, mixed-type colours)": util.kitchen_sink_spec(),
    "irregular Delaunay tets, FEM": util.delaunay_solid_spec(250, solid_method=2),
    "irregular Delaunay tets, distance+volume": util.delaunay_solid_spec(250, solid_method=1),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("tile", [0, 33, 250])
def test_plan_is_equivalent_to_colour_sequential_sweep(name, tile):
    m = util.build_mine(CASES[name])
    info = m.planCheck(tile_particles=tile)
    n = m.getParticles().size()
    assert info["built"] == 1 and info["num_segments"] >= 1
    assert info["num_colours"] == len(m.getConstraintGroups())
    assert info["slots_per_sweep"] >= m.numConstraints()
    assert info["redundancy"] >= 1.0
    if tile:
        assert info["num_tiles"] == (n + tile - 1) // tile


def test_plan_respects_lds_capacity_and_segment_cap():
    m = util.build_mine(util.cloth_spec(60, 60, 4, 3))
    a = m.planCheck(tile_particles=200, lds_particles=400)
    assert a["max_local"] <= 400
    b = m.planCheck(tile_particles=200, lds_particles=10240)
    assert b["num_segments"] <= a["num_segments"]          # more LDS -> longer segments
    c = m.planCheck(tile_particles=200, max_segment_colours=1)
    assert c["num_segments"] == c["num_colours"]
    one = m.planCheck(tile_particles=3600)                  # one tile owns everything: no halo, no redundancy
    assert one["num_tiles"] == 1 and one["redundancy"] == 1.0 and one["max_local"] == 3600


def test_plan_fails_loudly_when_a_colour_cannot_fit():
    import positionbaseddynamics_amd as pbd
    m = util.build_mine(util.cloth_spec(30, 30, 4, 3))
    with pytest.raises(pbd.PbdxError):
        m.planCheck(tile_particles=200, lds_particles=201)


def test_plan_160k_particle_cloth_fits_lds_with_moderate_halo():
    """A 400x400 cloth (the BASELINE builder at a size this container plans in a second; the 1000x1000 plan itself is
    checked where it runs, tests/test_gpu_parity.py): the planner must fit 160 KiB of LDS per tile and keep the redundant
    halo work moderate."""
    m = util.build_mine(util.cloth_spec(400, 400, 4, 3))
    info = m.planCheck()
    print(info)
    assert info["max_local"] <= 10240
    assert info["num_segments"] <= 6
    assert info["redundancy"] <= 3.0


def test_plan_star_graph_more_than_64_colours_and_huge_valence():
    """A star: 200 distance constraints share particle 0 -> 200 colours of one constraint each; the hub
    has valence 200.  Long colour sequences must be split into segments and stay exact."""
    n = 200
    ops = [("vertex", (float(i), 0.0, 0.0)) for i in range(n + 1)]
    ops += [("constraint", "distance", [0, i + 1], 1.0) for i in range(n)]
    m = util.build_mine(ops)
    for tile in (0, 16, 201):
        info = m.planCheck(tile_particles=tile)
        assert info["num_colours"] == n
        assert info["num_segments"] >= n // 16      # default cap: 16 colours per launch


@pytest.mark.parametrize("name,ops,tile", [
    ("5 instanced cloths, 60 x 60", util.cloth_spec(60, 60, 4, 3, instances=5, instance_offset=(0.0, 0.0, 12.0), instanced=True), 0),
    ("5 instanced cloths, small tiles", util.cloth_spec(30, 30, 4, 3, instances=5, instance_offset=(0.3, 0.0, 5.0), instanced=True), 100),
    ("3 instanced FEM bars", util.bar_spec(14, 5, 4, 2, instances=3, instanced=True), 0),
    ("4 instanced bars, XPBD distance + volume", util.bar_spec(10, 4, 4, 6, instances=4, instanced=True), 64),
])
def test_instanced_plan_is_one_instance_replicated_and_exact(name, ops, tile):
    """for K congruent instances the planner plans ONE and replicates tiles / steps / streams with offset
    particle ids and every copy's own parameter records.  The replicated plan of the WHOLE goes through the same symbolic
    execution and asynchronous-execution checks as any other plan; it has K x the tiles and slots of the one-instance plan
    and the same redundancy."""
    from oracle.scene_ref import expand_instances
    m = util.build_mine(ops)
    K = m.numInstances()
    assert K > 1
    info = m.planCheck(tile_particles=tile)
    single = util.build_mine([op for op in ops if op[0] != "instances"])
    n1 = single.getParticles().size()
    t1 = info["num_tiles"] // K
    one = single.planCheck(tile_particles=tile if tile else (n1 + t1 - 1) // t1)
    assert info["num_tiles"] == K * one["num_tiles"] and info["slots_per_sweep"] == K * one["slots_per_sweep"]
    assert info["num_segments"] == one["num_segments"] and info["max_local"] == one["max_local"]
    assert abs(info["redundancy"] - one["redundancy"]) < 1e-12
    # the same scene built the long way (K rounds of builder calls) plans to the same amount of work per sweep when it is
    # given the same tile size (its tiles may straddle instances, so only the totals are comparable)
    plain = util.build_mine(expand_instances(ops))
    assert plain.numInstances() == 1 and plain.numConstraints() == m.numConstraints()


def test_parameter_stream_forms_and_their_conversion():
    """The two forms of a step's parameter stream (pbdx_plan.h param_float_index): planes (1 024-thread workgroups) and vector segments
    (up to 512: planes 4s..4s+3 of a slot adjacent, one 16-byte load).  Both index functions are bijections onto the step's block, the
    vector form puts a slot's planes where ONE aligned load of 4 / 3 / 2 / 1 floats finds them, and the planner's conversion
    (relayout_params, through a developer entry point) moves every (plane, slot) value to its place in the other form and back."""
    import ctypes as C
    from positionbaseddynamics_amd import _ffi
    lib = _ffi.lib
    idx = lib.pbdx_debug_param_float_index
    for planes in (1, 2, 3, 4, 5, 9, 10, 12, 13, 17):
        for slots in (1, 63, 64, 65, 200):
            groups = (slots + 63) // 64
            total = groups * planes * 64
            for vec in (0, 1):
                seen = set()
                for slot in range(groups * 64):
                    for plane in range(planes):
                        seen.add(idx(vec, planes, plane, slot))
                assert seen == set(range(total)), (planes, slots, vec)
            # vector form: the planes of a full segment are consecutive floats starting at a multiple of 4; the tail's planes are consecutive too
            for slot in (0, 1, 63, 64 * (groups - 1) + 5):
                for sgm in range(planes // 4):
                    base = idx(1, planes, 4 * sgm, slot)
                    assert base % 4 == 0 and [idx(1, planes, 4 * sgm + c, slot) for c in range(4)] == [base + c for c in range(4)]
                tail = planes % 4
                if tail:
                    base = idx(1, planes, planes - tail, slot)
                    assert [idx(1, planes, planes - tail + c, slot) for c in range(tail)] == [base + c for c in range(tail)]
            # plane form: a plane's 64 slots are consecutive
            assert [idx(0, planes, planes - 1, q) for q in range(64)] == list(range((planes - 1) * 64, planes * 64))
    # the conversion, on the plane counts of real types (compact and full layouts)
    rng = np.random.default_rng(5)
    checked = 0
    for ctype in range(13):
        for compact in (0, 1):
            for slots in (1, 64, 150):
                np_out = C.c_uint32(0)
                probe = np.zeros(64 * 64, dtype=np.float32)
                assert lib.pbdx_debug_relayout_params(ctype, compact, 1, 0, probe.ctypes.data_as(C.POINTER(C.c_float)), C.byref(np_out)) == 0
                planes = np_out.value
                if planes == 0:
                    continue
                groups = (slots + 63) // 64
                values = rng.random((planes, groups * 64)).astype(np.float32)         # value of (plane, slot)
                block = np.zeros(groups * planes * 64, dtype=np.float32)
                for plane in range(planes):
                    for slot in range(groups * 64):
                        block[idx(0, planes, plane, slot)] = values[plane, slot]
                start = block.copy()
                assert lib.pbdx_debug_relayout_params(ctype, compact, slots, 0, block.ctypes.data_as(C.POINTER(C.c_float)), None) == 0
                for plane in range(planes):
                    for slot in range(0, groups * 64, 7):
                        assert block[idx(1, planes, plane, slot)] == values[plane, slot], (ctype, compact, slots, plane, slot)
                assert lib.pbdx_debug_relayout_params(ctype, compact, slots, 1, block.ctypes.data_as(C.POINTER(C.c_float)), None) == 0
                assert np.array_equal(block, start)
                checked += 1
    assert checked > 20
