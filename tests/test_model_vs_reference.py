"""Host model mirror (pbdx_model_*) against the oracle: mesh topology, constraint creation order,
rest data and the greedy colouring must be INTEGER-/BIT-exact, because the colouring is the
Gauss-Seidel order of the solver.  The oracle is the reference itself when oracle/_ref is built
(this container and the GPU box), otherwise the plain-C port."""
import numpy as np
import pytest

from tests import util


def _compare(ops, sample=4000):
    o = util.get_oracle("f32")
    util.apply_ref(o, ops)
    m = util.build_mine(ops)
    pd = m.getParticles()
    assert o.num_particles() == pd.size()
    for which in (1, 0, 6, 7):
        assert util.bitwise_equal(o.get_array(which).astype(np.float32), pd.array(which)), "particle array %d" % which
    tr, tm = o.constraint_types(), m.constraintTypes()
    assert np.array_equal(tr, tm)
    n = len(tr)
    for c in range(0, n, max(1, n // sample)):
        assert np.array_equal(o.constraint_bodies(c), m.constraintBodies(c)), "bodies of constraint %d" % c
        pr = o.constraint_params(c).astype(np.float32)
        assert util.bitwise_equal(pr, m.constraintParams(c)), "rest data of constraint %d (type %d)" % (c, tr[c])
    gr, gm = o.groups(), m.getConstraintGroups()
    assert len(gr) == len(gm)
    for a, b in zip(gr, gm):
        assert np.array_equal(a, b)
    return o, m


@pytest.mark.parametrize("cloth_method,bending_method", [(1, 2), (4, 3), (2, 1), (3, 0), (1, 1)])
def test_cloth_scene_build_and_colouring(cloth_method, bending_method):
    o, m = _compare(util.cloth_spec(50, 50, cloth_method, bending_method))
    if (cloth_method, bending_method) == (1, 2):
        # C1 of BASELINE.json: 2500 particles, 7301 + 7105 constraints, 26 colours
        assert m.numConstraints() == 14406 and len(m.getConstraintGroups()) == 26
    edges_ref = o.triangle_model_edges(0)
    assert np.array_equal(edges_ref, m.getTriangleModels()[0].getEdges())


@pytest.mark.parametrize("solid_method", [1, 2, 3, 4, 5, 6])
def test_bar_scene_build_and_colouring(solid_method):
    o, m = _compare(util.bar_spec(30, 5, 5, solid_method))
    assert np.array_equal(o.tet_model_edges(0), m.getTetModels()[0].getEdges())
    if solid_method == 2:
        assert m.numConstraints() == 2320 and len(m.getConstraintGroups()) == 38


def test_non_square_and_offset_instances():
    _compare(util.cloth_spec(37, 23, 4, 3, width=7.0, height=3.0, instances=3, instance_offset=(12, 0, 0)))


def test_ensemble_colouring_is_k_times_single():
    """first-fit colouring of K appended identical instances == K x the single colouring."""
    m1 = util.build_mine(util.cloth_spec(20, 20, 4, 3))
    mk = util.build_mine(util.cloth_spec(20, 20, 4, 3, instances=4))
    assert [len(g) for g in mk.getConstraintGroups()] == [4 * len(g) for g in m1.getConstraintGroups()]


def test_irregular_triangle_soup_and_tets():
    rng = np.random.default_rng(7)
    # a small irregular (non-grid) triangle mesh: fan + strip, including a boundary and a non-manifold edge
    pts = rng.standard_normal((12, 3))
    faces = [(0, 1, 2), (0, 2, 3), (0, 3, 4), (0, 4, 5), (2, 1, 6), (6, 1, 7), (7, 1, 8), (3, 2, 9), (9, 2, 6), (10, 11, 0), (0, 2, 11)]
    ops = [("trimesh", pts, faces), ("mass", 0, 0.0), ("cloth", 0, 1, 0.7, 1.0, 1.0, 1.0, 0.3, 0.3, False, False), ("bending", 0, 2, 0.05),
           ("cloth", 0, 2, 1.0, 0.9, 1.1, 0.8, 0.25, 0.2, False, False)]
    _compare(ops)
    tp = rng.standard_normal((9, 3))
    tets = [(0, 1, 2, 3), (1, 2, 3, 4), (2, 3, 4, 5), (5, 6, 7, 8), (0, 2, 5, 8)]
    _compare([("tetmesh", tp, tets), ("solid", 0, 6, 1e4, 0.3, 1e4, False, False), ("solid", 0, 5, 0.5, 0.3, 1.0, False, False)])


def test_more_than_64_colours():
    """The bitset colouring must agree with the reference's per-group byte maps beyond one machine word."""
    n = 80
    ops = [("vertex", (float(i), 0.0, 0.0)) for i in range(n + 1)]
    # a star: every constraint shares particle 0 => one colour per constraint
    ops += [("constraint", "distance", [0, i + 1], 1.0) for i in range(n)]
    o, m = _compare(ops)
    assert len(m.getConstraintGroups()) == n


def test_empty_model_and_reset():
    import positionbaseddynamics_amd as pbd
    m = pbd.SimulationModel()
    assert m.numConstraints() == 0 and m.getConstraintGroups() == [] and m.getParticles().size() == 0
    m.addRegularTriangleModel(5, 5)
    x0 = m.getParticles().array(1).copy()
    m.getParticles().set_array(0, x0 + 1.0)
    m.getParticles().set_array(2, np.ones_like(x0))
    m.reset()                                      # SimulationModel::reset  SimulationModel.cpp:270-304
    assert np.array_equal(m.getParticles().positions(), x0) and not m.getParticles().velocities().any()
    m.cleanup()
    assert m.getParticles().size() == 0 and m.numConstraints() == 0


# ---------------------------------------------------------------------------
# instanced models (SimulationModel.addInstances): SURVEY 8e / 8f rank 3
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("spec", ["cloth", "bar_fem", "bar_distvol", "mixed"])
def test_instanced_model_equals_the_reference_built_instance_after_instance(spec):
    """One prototype + addInstances(K - 1 offsets) must BE the model the reference builds when the same builders are called K
    times with translated meshes: particle arrays, constraint order, per-instance rest data (bitwise: every copy's rest
    lengths / matrices are evaluated at ITS rest positions) and the colour groups -- integer-exact at K = 3 -- although
    topology, constraints and groups are stored once."""
    if spec == "cloth":
        ops = util.cloth_spec(21, 17, 4, 3, width=7.0, height=3.0, instances=3, instance_offset=(0.37, 0.0, 12.1), instanced=True)
    elif spec == "bar_fem":
        ops = util.bar_spec(9, 4, 3, 2, instances=3, instance_offset=(0.0, 0.1, 3.3), instanced=True)
    elif spec == "bar_distvol":
        ops = util.bar_spec(8, 3, 3, 6, instances=3, instanced=True)
    else:
        rng = np.random.default_rng(11)
        pts = rng.standard_normal((10, 3)).astype(np.float32)
        faces = [(0, 1, 2), (0, 2, 3), (0, 3, 4), (2, 1, 6), (6, 1, 7), (3, 2, 9), (9, 2, 6), (4, 3, 8), (8, 3, 9), (5, 0, 4)]
        ops = [("tri", 6, 5, (0.0, 1.0, 0.0), util.rot_x_half_pi(), (2.0, 2.0)), ("trimesh", pts, faces), ("vertex", (3.0, 3.0, 3.0)),
               ("mass", 0, 0.0), ("cloth", 0, 4, 5e4, 1.0, 1.0, 1.0, 0.3, 0.3, False, False), ("bending", 0, 3, 50.0),
               ("cloth", 1, 2, 1.0, 0.9, 1.1, 0.8, 0.25, 0.2, False, False), ("bending", 1, 1, 0.02),
               ("constraint", "distance", [3, 30 + 10], 0.5),
               ("instances", [(0.0, 0.0, 5.5), (7.25, 0.0, -3.0)])]
    o, m = _compare(ops, sample=100000)          # every constraint
    assert m.numInstances() == 3
    # the same model built the long way round
    from oracle.scene_ref import expand_instances
    plain = util.build_mine(expand_instances(ops))
    assert plain.numInstances() == 1 and plain.numConstraints() == m.numConstraints()
    for which in (0, 1, 2, 4, 5, 6, 7):
        assert util.bitwise_equal(plain.getParticles().array(which), m.getParticles().array(which)), which
    assert [list(g) for g in plain.getConstraintGroups()] == [list(g) for g in m.getConstraintGroups()]
    assert len(plain.getTriangleModels()) == len(m.getTriangleModels()) and len(plain.getTetModels()) == len(m.getTetModels())
    for a, b in zip(plain.getTriangleModels(), m.getTriangleModels()):
        assert a.getIndexOffset() == b.getIndexOffset() and np.array_equal(a.getEdges(), b.getEdges()) and np.array_equal(a.getParticleMesh().getFaces(), b.getParticleMesh().getFaces())
    for a, b in zip(plain.getTetModels(), m.getTetModels()):
        assert a.getIndexOffset() == b.getIndexOffset() and np.array_equal(a.getEdges(), b.getEdges())


def test_instanced_model_is_sealed_and_rejects_non_congruent_copies():
    import positionbaseddynamics_amd as pbd
    m = util.build_mine(util.cloth_spec(6, 6, 4, 3, instances=2, instance_offset=(0, 0, 3.0), instanced=True))
    nc = m.numConstraints()
    assert not m.addDistanceConstraint(0, 1, 1.0) and m.numConstraints() == nc
    with pytest.raises(pbd.PbdxError):
        m.addRegularTriangleModel(3, 3)
    with pytest.raises(pbd.PbdxError):
        m.addInstances([(1.0, 0.0, 0.0)])
    # user parameters live in the prototype: editing prototype constraint 5 changes constraint 5 of every copy
    p = m.constraintParams(5)
    p[1] = 77.0
    m.setConstraintParams(5, p)
    assert m.constraintParams(5 + nc // 2)[1] == 77.0
    with pytest.raises(pbd.PbdxError):
        m.setConstraintParams(5 + nc // 2, p)
    m.cleanup()
    assert m.numInstances() == 1 and m.numConstraints() == 0
