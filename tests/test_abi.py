"""C-ABI surface: libpbdx.so loads, exports every symbol include/pbdx.h declares, and refuses to
compute without a GPU (no CPU fallback exists)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests import util

HEADER = os.path.join(util.ROOT, "include", "pbdx.h")
DEBUG_HEADER = os.path.join(util.ROOT, "include", "pbdx_debug.h")


def declared_symbols(headers=(HEADER, DEBUG_HEADER)):
    syms = set()
    for h in headers:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        inline = set(re.findall(r"static\s+inline\s+[a-z0-9_]+\s+(pbdx_[a-z0-9_]+)\s*\(", text))     # header-only helpers: no symbol
        syms |= set(re.findall(r"\b(pbdx_[a-z0-9_]+)\s*\(", text)) - inline
    return sorted(syms)


def test_debug_entry_points_are_not_part_of_the_boundary():
    """include/pbdx.h is what a reference-side binding sees; developer aids live in include/pbdx_debug.h."""
    assert not [s for s in declared_symbols((HEADER,)) if s.startswith("pbdx_debug_")]
    assert all(s.startswith("pbdx_debug_") for s in declared_symbols((DEBUG_HEADER,)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 80
    for must in ("pbdx_solver_create", "pbdx_solver_add_batch", "pbdx_solver_step", "pbdx_model_init_constraint_groups",
                 "pbdx_timestep_step", "pbdx_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import positionbaseddynamics_amd._ffi as ffi
    lib = ctypes.CDLL(ffi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    bound = {s[0] for s in ffi.SIGNATURES}
    unbound = [s for s in declared_symbols() if s not in bound]
    assert not unbound, "declared but not bound in _ffi.py: %s" % unbound


def test_option_numbers_match_the_header():
    """The launch options of include/pbdx.h and the constants of the Python mirror (Solver.OPT_*) are one list;
    the ctypes mirrors of the info structs have the size the C structs have (field by field on a natural-alignment ABI)."""
    import positionbaseddynamics_amd as pbd
    import positionbaseddynamics_amd._ffi as ffi
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    header = {name: int(val) for name, val in re.findall(r"\bPBDX_(OPT_[A-Z_]+)\s*=\s*(\d+)", text)}
    mirror = {k: v for k, v in vars(pbd.Solver).items() if k.startswith("OPT_")}
    assert header == mirror and len(header) >= 12
    ctype_of = {"int": ctypes.c_int, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64, "double": ctypes.c_double, "float": ctypes.c_float}
    for cname, mirror_struct in (("pbdx_plan_info", ffi.PlanInfo), ("pbdx_segment_info", ffi.SegmentInfo), ("pbdx_persistent_info", ffi.PersistentInfo)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.split(None, 1)
            fields += [(n.strip(), ctype_of[ctype]) for n in names.split(",")]
        assert [(n, t) for n, t in mirror_struct._fields_] == fields, cname


def test_type_table():
    import positionbaseddynamics_amd as pbd
    T = pbd.ConstraintType
    assert [T.num_bodies(t) for t in range(T.COUNT)] == [2, 2, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4]
    assert [T.param_stride(t) for t in range(T.COUNT)] == [2, 2, 2, 17, 17, 10, 9, 2, 2, 12, 12, 13, 24]
    # SURVEY.md 8d algorithmic bytes per projection
    assert T.algorithmic_bytes(T.DISTANCE_XPBD) == 76 and T.algorithmic_bytes(T.ISOMETRIC_BENDING_XPBD) == 200
    assert T.algorithmic_bytes(T.FEM_TET) == 168 and T.num_bodies(99) == 0
    assert T.name(T.FEM_TET_XPBD) == "XPBD_FEMTetConstraint"


def test_no_cpu_fallback(have_gpu):
    import positionbaseddynamics_amd as pbd
    if have_gpu:
        pytest.skip("GPU present: the refusal path is only reachable without a device")
    # a time step can be created and configured without a device (like the reference's), but every call
    # that would compute refuses: step, stepResident, project, the engine handle
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 7)
    assert ts.getValueUInt(pbd.TimeStepController.MAX_ITERATIONS) == 7
    m = pbd.SimulationModel()
    m.addRegularTriangleModel(4, 4)
    m.addClothConstraints(0, 4, 1000.0)
    x0 = m.getParticles().positions().copy()
    for call in (lambda: ts.step(m), lambda: ts.stepResident(m, 1), lambda: ts.project(m, 1), lambda: ts.solver(), lambda: ts.syncFromHost(m)):
        with pytest.raises(pbd.PbdxError) as e:
            call()
        assert e.value.code == 2
    assert "no CPU fallback" in str(e.value)
    assert np.array_equal(m.getParticles().positions(), x0), "a refused step must not move anything"
    with pytest.raises(pbd.PbdxError):
        pbd.Solver()


def test_error_reporting_and_argument_checks():
    import positionbaseddynamics_amd as pbd
    m = pbd.SimulationModel()
    m.addRegularTriangleModel(4, 4)
    assert not m.addDistanceConstraint(0, 99, 1.0)          # out-of-range particle -> false like a failed initConstraint
    assert b"out of range" in pbd.lib.pbdx_last_error()
    with pytest.raises(pbd.PbdxError):
        m.addClothConstraints(7, 1, 1.0)                    # no such triangle model
    with pytest.raises(pbd.PbdxError):
        m.addRegularTriangleModel(1, 5)
    assert not m.addShapeMatchingConstraint(3, [0, 1, 2], [1, 1, 1], 1.0)   # only 4-particle clusters are on the path
    # degenerate elements are rejected exactly like the reference's init functions
    m2 = pbd.SimulationModel()
    for p in ((0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0)):
        m2.getParticles().addVertex(p)
    assert not m2.addFEMTetConstraint(0, 1, 2, 3, 1.0, 0.3)
    assert not m2.addFEMTriangleConstraint(0, 1, 2, 1, 1, 1, 0.3, 0.3)
    assert m2.numConstraints() == 0


def test_pypbd_style_surface():
    """The python mirror keeps pypbd's names (pyPBD/*Module.cpp) for the path's classes."""
    import positionbaseddynamics_amd as pbd
    for cls, names in ((pbd.SimulationModel, ["addRegularTriangleModel", "addRegularTetModel", "addTriangleModel", "addTetModel",
                                              "addDistanceConstraint", "addDistanceConstraint_XPBD", "addDihedralConstraint",
                                              "addIsometricBendingConstraint", "addIsometricBendingConstraint_XPBD",
                                              "addFEMTriangleConstraint", "addStrainTriangleConstraint", "addVolumeConstraint",
                                              "addVolumeConstraint_XPBD", "addFEMTetConstraint", "addStrainTetConstraint",
                                              "addShapeMatchingConstraint", "addClothConstraints", "addBendingConstraints",
                                              "addSolidConstraints", "getParticles", "getTriangleModels", "getTetModels",
                                              "getConstraintGroups", "initConstraintGroups", "reset", "cleanup"]),
                       (pbd.ParticleData, ["addVertex", "getPosition", "setPosition", "getPosition0", "getMass", "getInvMass",
                                           "setMass", "getVelocity", "setVelocity", "getAcceleration", "getNumberOfParticles",
                                           "size", "getVertices"]),
                       (pbd.TimeStepController, ["step", "reset", "init", "setValueUInt", "getValueUInt", "NUM_SUB_STEPS",
                                                 "MAX_ITERATIONS", "MAX_ITERATIONS_V", "VELOCITY_UPDATE_METHOD",
                                                 "ENUM_VUPDATE_FIRST_ORDER", "ENUM_VUPDATE_SECOND_ORDER"]),
                       (pbd.Simulation, ["getCurrent", "setCurrent", "hasCurrent", "init", "initDefault", "reset", "getModel",
                                         "setModel", "getTimeStep", "setTimeStep"]),
                       (pbd.TimeManager, ["getCurrent", "getTime", "setTime", "getTimeStepSize", "setTimeStepSize"])):
        for n in names:
            assert hasattr(cls, n), "%s.%s" % (cls.__name__, n)
    pd_model = pbd.SimulationModel()
    pd_model.addRegularTriangleModel(3, 3)
    pd = pd_model.getParticles()
    pd.setMass(2, 4.0)
    assert pd.getMass(2) == 4.0 and pd.getInvMass(2) == 0.25
    pd.setMass(2, 0.0)
    assert pd.getInvMass(2) == 0.0
    view = pd.getVertices()
    view[1, 1] = 42.0                      # zero-copy: writes land in the model
    assert pd.getPosition(1)[1] == 42.0


def test_bounds_report_of_the_product_build_says_unchecked():
    """The product library compiles none of the range checks of the sanitizer-grade debug build (csrc/pbdx_bounds.h: the macros fold to the plain
    access) and says so; needs no GPU."""
    import os
    import positionbaseddynamics_amd as pbd
    if "bounds" in os.path.basename(os.environ.get("PBDX_LIB", "")):
        pytest.skip("the library under test IS the debug build")
    rep = pbd.bounds_report(0)
    assert rep["checked"] is False and rep["violations"] == 0


def test_no_caller_memory_is_handed_to_the_gpu():
    """Host memory the library does not own never reaches the device by ADDRESS: the built libraries do not import hipHostRegister, and
    in the sources every copy with a host side either goes through csrc/pbdx_hostio.hip (the library's page-locked bounce buffer), through
    an engine's page-locked mirror, or is one of the 32-byte counter reads (profiles/HISTORY.md [9]: a GPU memory fault at a host heap
    address).  Needs no GPU."""
    import glob
    import subprocess
    libs = glob.glob(os.path.join(util.ROOT, "positionbaseddynamics_amd", "_lib", "libpbdx*.so"))
    assert libs
    for lib in libs:
        syms = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
        assert "hipHostRegister" not in syms and "hipHostUnregister" not in syms, lib
    csrc = os.path.join(util.ROOT, "positionbaseddynamics_amd", "csrc")
    own = re.compile(r"hipMemcpy(Async)?\((mir\b|mir \+|j\.dst, slot|st \+ per \* first, slot|c, s->d_(tet|contact|dyn)_counters|status, d_status|out, s->d_tet_counters, 8 \*|c->d_buf, c->h_mir,|c->h_mir \+)")
    bad = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.h"))):
        if os.path.basename(path) == "pbdx_hostio.hip":
            continue
        for ln, line in enumerate(open(path), 1):
            if re.search(r"hipMemcpy(Async)?\(", line) and "DeviceToDevice" not in line and not own.search(line):
                bad.append("%s:%d: %s" % (os.path.basename(path), ln, line.strip()))
            if "hipHostRegister(" in line:
                bad.append("%s:%d: %s" % (os.path.basename(path), ln, line.strip()))
    assert not bad, "\n".join(bad)
    plug = os.path.join(util.ROOT, "positionbaseddynamics_amd", "plugin")
    for path in glob.glob(os.path.join(plug, "*.cpp")) + glob.glob(os.path.join(plug, "*.h")):
        text = open(path).read()
        assert "hipMemcpy" not in text and "hipHostRegister(" not in text, path
