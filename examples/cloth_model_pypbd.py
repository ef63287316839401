#!/usr/bin/env python3
"""pyPBD/examples/cloth_model.py of the reference through the reference's OWN python package, on the MI355X engine.

`import pypbd as pbd` here is positionbaseddynamics_amd/plugin/_build/pypbd*.so: the reference's pyPBD/*.cpp compiled
UNMODIFIED (plugin/Makefile; Discregrid replaced by a compile-only shim) plus the ONE class the drop-in adds,
`pbd.TimeStepControllerHIP` (plugin/TimeStepHIPModule.cpp).  The scene-building and stepping code is the reference
example's (pyPBD/examples/cloth_model.py:18-110; the pygame / OpenGL viewer removed); the marked lines install the GPU time
step the way the reference installs a custom time step (Demos/PositionBasedElasticRodsDemo/PositionBasedElasticRodsDemo.cpp:51-54).
    python examples/cloth_model_pypbd.py [--cpu]      # --cpu: the reference's own TimeStepController
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "positionbaseddynamics_amd", "plugin", "_build"))
import pypbd as pbd  # noqa: E402

nRows = 50
nCols = 50
width = 10.0
height = 10.0


def rotation_matrix(angle, axis):
    """math_tools.rotation_matrix of the reference examples (axis-angle -> 3x3)."""
    x, y, z = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                     [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                     [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])


_keep = []


def installTimeStep(gpu):
    sim = pbd.Simulation.getCurrent()
    cd = sim.getTimeStep().getCollisionDetection()      # initDefault() gave the default time step a collision detection
    ts = pbd.TimeStepControllerHIP() if gpu else pbd.TimeStepController()      # <-- the drop-in: the class name ...
    ts.init()                                                                  # <--
    ts.setCollisionDetection(sim.getModel(), cd)                               # <-- (what initDefault did for the old one)
    sim.setTimeStep(ts)                                                        # <-- ... installed like any custom time step
    if not gpu:
        _keep.append(ts)      # pypbd's own TimeStepController is python-owned (pyPBD/TimeStepModule.cpp:30): keep it alive


def buildModel(simModel=2, bendingModel=2, gpu=True):
    sim = pbd.Simulation.getCurrent()
    sim.initDefault()
    installTimeStep(gpu)
    createMesh(simModel, bendingModel)
    ts = sim.getTimeStep()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 3)


def createMesh(simModel, bendingModel):
    sim = pbd.Simulation.getCurrent()
    model = sim.getModel()
    triModel = model.addRegularTriangleModel(nCols, nRows,
                                             [0, 0, 0],
                                             rotation_matrix(math.pi * 0.5, [1.0, 0.0, 0.0]),
                                             [width, height],
                                             testMesh=False)
    pd = model.getParticles()
    pd.setMass(0, 0.0)
    pd.setMass(nRows - 1, 0.0)
    stiffness = 1.0
    if simModel == 4:
        stiffness = 100000
    poissonRatio = 0.3
    model.addClothConstraints(triModel, simModel, stiffness, stiffness, stiffness, stiffness,
                              poissonRatio, poissonRatio, False, False)
    bending_stiffness = 0.01
    if bendingModel == 3:
        bending_stiffness = 50.0
    model.addBendingConstraints(triModel, bendingModel, bending_stiffness)
    print("Number of triangles: " + str(triModel.getParticleMesh().numFaces()))
    print("Number of vertices: " + str(nRows * nCols))


def timeStep():
    sim = pbd.Simulation.getCurrent()
    model = sim.getModel()
    for i in range(8):
        sim.getTimeStep().step(model)
    for triModel in model.getTriangleModels():
        triModel.updateMeshNormals(model.getParticles())


def main(frames=3, simModel=2, bendingModel=2, gpu=True):
    buildModel(simModel, bendingModel, gpu)
    for frame in range(frames):
        timeStep()
    sim = pbd.Simulation.getCurrent()
    x = np.array(sim.getModel().getParticles().getVertices(), copy=True)
    ts = sim.getTimeStep()
    print("Time: {:.2f}".format(pbd.TimeManager.getCurrent().getTime()))
    print("bounding box: %s .. %s" % (x.min(axis=0), x.max(axis=0)))
    if gpu:
        print("steps on the GPU: %d, refused: %d, on the reference CPU path: %d" % (ts.numGpuSteps(), ts.numFailedSteps(), ts.numFallbackSteps()))
    return x


if __name__ == "__main__":
    main(gpu="--cpu" not in sys.argv)
