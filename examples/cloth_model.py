#!/usr/bin/env python3
"""pyPBD/examples/cloth_model.py of the reference, headless, running on the MI355X engine.

The scene-building and stepping code below is the reference example's code with one change: the import
(`import positionbaseddynamics_amd as pbd` instead of `import pypbd as pbd`); the pygame / OpenGL viewer
is replaced by a printed summary.  buildModel / createMesh / timeStep keep the reference's calls line by
line (pyPBD/examples/cloth_model.py:18-110)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import positionbaseddynamics_amd as pbd  # noqa: E402

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


def buildModel(simModel=2, bendingModel=2):
    sim = pbd.Simulation.getCurrent()
    sim.initDefault()
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


def reset():
    pbd.Simulation.getCurrent().reset()
    pbd.Simulation.getCurrent().getModel().cleanup()
    buildModel()


def main(frames=10, simModel=2, bendingModel=2):
    pbd.Logger.addConsoleSink(pbd.LogLevel.INFO)
    buildModel(simModel, bendingModel)
    for frame in range(frames):
        timeStep()
    x = pbd.Simulation.getCurrent().getModel().getParticles().getVertices()
    print("Time: {:.2f}".format(pbd.TimeManager.getCurrent().getTime()))
    print("bounding box: %s .. %s" % (x.min(axis=0), x.max(axis=0)))
    return np.array(x, copy=True)


if __name__ == "__main__":
    main()
