#!/usr/bin/env python3
"""pyPBD/examples/beam_model.py of the reference through the reference's OWN python package, on the MI355X engine.

`import pypbd as pbd` = positionbaseddynamics_amd/plugin/_build/pypbd*.so (pyPBD/*.cpp compiled unmodified + the one
added class `pbd.TimeStepControllerHIP`, see cloth_model_pypbd.py).  Scene building and stepping are the reference
example's (pyPBD/examples/beam_model.py:15-87: a 30x5x5 tet beam, first slab pinned, addSolidConstraints; viewer removed).
    python examples/beam_model_pypbd.py [--cpu] [simModel]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "positionbaseddynamics_amd", "plugin", "_build"))
import pypbd as pbd  # noqa: E402

width = 30
depth = 5
height = 5

_keep = []


def installTimeStep(gpu):
    sim = pbd.Simulation.getCurrent()
    cd = sim.getTimeStep().getCollisionDetection()
    ts = pbd.TimeStepControllerHIP() if gpu else pbd.TimeStepController()      # <-- the drop-in
    ts.init()                                                                  # <--
    ts.setCollisionDetection(sim.getModel(), cd)                               # <--
    sim.setTimeStep(ts)                                                        # <--
    if not gpu:
        _keep.append(ts)


def buildModel(simModel=3, gpu=True):
    sim = pbd.Simulation.getCurrent()
    sim.initDefault()
    installTimeStep(gpu)
    # 1 = distance constraints (PBD)      2 = FEM tet constraints (PBD)     3 = FEM tet constraints (XPBD)
    # 4 = strain tet constraints (PBD)    5 = shape matching                6 = distance constraints (XPBD)
    createMesh(simModel)
    ts = sim.getTimeStep()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 3)


def createMesh(simModel):
    sim = pbd.Simulation.getCurrent()
    model = sim.getModel()
    tetModel = model.addRegularTetModel(width, height, depth, translation=[5, 0, 0], scale=[10, 1.5, 1.5])
    pd = model.getParticles()
    for i in range(1):
        for j in range(height):
            for k in range(depth):
                pd.setMass(i * height * depth + j * depth + k, 0.0)
    stiffness = 1.0
    if simModel == 3:
        stiffness = 1000000
    if simModel == 6:
        stiffness = 100000
    poissonRatio = 0.3
    model.addSolidConstraints(tetModel, simModel, stiffness, poissonRatio, stiffness, False, False)
    tetModel.updateMeshNormals(pd)
    print("Number of tets: " + str(tetModel.getParticleMesh().numTets()))
    print("Number of vertices: " + str(width * height * depth))


def timeStep():
    sim = pbd.Simulation.getCurrent()
    model = sim.getModel()
    for i in range(8):
        sim.getTimeStep().step(model)
    for tetModel in model.getTetModels():
        tetModel.updateMeshNormals(model.getParticles())


def main(frames=3, simModel=3, gpu=True):
    buildModel(simModel, gpu)
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
    args = [a for a in sys.argv[1:] if a != "--cpu"]
    main(simModel=int(args[0]) if args else 3, gpu="--cpu" not in sys.argv)
