// pypbd (REDUCED) + TimeStepControllerHIP -- the python side of the drop-in.
//
// The reference's python package `pypbd` (pyPBD/*.cpp) cannot be built in this container: its SimulationModule pulls in
// CubicSDFCollisionDetection -> Discregrid, which is fetched at configure time and is not under /root/reference.  This file
// is an own, minimal pybind11 module over the SAME unmodified reference classes (linked from oracle/_ref/libpbdref_*.so,
// the stand-in for the reference's libraries) that registers
//   * the classes a cloth / solid script on the path touches, under pypbd's names and call signatures
//     (pyPBD/SimulationModule.cpp:14-36, SimulationModelModule.cpp:90-383, ParticleDataModule.cpp:34-58,
//      TimeStepModule.cpp:14-30, ParameterObjectModule.cpp:19-29, TimeModule), and
//   * the ONE class the drop-in adds to pypbd: `TimeStepControllerHIP` (derived from TimeStepController; constructible;
//     accepted by Simulation.setTimeStep exactly like the reference's own class) with its device-resident extensions.
// In a full build of the reference the five lines that register TimeStepControllerHIP are all that is added to
// pyPBD/TimeStepModule.cpp; everything else here exists only because pypbd itself cannot be compiled in this container.
// Differences to pypbd: initDefault() installs no collision detection (pypbd installs the Discregrid-based one).
#include "TimeStepControllerHIP.h"
#include "Simulation/Simulation.h"
#include "Simulation/SimulationModel.h"
#include "Simulation/TimeManager.h"
#include "Simulation/TimeStepController.h"

#include <pybind11/pybind11.h>
#include <pybind11/eigen.h>
#include <pybind11/numpy.h>
#include <pybind11/stl.h>

namespace py = pybind11;
using namespace PBD;

PYBIND11_MODULE(pypbd, m)
{
	m.doc() = "reduced pypbd (classes on the PBD/XPBD solver path) + TimeStepControllerHIP (MI355X engine)";

	py::class_<GenParam::ParameterObject>(m, "ParameterObject")
		.def("getValueBool", &GenParam::ParameterObject::getValue<bool>)
		.def("getValueInt", &GenParam::ParameterObject::getValue<int>)
		.def("getValueUInt", &GenParam::ParameterObject::getValue<unsigned int>)
		.def("getValueFloat", &GenParam::ParameterObject::getValue<Real>)
		.def("setValueBool", &GenParam::ParameterObject::setValue<bool>)
		.def("setValueInt", &GenParam::ParameterObject::setValue<int>)
		.def("setValueUInt", &GenParam::ParameterObject::setValue<unsigned int>)
		.def("setValueFloat", &GenParam::ParameterObject::setValue<Real>);

	py::class_<TimeManager>(m, "TimeManager")
		.def_static("getCurrent", &TimeManager::getCurrent, py::return_value_policy::reference)
		.def("getTime", &TimeManager::getTime)
		.def("setTime", &TimeManager::setTime)
		.def("getTimeStepSize", &TimeManager::getTimeStepSize)
		.def("setTimeStepSize", &TimeManager::setTimeStepSize);

	py::class_<ParticleData>(m, "ParticleData")
		.def("addVertex", &ParticleData::addVertex)
		.def("getPosition", (const Vector3r &(ParticleData::*)(const unsigned int) const)(&ParticleData::getPosition))
		.def("setPosition", &ParticleData::setPosition)
		.def("getPosition0", (const Vector3r &(ParticleData::*)(const unsigned int) const)(&ParticleData::getPosition0))
		.def("getMass", (const Real (ParticleData::*)(const unsigned int) const)(&ParticleData::getMass))
		.def("getInvMass", &ParticleData::getInvMass)
		.def("setMass", &ParticleData::setMass)
		.def("getVelocity", (const Vector3r &(ParticleData::*)(const unsigned int) const)(&ParticleData::getVelocity))
		.def("setVelocity", &ParticleData::setVelocity)
		.def("getNumberOfParticles", &ParticleData::getNumberOfParticles)
		.def("size", &ParticleData::size)
		// pyPBD/ParticleDataModule.cpp:54-58: the packed position array as a numpy view (no copy)
		.def("getVertices", [](ParticleData &pd) {
			return py::array_t<Real>({ (py::ssize_t)pd.size(), (py::ssize_t)3 }, { (py::ssize_t)(3 * sizeof(Real)), (py::ssize_t)sizeof(Real) },
				pd.size() ? &pd.getPosition(0)[0] : nullptr, py::cast(&pd));
		});

	py::class_<Utilities::IndexedFaceMesh>(m, "IndexedFaceMesh")
		.def("numVertices", &Utilities::IndexedFaceMesh::numVertices)
		.def("numFaces", &Utilities::IndexedFaceMesh::numFaces)
		.def("numEdges", &Utilities::IndexedFaceMesh::numEdges);

	py::class_<TriangleModel>(m, "TriangleModel")
		.def("getIndexOffset", &TriangleModel::getIndexOffset)
		.def("getParticleMesh", (TriangleModel::ParticleMesh &(TriangleModel::*)())(&TriangleModel::getParticleMesh), py::return_value_policy::reference_internal)
		.def("updateMeshNormals", &TriangleModel::updateMeshNormals);

	py::class_<SimulationModel, GenParam::ParameterObject>(m, "SimulationModel")
		.def(py::init<>())
		.def("init", &SimulationModel::init)
		.def("reset", &SimulationModel::reset)
		.def("cleanup", &SimulationModel::cleanup)
		.def("getParticles", &SimulationModel::getParticles, py::return_value_policy::reference_internal)
		.def("getTriangleModels", &SimulationModel::getTriangleModels, py::return_value_policy::reference_internal)
		.def("addRegularTriangleModel", [](SimulationModel &model, const int width, const int height, const Vector3r &translation,
			const Matrix3r &rotation, const Vector2r &scale, const bool) {
				auto &triModels = model.getTriangleModels();
				const size_t i = triModels.size();
				model.addRegularTriangleModel(width, height, translation, rotation, scale);
				return triModels[i];
			}, py::arg("width"), py::arg("height"), py::arg("translation") = Vector3r::Zero(), py::arg("rotation") = Matrix3r::Identity(),
			py::arg("scale") = Vector2r::Ones(), py::arg("testMesh") = false, py::return_value_policy::reference)
		.def("addClothConstraints", &SimulationModel::addClothConstraints)
		.def("addBendingConstraints", &SimulationModel::addBendingConstraints)
		.def("numConstraints", [](SimulationModel &model) { return model.getConstraints().size(); });

	// (py::nodelete on the time step classes: Simulation::setTimeStep takes ownership and ~Simulation deletes its time step,
	// Simulation.cpp:22-28 -- the pattern of Demos/PositionBasedElasticRodsDemo/PositionBasedElasticRodsDemo.cpp:51-54)
	py::class_<TimeStep, GenParam::ParameterObject, std::unique_ptr<TimeStep, py::nodelete>>(m, "TimeStep")
		.def("step", &TimeStep::step)
		.def("reset", &TimeStep::reset)
		.def("init", &TimeStep::init);

	py::class_<TimeStepController, TimeStep, std::unique_ptr<TimeStepController, py::nodelete>>(m, "TimeStepController")
		.def_readwrite_static("NUM_SUB_STEPS", &TimeStepController::NUM_SUB_STEPS)
		.def_readwrite_static("MAX_ITERATIONS", &TimeStepController::MAX_ITERATIONS)
		.def_readwrite_static("MAX_ITERATIONS_V", &TimeStepController::MAX_ITERATIONS_V)
		.def_readwrite_static("VELOCITY_UPDATE_METHOD", &TimeStepController::VELOCITY_UPDATE_METHOD)
		.def_readwrite_static("ENUM_VUPDATE_FIRST_ORDER", &TimeStepController::ENUM_VUPDATE_FIRST_ORDER)
		.def_readwrite_static("ENUM_VUPDATE_SECOND_ORDER", &TimeStepController::ENUM_VUPDATE_SECOND_ORDER)
		.def(py::init<>());

	// ---- the addition to pypbd ---------------------------------------------------------------------------------------
	py::class_<TimeStepControllerHIP, TimeStepController, std::unique_ptr<TimeStepControllerHIP, py::nodelete>>(m, "TimeStepControllerHIP")
		.def(py::init<int>(), py::arg("device") = 0)
		.def("stepResident", &TimeStepControllerHIP::stepResident, py::arg("model"), py::arg("numSteps") = 1)
		.def("syncToHost", &TimeStepControllerHIP::syncToHost)
		.def("syncFromHost", &TimeStepControllerHIP::syncFromHost)
		.def("markHostDirty", &TimeStepControllerHIP::markHostDirty)
		.def("deviceAhead", &TimeStepControllerHIP::deviceAhead)
		.def("invalidate", &TimeStepControllerHIP::invalidate)
		.def("refreshParameters", &TimeStepControllerHIP::refreshParameters)
		.def("setFullParameterScan", &TimeStepControllerHIP::setFullParameterScan)
		.def("numPartialUploads", &TimeStepControllerHIP::numPartialUploads)
		.def("setAllowReferenceFallback", &TimeStepControllerHIP::setAllowReferenceFallback)
		.def("numGpuSteps", &TimeStepControllerHIP::numGpuSteps)
		.def("numFallbackSteps", &TimeStepControllerHIP::numFallbackSteps)
		.def("numFailedSteps", &TimeStepControllerHIP::numFailedSteps);

	py::class_<Simulation, GenParam::ParameterObject>(m, "Simulation")
		.def(py::init<>())
		.def_static("getCurrent", &Simulation::getCurrent, py::return_value_policy::reference)
		.def_static("setCurrent", &Simulation::setCurrent)
		.def_static("hasCurrent", &Simulation::hasCurrent)
		.def("init", &Simulation::init)
		.def("reset", &Simulation::reset)
		.def("getModel", &Simulation::getModel, py::return_value_policy::reference_internal)
		.def("setModel", &Simulation::setModel)
		.def("getTimeStep", &Simulation::getTimeStep, py::return_value_policy::reference_internal)
		.def("setTimeStep", [](Simulation &sim, TimeStep *ts) {
			// the reference's pattern: the simulation owns its time step; replacing it deletes the old one
			if (sim.getTimeStep() != ts) delete sim.getTimeStep();
			sim.setTimeStep(ts);
		})
		.def("initDefault", [](Simulation &sim) {
			sim.setModel(new SimulationModel());
			sim.getModel()->init();
		});
}
