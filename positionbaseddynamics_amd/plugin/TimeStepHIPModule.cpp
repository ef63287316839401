// TimeStepHIPModule -- the ONE binding the drop-in adds to the reference's python package `pypbd`.
//
// Written in the style of pyPBD/*Module.cpp (one `void XModule(py::module)` per file, called from PYBIND11_MODULE(pypbd, m)
// in pyPBD/main.cpp:25-37).  A maintainer adds this file to pyPBD/CMakeLists.txt and two lines to pyPBD/main.cpp:
//
//     void TimeStepHIPModule(py::module);          // next to the other declarations, main.cpp:14-23
//     TimeStepHIPModule(m);                        // after TimeStepModule(m), main.cpp:34 (the base class must be registered first)
//
// Here the reference tree is read-only and its sources are compiled where they lie, UNMODIFIED (plugin/Makefile: every
// pyPBD/*.cpp, plus Simulation/CubicSDFCollisionDetection.cpp against the compile-only Discregrid shim), so main.cpp cannot
// receive those two lines.  The linker supplies them instead: the module is linked with
// `-Wl,--wrap=_Z14TimeStepModuleN8pybind117module_E`, which turns main.cpp's call `TimeStepModule(m)` into a call of
// __wrap_... below; that runs the reference's own TimeStepModule (__real_...) and then TimeStepHIPModule -- exactly the call
// order the two added lines produce.
//
// Ownership.  The reference's convention for a custom time step is "the Simulation owns it"
// (Demos/PositionBasedElasticRodsDemo/PositionBasedElasticRodsDemo.cpp:51-54: delete the old one, setTimeStep(new ...);
// Simulation::~Simulation deletes m_timeStep, Simulation.cpp:22-28), and pyPBD binds Simulation::setTimeStep as it is
// (pyPBD/SimulationModule.cpp:25: a raw pointer changes hands, no keep_alive).  A python-constructed object that python
// also deleted would leave the Simulation with a dangling pointer as soon as the script drops its variable, so the
// constructor below creates the C++ object WITHOUT making the python wrapper its owner (the wrapper behaves like the
// `reference` return policy): whoever receives it through setTimeStep owns it, as in the C++ demos.
#include "TimeStepControllerHIP.h"

#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <memory>

namespace py = pybind11;

void TimeStepHIPModule(py::module m_sub)
{
	py::class_<PBD::TimeStepControllerHIP, PBD::TimeStepController>(m_sub, "TimeStepControllerHIP")
		.def("__init__", [](py::detail::value_and_holder &v_h, int device)
			{
				v_h.value_ptr() = new PBD::TimeStepControllerHIP(device);
				std::unique_ptr<PBD::TimeStepControllerHIP> nobody;         // an EMPTY holder: the wrapper is registered, python deletes nothing --
				v_h.type->init_instance(v_h.inst, &nobody);                // the Simulation the object is handed to owns it (see above)
			}, py::detail::is_new_style_constructor(), py::arg("device") = 0)
		.def("stepResident", &PBD::TimeStepControllerHIP::stepResident, py::arg("model"), py::arg("numSteps") = 1)
		.def("syncToHost", &PBD::TimeStepControllerHIP::syncToHost)
		.def("syncFromHost", &PBD::TimeStepControllerHIP::syncFromHost)
		.def("markHostDirty", &PBD::TimeStepControllerHIP::markHostDirty)
		.def("deviceAhead", &PBD::TimeStepControllerHIP::deviceAhead)
		.def("invalidate", &PBD::TimeStepControllerHIP::invalidate)
		.def("refreshParameters", &PBD::TimeStepControllerHIP::refreshParameters)
		.def("setFullParameterScan", &PBD::TimeStepControllerHIP::setFullParameterScan)
		.def("setAllowReferenceFallback", &PBD::TimeStepControllerHIP::setAllowReferenceFallback)
		.def("numGpuSteps", &PBD::TimeStepControllerHIP::numGpuSteps)
		.def("numFallbackSteps", &PBD::TimeStepControllerHIP::numFallbackSteps)
		.def("numFailedSteps", &PBD::TimeStepControllerHIP::numFailedSteps)
		.def("numParameterRefreshes", &PBD::TimeStepControllerHIP::numParameterRefreshes)
		.def("numScheduleBuilds", &PBD::TimeStepControllerHIP::numScheduleBuilds)
		.def("numUploads", &PBD::TimeStepControllerHIP::numUploads)
		.def("numPartialUploads", &PBD::TimeStepControllerHIP::numPartialUploads);
}

// ---- stand-in for the two lines in pyPBD/main.cpp (see the header comment) ---------------------------------------------
void TimeStepModule(py::module);                                                                 // pyPBD/TimeStepModule.cpp:12
extern "C" void __real__Z14TimeStepModuleN8pybind117module_E(py::module);
extern "C" void __wrap__Z14TimeStepModuleN8pybind117module_E(py::module m)
{
	__real__Z14TimeStepModuleN8pybind117module_E(m);
	TimeStepHIPModule(m);
}
