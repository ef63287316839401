// TimeStepControllerHIP -- reference-side binding of the MI355X engine.
//
// A drop-in PBD::TimeStep plug-in for InteractiveComputerGraphics/PositionBasedDynamics: it
// derives from the reference's own PBD::TimeStepController (Simulation/TimeStepController.h), so
// it inherits the GenericParameters ids / names ("subSteps", "maxIterations", "maxIterationsV",
// "velocityUpdateMethod") that json scenes and pypbd use, and overrides step() to run the
// particle part of the time step (TimeStepController.cpp:75-176, 251-295) on the GPU through
// the C ABI of libpbdx (include/pbdx.h).  Install it exactly like the reference installs a
// custom time step (Demos/PositionBasedElasticRodsDemo/PositionBasedElasticRodsDemo.cpp:51-54):
//
//     delete Simulation::getCurrent()->getTimeStep();
//     Simulation::getCurrent()->setTimeStep(new PBD::TimeStepControllerHIP());
//     Simulation::getCurrent()->getTimeStep()->init();
//
// Scope: models whose constraints are all particle constraints known to the engine.  Rigid bodies are
// accepted when they are all static (mass 0) colliders of a DistanceFieldCollisionDetection with analytic
// distance fields (box, sphere, torus, cylinder, hollow sphere / box): the particle vs rigid body contacts
// are then detected and solved on the GPU as well.  No dynamic rigid bodies, joints, orientations.  There is NO silent CPU path: a step the
// engine cannot run (no HIP device, HIP error, unsupported model) logs an error, leaves the
// model untouched and is counted in numFailedSteps().  A host application that prefers the
// reference's own CPU TimeStepController for such models opts in explicitly with
// setAllowReferenceFallback(true); those steps are counted in numFallbackSteps().
// This file needs the reference's headers to compile; it is built only where the reference
// tree is available (positionbaseddynamics_amd/plugin/Makefile).
#ifndef __TimeStepControllerHIP_h__
#define __TimeStepControllerHIP_h__

#include "Simulation/TimeStepController.h"
#include "../../include/pbdx.h"
#include <vector>

namespace PBD
{
	class TimeStepControllerHIP : public TimeStepController
	{
	public:
		TimeStepControllerHIP(int device = 0);
		virtual ~TimeStepControllerHIP();

		virtual void step(SimulationModel &model);
		virtual void reset();

		/** Drop the device image (call after editing constraint parameters in place). */
		void invalidate() { m_scheduleValid = false; }
		/** Number of steps that ran on the GPU / fell back to the reference's CPU path. */
		unsigned int numGpuSteps() const { return m_gpuSteps; }
		unsigned int numFallbackSteps() const { return m_fallbackSteps; }
		unsigned int numFailedSteps() const { return m_failedSteps; }
		/** Opt in to running unsupported models / failed steps on the reference's CPU path (default off). */
		void setAllowReferenceFallback(bool b) { m_allowFallback = b; }
		pbdx_solver *solver() { return m_solver; }

	protected:
		bool supported(SimulationModel &model) const;
		bool buildSchedule(SimulationModel &model);
		bool uploadParticles(SimulationModel &model);
		bool uploadColliders(SimulationModel &model);
		bool downloadParticles(SimulationModel &model);

		pbdx_solver *m_solver;
		int m_device;
		bool m_scheduleValid;
		size_t m_numConstraints;
		unsigned int m_numParticles;
		unsigned int m_gpuSteps, m_fallbackSteps, m_failedSteps;
		bool m_allowFallback;
		void refuse(SimulationModel &model, const char *why);
		std::vector<float> m_x, m_v, m_old, m_last, m_mass, m_invMass;
	};
}

/** Factory with C linkage so that a host application can dlopen() the plug-in. */
extern "C" PBD::TimeStep *pbdx_create_timestep_hip();

#endif
