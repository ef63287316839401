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
// Two ways to step:
//   step(model)               TimeStep::step's contract: host ParticleData in -> one step on the device -> host
//                             ParticleData out.  The ParticleData arrays (std::vector<Vector3r>, packed Real[3],
//                             ParticleData.h:91-100) are handed to the engine as they are -- no per-element
//                             conversion loops; a double host is converted on the device.
//   stepResident(model, n)    SURVEY 8f rank 1: n steps with the state resident in HBM; nothing is downloaded until
//                             syncToHost(model).  Contract: after stepResident the host arrays are STALE until syncToHost.
//                             Host writes are found by full-coverage block hashes of the arrays (every word is hashed,
//                             PBDX_HASH_BLOCK elements per hash; no sampling): after a syncToHost / step() the changed
//                             blocks are uploaded; while the host is stale a written ARRAY replaces the device's as a whole
//                             (the others are kept) -- to edit single particles, syncToHost first.
// Masses: ParticleData::setMass between steps (pinning, mouse attach) is found the same way; the inverse masses of the changed
// blocks are re-derived.  Constraint parameters: before every step EVERY parameter record of EVERY constraint is compared (block
// hashes, multi-threaded walk) with what the device image was built from, so bulk edits (setClothStiffness ...) and an edit of a
// single constraint's field (python: `constraint.stiffness = ...`) are both found without any announcement; a change refreshes
// only the parameter streams.  The walk costs host time proportional to the number of constraints (about 1 ms per million on a
// many-core host); a host that steps multi-million-constraint models one step() at a time and never edits single constraints can
// opt in to a sampled check with setFullParameterScan(false) -- or amortise the walk with stepResident(model, n).
//
// Scope: models whose constraints are all particle constraints known to the engine; a pure particle model that ALSO holds constraint
// classes the engine does not know (PositionBasedGenericConstraints.h, user subclasses) runs as a MIXED model: known buckets on the GPU,
// the others through the reference's own virtual call on the host inside the same colour groups (numMixedGroups(); exact, slow).  Rigid bodies are
// accepted when they are all static (mass 0) colliders of a DistanceFieldCollisionDetection with analytic
// distance fields (box, sphere, torus, cylinder, hollow sphere / box): the particle vs rigid body contacts
// are then detected and solved on the GPU as well; so are the contacts between tet models that carry such a distance field
// (ParticleTetContactConstraint; friction 0 only).  No dynamic rigid bodies, joints, orientations.  There is NO silent CPU path: a step the
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
#include <stdint.h>
#include <vector>
#include <string>

namespace PBD
{
	class TimeStepControllerHIP : public TimeStepController
	{
	public:
		TimeStepControllerHIP(int device = 0);
		virtual ~TimeStepControllerHIP();

		virtual void step(SimulationModel &model);
		virtual void reset();

		/** n steps with the particle state resident on the device (no download).  Returns false (and counts a failed step)
		 * if the engine cannot run the model.  Host-visible side effects that cost a pass over all particles
		 * (clearAccelerations) are applied by syncToHost. */
		bool stepResident(SimulationModel &model, unsigned int numSteps = 1);
		/** device state -> ParticleData (x, v, oldX, lastX); accelerations as TimeStep::clearAccelerations leaves them */
		bool syncToHost(SimulationModel &model);
		/** ParticleData -> device, unconditionally (the host is authoritative) */
		bool syncFromHost(SimulationModel &model);
		/** "everything on the host is newer": the next step uploads all arrays without looking.  Not needed for correctness --
		 * host writes are found by full-coverage block hashes (every word of x, v, oldX, lastX, masses is looked at before every
		 * step) -- only to override the device-ahead merge rule below. */
		void markHostDirty() { m_hostDirty = true; m_accelValid = false; }
		/** transfers of step() through the engine's page-locked mirror (PBDX_OPT_PIN_HOST; default on) */
		void setPinHostArrays(bool b) { if (m_solver) pbdx_solver_set_option(m_solver, PBDX_OPT_PIN_HOST, b ? 1 : 0); }
		/** the device holds a newer state than ParticleData */
		bool deviceAhead() const { return m_deviceAhead; }

		/** Drop the device image (call after editing the topology behind the model's back). */
		void invalidate() { m_scheduleValid = false; }
		/** Run-time parameter edits (setClothStiffness, m_stiffness / m_restLength edits, SimulationModel.h
		 * setConstraintValue<>) are found by the exact parameter scan before every step and refresh ONLY the parameter streams (no
		 * replanning).  refreshParameters() forces such a refresh; it is needed only after setFullParameterScan(false), when an
		 * edit of a single constraint in a large model can fall between the samples. */
		void refreshParameters() { m_paramsDirty = true; }
		/** Number of steps that ran on the GPU / fell back to the reference's CPU path. */
		unsigned int numGpuSteps() const { return m_gpuSteps; }
		unsigned int numFallbackSteps() const { return m_fallbackSteps; }
		unsigned int numFailedSteps() const { return m_failedSteps; }
		unsigned int numParameterRefreshes() const { return m_paramRefreshes; }
		/** step() runs the device step and the download SPECULATIVELY while the exact parameter scan still walks the constraints on the host's worker
		 * threads (the scan costs several times the step at millions of constraints); an edit found by the scan undoes the step on the device
		 * (pbdx_solver_restore_state), refreshes the parameter streams and repeats it.  Counters: steps taken that way / steps that had to be repeated. */
		unsigned int numSpeculativeSteps() const { return m_speculativeSteps; }
		unsigned int numRepeatedSteps() const { return m_repeatedSteps; }
		void setSpeculativeStep(bool b) { m_speculate = b; }
		/** Test hook: the repeat of a speculative step (after the scan found an edit) is made to fail once the checkpoint is back on the device, so
		 * that the tests can check what the host holds then (the pre-step state, not the stale speculative result). */
		void setFailRepeatForTest(bool b) { m_failRepeatForTest = b; }
		unsigned int numScheduleBuilds() const { return m_scheduleBuilds; }
		unsigned int numUploads() const { return m_uploads; }
		/** uploads of only the blocks the host wrote (host current, full-coverage block hashes) */
		unsigned int numPartialUploads() const { return m_partialUploads; }
		/** true (DEFAULT): every parameter record is compared before every step (exact).  false: a strided sample of ~4096 records
		 * (sees bulk edits; single-constraint edits then need refreshParameters()). */
		void setFullParameterScan(bool b) { m_fullParameterScan = b; }
		/** Mixed models: colour groups of the current schedule that also hold constraints of classes the engine does not know (run on the
		 * host through the reference's own solvePositionConstraint, one round trip of the positions per such group and iteration); 0 for
		 * a model the engine runs alone. */
		unsigned int numMixedGroups() const { unsigned int k = 0; for (const std::vector<unsigned int> &v : m_hostGroups) if (!v.empty()) k++; return m_mixed ? k : 0u; }
		/** Opt in to running unsupported models / failed steps on the reference's CPU path (default off). */
		void setAllowReferenceFallback(bool b) { m_allowFallback = b; }
		/** accumulated host milliseconds of step(): [0] block hashes of the host arrays, [1] full uploads, [2] parameter check, [3] collider refresh, [4] engine step (host wall clock), [5] download, [6] the engine steps' device-event time */
		void timing(double out[7], bool reset) { for (int k = 0; k < 6; k++) { out[k] = m_ms[k]; if (reset) m_ms[k] = 0.0; } out[6] = m_deviceMs; if (reset) m_deviceMs = 0.0; }
		pbdx_solver *solver() { return m_solver; }

	protected:
		bool supported(SimulationModel &model);
		bool buildSchedule(SimulationModel &model, bool paramsOnly);
		bool uploadParticles(SimulationModel &model);
		bool uploadColliders(SimulationModel &model);
		// dynamic rigid bodies (finite mass) as impulse sinks of the particle contacts: their time integration stays here, on the host
		// (TimeStepController.cpp:94-104,137-152,178-186), the contact velocity solve runs on the device in the reference's contact order
		void integrateBodies(SimulationModel &model);
		bool applyBodyVelocities(SimulationModel &model);
		bool uploadTetColliders(SimulationModel &model, float tolerance);
		bool downloadParticles(SimulationModel &model);
		bool prepare(SimulationModel &model, bool forceUpload, bool *scanDeferred = NULL);
		bool stepRaw(unsigned int numSteps);
		void finishSteps(unsigned int numSteps);
		void readGlobals();
		float m_rawH, m_rawG[3];
		std::string m_helperError;
		bool runSteps(SimulationModel &model, unsigned int numSteps);
		bool runMixedSteps(SimulationModel &model, unsigned int numSteps, const float g[3]);
		bool uploadChanges(SimulationModel &model, std::vector<uint64_t> now[5]);
		void hashParameters(SimulationModel &model, std::vector<uint64_t> &out) const;
		void hashHostState(SimulationModel &model, std::vector<uint64_t> out[5]) const;
		void refuse(SimulationModel &model, const char *why);
		void refreshAccelerations(SimulationModel &model);

		pbdx_solver *m_solver;
		int m_device;
		bool m_scheduleValid;
		size_t m_numConstraints;
		unsigned int m_numParticles;
		unsigned int m_gpuSteps, m_fallbackSteps, m_failedSteps, m_paramRefreshes, m_scheduleBuilds, m_uploads;
		bool m_allowFallback;
		// device-resident state
		bool m_deviceAhead;            // device state newer than ParticleData
		bool m_hostDirty;              // markHostDirty()
		bool m_imageValid;             // particles were uploaded at least once for the current model
		std::vector<uint64_t> m_blockHash[5]; // full-coverage block hashes (include/pbdx.h) of x, v, oldX, lastX, masses as of the last upload / download
		unsigned int m_partialUploads;
		double m_deviceMs;             // device-event time of the engine's steps (pbdx_step_stats.total_ms), to tell GPU time from host-side waiting
		double m_ms[6];                // host milliseconds spent in: hashing the host arrays, full uploads, the parameter check, the collider refresh, the engine's step, the download
		bool m_fullParameterScan;
		bool m_speculate;
		bool m_failRepeatForTest;
		unsigned int m_speculativeSteps, m_repeatedSteps;
		// parameters
		bool m_paramsDirty;
		std::vector<uint64_t> m_paramHash;   // exact scan: one hash per block of constraints (+ the count); sampled scan: one hash
		// what `supported` was last evaluated for
		const void *m_supportedFor; size_t m_supportedConstraints, m_supportedBodies, m_supportedObjects; bool m_supported;
		bool m_accelValid; Real m_accelGravity[3];
		uint64_t m_tetSignature;       // which set of deformable colliders the engine holds (0 = none)
		std::vector<Real> m_invMass;
		// mixed models: per colour group the constraints that stay with the host
		bool m_mixed; unsigned int m_mixedGroupsLast;
		bool m_dynamicBodies;              // some rigid body has a finite mass
		std::vector<unsigned int> m_colliderBody;      // collider -> rigid body index (uploadColliders)
		std::vector<uint32_t> m_contactRank, m_contactRangeObject;      // what pbdx_solver_set_contact_order was last called with
		std::vector<std::vector<unsigned int> > m_hostGroups;
	};
}

/** Factory with C linkage so that a host application can dlopen() the plug-in. */
extern "C" PBD::TimeStep *pbdx_create_timestep_hip();

#endif
