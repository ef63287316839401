#include "TimeStepControllerHIP.h"
#include "Simulation/Simulation.h"
#include "Simulation/TimeManager.h"
#include "Simulation/Constraints.h"
#include "Utils/Logger.h"
#include "Utils/Timing.h"
#include <stdio.h>

using namespace PBD;

namespace
{
	// reference TYPE_IDs are run-time counters (Simulation/Constraints.cpp:17-49): map by identity
	int engineType(Constraint *c)
	{
		const int t = c->getTypeId();
		if (t == DistanceConstraint::TYPE_ID) return PBDX_DISTANCE;
		if (t == DistanceConstraint_XPBD::TYPE_ID) return PBDX_DISTANCE_XPBD;
		if (t == DihedralConstraint::TYPE_ID) return PBDX_DIHEDRAL;
		if (t == IsometricBendingConstraint::TYPE_ID) return PBDX_ISOMETRIC_BENDING;
		if (t == IsometricBendingConstraint_XPBD::TYPE_ID) return PBDX_ISOMETRIC_BENDING_XPBD;
		if (t == FEMTriangleConstraint::TYPE_ID) return PBDX_FEM_TRIANGLE;
		if (t == StrainTriangleConstraint::TYPE_ID) return PBDX_STRAIN_TRIANGLE;
		if (t == VolumeConstraint::TYPE_ID) return PBDX_VOLUME;
		if (t == VolumeConstraint_XPBD::TYPE_ID) return PBDX_VOLUME_XPBD;
		if (t == FEMTetConstraint::TYPE_ID) return PBDX_FEM_TET;
		if (t == XPBD_FEMTetConstraint::TYPE_ID) return PBDX_FEM_TET_XPBD;
		if (t == StrainTetConstraint::TYPE_ID) return PBDX_STRAIN_TET;
		if (t == ShapeMatchingConstraint::TYPE_ID && c->numberOfBodies() == 4) return PBDX_SHAPE_MATCHING;
		return -1;
	}

	template <typename M> void pushColMajor(std::vector<float> &p, const M &m, int rows, int cols)
	{
		for (int c = 0; c < cols; c++) for (int r = 0; r < rows; r++) p.push_back((float)m(r, c));
	}

	// parameter record in the layout of include/pbdx.h
	void pushParams(std::vector<float> &p, int type, Constraint *c)
	{
		switch (type)
		{
		case PBDX_DISTANCE: { auto *k = (DistanceConstraint*)c; p.push_back((float)k->m_restLength); p.push_back((float)k->m_stiffness); break; }
		case PBDX_DISTANCE_XPBD: { auto *k = (DistanceConstraint_XPBD*)c; p.push_back((float)k->m_restLength); p.push_back((float)k->m_stiffness); break; }
		case PBDX_DIHEDRAL: { auto *k = (DihedralConstraint*)c; p.push_back((float)k->m_restAngle); p.push_back((float)k->m_stiffness); break; }
		case PBDX_ISOMETRIC_BENDING: { auto *k = (IsometricBendingConstraint*)c; p.push_back((float)k->m_stiffness); pushColMajor(p, k->m_Q, 4, 4); break; }
		case PBDX_ISOMETRIC_BENDING_XPBD: { auto *k = (IsometricBendingConstraint_XPBD*)c; p.push_back((float)k->m_stiffness); pushColMajor(p, k->m_Q, 4, 4); break; }
		case PBDX_FEM_TRIANGLE: { auto *k = (FEMTriangleConstraint*)c; p.push_back((float)k->m_area); pushColMajor(p, k->m_invRestMat, 2, 2);
			p.push_back((float)k->m_xxStiffness); p.push_back((float)k->m_yyStiffness); p.push_back((float)k->m_xyStiffness);
			p.push_back((float)k->m_xyPoissonRatio); p.push_back((float)k->m_yxPoissonRatio); break; }
		case PBDX_STRAIN_TRIANGLE: { auto *k = (StrainTriangleConstraint*)c; pushColMajor(p, k->m_invRestMat, 2, 2);
			p.push_back((float)k->m_xxStiffness); p.push_back((float)k->m_yyStiffness); p.push_back((float)k->m_xyStiffness);
			p.push_back(k->m_normalizeStretch ? 1.0f : 0.0f); p.push_back(k->m_normalizeShear ? 1.0f : 0.0f); break; }
		case PBDX_VOLUME: { auto *k = (VolumeConstraint*)c; p.push_back((float)k->m_restVolume); p.push_back((float)k->m_stiffness); break; }
		case PBDX_VOLUME_XPBD: { auto *k = (VolumeConstraint_XPBD*)c; p.push_back((float)k->m_restVolume); p.push_back((float)k->m_stiffness); break; }
		case PBDX_FEM_TET: { auto *k = (FEMTetConstraint*)c; p.push_back((float)k->m_volume); pushColMajor(p, k->m_invRestMat, 3, 3);
			p.push_back((float)k->m_stiffness); p.push_back((float)k->m_poissonRatio); break; }
		case PBDX_FEM_TET_XPBD: { auto *k = (XPBD_FEMTetConstraint*)c; p.push_back((float)k->m_volume); pushColMajor(p, k->m_invRestMat, 3, 3);
			p.push_back((float)k->m_stiffness); p.push_back((float)k->m_poissonRatio); break; }
		case PBDX_STRAIN_TET: { auto *k = (StrainTetConstraint*)c; pushColMajor(p, k->m_invRestMat, 3, 3);
			p.push_back((float)k->m_stretchStiffness); p.push_back((float)k->m_shearStiffness);
			p.push_back(k->m_normalizeStretch ? 1.0f : 0.0f); p.push_back(k->m_normalizeShear ? 1.0f : 0.0f); break; }
		case PBDX_SHAPE_MATCHING: { auto *k = (ShapeMatchingConstraint*)c; p.push_back((float)k->m_stiffness);
			for (int j = 0; j < 3; j++) p.push_back((float)k->m_restCm[j]);
			for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) p.push_back((float)k->m_x0[i][j]);
			for (int i = 0; i < 4; i++) p.push_back((float)k->m_w[i]);
			for (int i = 0; i < 4; i++) p.push_back((float)k->m_numClusters[i]);
			break; }
		default: break;
		}
	}
}

TimeStepControllerHIP::TimeStepControllerHIP(int device) :
	TimeStepController(), m_solver(nullptr), m_device(device), m_scheduleValid(false),
	m_numConstraints(0), m_numParticles(0), m_gpuSteps(0), m_fallbackSteps(0), m_failedSteps(0), m_allowFallback(false)
{
	if (pbdx_solver_create(&m_solver, device) != PBDX_OK)
	{
		LOG_ERR << "TimeStepControllerHIP: " << pbdx_last_error() << " -- every step() will fail (no CPU path)";
		m_solver = nullptr;
	}
}

// A step the engine cannot run: loud error, model untouched -- unless the host opted in to the
// reference's own CPU path.
void TimeStepControllerHIP::refuse(SimulationModel &model, const char *why)
{
	if (m_allowFallback)
	{
		LOG_WARN << "TimeStepControllerHIP: " << why << " -- running this step on the reference CPU path (opted in)";
		m_fallbackSteps++;
		TimeStepController::step(model);
		return;
	}
	LOG_ERR << "TimeStepControllerHIP: " << why << " -- step NOT executed (call setAllowReferenceFallback(true) to use the CPU TimeStepController)";
	fprintf(stderr, "TimeStepControllerHIP: %s -- step NOT executed\n", why);
	m_failedSteps++;
}

TimeStepControllerHIP::~TimeStepControllerHIP()
{
	pbdx_solver_destroy(m_solver);
}

void TimeStepControllerHIP::reset()
{
	TimeStepController::reset();
	m_scheduleValid = false;
}

bool TimeStepControllerHIP::supported(SimulationModel &model) const
{
	if (!m_solver) return false;
	if (!model.getRigidBodies().empty() || model.getOrientations().size() != 0) return false;
	if (!model.getParticleSolidContactConstraints().empty() || !model.getParticleRigidBodyContactConstraints().empty() ||
		!model.getRigidBodyContactConstraints().empty()) return false;
	if (m_collisionDetection != NULL) return false;       // contacts are produced per step on the CPU
	for (Constraint *c : model.getConstraints())
		if (engineType(c) < 0) return false;                // e.g. GenericConstraints, joints, rods
	return true;
}

bool TimeStepControllerHIP::uploadParticles(SimulationModel &model)
{
	ParticleData &pd = model.getParticles();
	const unsigned int n = pd.size();
	m_x.resize(3 * n); m_v.resize(3 * n); m_old.resize(3 * n); m_last.resize(3 * n); m_mass.resize(n); m_invMass.resize(n);
	for (unsigned int i = 0; i < n; i++)
	{
		for (int k = 0; k < 3; k++)
		{
			m_x[3 * i + k] = (float)pd.getPosition(i)[k];
			m_v[3 * i + k] = (float)pd.getVelocity(i)[k];
			m_old[3 * i + k] = (float)pd.getOldPosition(i)[k];
			m_last[3 * i + k] = (float)pd.getLastPosition(i)[k];
		}
		m_mass[i] = (float)pd.getMass(i);
		m_invMass[i] = (float)pd.getInvMass(i);
	}
	m_numParticles = n;
	return pbdx_solver_set_particles(m_solver, n, m_x.data(), m_v.data(), m_old.data(), m_last.data(), m_mass.data(), m_invMass.data()) == PBDX_OK;
}

bool TimeStepControllerHIP::downloadParticles(SimulationModel &model)
{
	ParticleData &pd = model.getParticles();
	const unsigned int n = pd.size();
	if (pbdx_solver_get_particles(m_solver, n, m_x.data(), m_v.data(), m_old.data(), m_last.data()) != PBDX_OK)
		return false;
	for (unsigned int i = 0; i < n; i++)
		for (int k = 0; k < 3; k++)
		{
			pd.getPosition(i)[k] = (Real)m_x[3 * i + k];
			pd.getVelocity(i)[k] = (Real)m_v[3 * i + k];
			pd.getOldPosition(i)[k] = (Real)m_old[3 * i + k];
			pd.getLastPosition(i)[k] = (Real)m_last[3 * i + k];
		}
	return true;
}

bool TimeStepControllerHIP::buildSchedule(SimulationModel &model)
{
	model.initConstraintGroups();                          // TimeStepController.cpp:256
	SimulationModel::ConstraintVector &constraints = model.getConstraints();
	SimulationModel::ConstraintGroupVector &groups = model.getConstraintGroups();
	if (pbdx_solver_begin_schedule(m_solver) != PBDX_OK) return false;
	std::vector<unsigned int> idx;
	std::vector<float> par;
	for (unsigned int g = 0; g < groups.size(); g++)
		for (int type = 0; type < PBDX_NUM_CONSTRAINT_TYPES; type++)
		{
			idx.clear(); par.clear();
			for (unsigned int ci : groups[g])
			{
				Constraint *c = constraints[ci];
				if (engineType(c) != type) continue;
				idx.insert(idx.end(), c->m_bodies.begin(), c->m_bodies.end());
				pushParams(par, type, c);
			}
			if (idx.empty()) continue;
			const unsigned int count = (unsigned int)(idx.size() / pbdx_type_num_bodies(type));
			if (pbdx_solver_add_batch(m_solver, g, type, count, idx.data(), par.data(), pbdx_type_param_stride(type)) != PBDX_OK)
				return false;
		}
	if (pbdx_solver_end_schedule(m_solver) != PBDX_OK) return false;
	m_numConstraints = constraints.size();
	m_scheduleValid = true;
	return true;
}

void TimeStepControllerHIP::step(SimulationModel &model)
{
	if (!supported(model))
	{
		refuse(model, m_solver ? "model contains rigid bodies / contacts / constraint types outside the engine's scope" : "no HIP engine");
		return;
	}
	START_TIMING("simulation step");
	TimeManager *tm = TimeManager::getCurrent();
	const Real h = tm->getTimeStepSize();

	bool ok = uploadParticles(model);
	// topology change: groups re-initialised (every add* clears m_groupsInitialized), counts changed
	if (ok && (!m_scheduleValid || !model.m_groupsInitialized || m_numConstraints != model.getConstraints().size()))
		ok = buildSchedule(model);
	if (ok)
	{
		clearAccelerations(model);                          // host-visible side effect of TimeStepController.cpp:84
		Simulation *sim = Simulation::getCurrent();
		const Real *gr = sim->getVecValue<Real>(Simulation::GRAVITATION);
		const float g[3] = { (float)gr[0], (float)gr[1], (float)gr[2] };
		START_TIMING("position constraints projection");
		ok = pbdx_solver_step(m_solver, (float)h, m_subSteps, m_maxIterations, m_velocityUpdateMethod, g, 1) == PBDX_OK;
		STOP_TIMING_AVG;
		m_iterations = m_maxIterations;
	}
	if (ok)
		ok = downloadParticles(model);
	if (!ok)
	{
		STOP_TIMING_AVG;
		m_scheduleValid = false;
		refuse(model, pbdx_last_error());
		return;
	}
	m_gpuSteps++;
	tm->setTime(tm->getTime() + h);                         // TimeStepController.cpp:239
	STOP_TIMING_AVG;
}

extern "C" PBD::TimeStep *pbdx_create_timestep_hip()
{
	return new PBD::TimeStepControllerHIP(0);
}

extern "C" unsigned int pbdx_timestep_hip_gpu_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numGpuSteps(); }
extern "C" unsigned int pbdx_timestep_hip_fallback_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numFallbackSteps(); }
extern "C" unsigned int pbdx_timestep_hip_failed_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numFailedSteps(); }
extern "C" void pbdx_timestep_hip_allow_reference_fallback(PBD::TimeStep *ts, int allow) { static_cast<PBD::TimeStepControllerHIP*>(ts)->setAllowReferenceFallback(allow != 0); }
