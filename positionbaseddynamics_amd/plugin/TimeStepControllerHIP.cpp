#include "TimeStepControllerHIP.h"
#include "Simulation/Simulation.h"
#include "Simulation/TimeManager.h"
#include "Simulation/Constraints.h"
#include "Simulation/DistanceFieldCollisionDetection.h"
#include "Simulation/RigidBody.h"
#include "Utils/Logger.h"
#include "Utils/Timing.h"
#include <stdio.h>
#include <string.h>

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
	if (model.getOrientations().size() != 0) return false;
	// rigid bodies: only static ones (mass 0), as colliders of a distance-field collision detection
	for (RigidBody *rb : model.getRigidBodies())
		if (rb->getMass() != 0.0) return false;
	if (!model.getRigidBodies().empty() && m_collisionDetection == NULL) return false;
	if (m_collisionDetection != NULL && dynamic_cast<DistanceFieldCollisionDetection*>(m_collisionDetection) == NULL) return false;
	if (m_collisionDetection != NULL)
	{
		typedef DistanceFieldCollisionDetection D;
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
		{
			const int t = co->getTypeId();
			if (co->m_bodyType == CollisionDetection::CollisionObject::RigidBodyCollisionObjectType)
			{
				if (t != D::DistanceFieldCollisionBox::TYPE_ID && t != D::DistanceFieldCollisionSphere::TYPE_ID && t != D::DistanceFieldCollisionTorus::TYPE_ID &&
					t != D::DistanceFieldCollisionCylinder::TYPE_ID && t != D::DistanceFieldCollisionHollowSphere::TYPE_ID && t != D::DistanceFieldCollisionHollowBox::TYPE_ID)
					return false;                             // e.g. cubic SDF (Discregrid) colliders
			}
			else if (t != D::DistanceFieldCollisionObjectWithoutGeometry::TYPE_ID)
				return false;                                 // deformable vs deformable contacts (ParticleTetContactConstraint)
		}
		// more than one tet model registered => solid-solid contacts are possible: not handled
		unsigned int tetObjects = 0;
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
			if (co->m_bodyType == CollisionDetection::CollisionObject::TetModelCollisionObjectType) tetObjects++;
		if (tetObjects > 1) return false;
	}
	for (Constraint *c : model.getConstraints())
		if (engineType(c) < 0) return false;                // e.g. GenericConstraints, joints, rods
	return true;
}

// Colliders = the rigid-body collision objects of the reference's DistanceFieldCollisionDetection, with
// the transformation the reference keeps per rigid body (RigidBody::getTransformationR/V1/V2); collision
// ranges = the triangle / tet models registered with testMesh (DistanceFieldCollisionDetection.cpp:124-154).
bool TimeStepControllerHIP::uploadColliders(SimulationModel &model)
{
	std::vector<pbdx_collider> cols;
	std::vector<pbdx_collision_range> ranges;
	float tolerance = 0.01f;
	if (m_collisionDetection != NULL)
	{
		typedef DistanceFieldCollisionDetection D;
		tolerance = (float)m_collisionDetection->getTolerance();
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
		{
			const int t = co->getTypeId();
			if (co->m_bodyType == CollisionDetection::CollisionObject::RigidBodyCollisionObjectType)
			{
				pbdx_collider c;
				memset(&c, 0, sizeof(c));
				D::DistanceFieldCollisionObject *dco = (D::DistanceFieldCollisionObject*)co;
				c.invert = dco->m_invertSDF < 0 ? 1 : 0;
				if (t == D::DistanceFieldCollisionBox::TYPE_ID) { c.shape = PBDX_SHAPE_BOX; for (int k = 0; k < 3; k++) c.params[k] = (float)((D::DistanceFieldCollisionBox*)co)->m_box[k]; }
				else if (t == D::DistanceFieldCollisionSphere::TYPE_ID) { c.shape = PBDX_SHAPE_SPHERE; c.params[0] = (float)((D::DistanceFieldCollisionSphere*)co)->m_radius; }
				else if (t == D::DistanceFieldCollisionTorus::TYPE_ID) { c.shape = PBDX_SHAPE_TORUS; c.params[0] = (float)((D::DistanceFieldCollisionTorus*)co)->m_radii[0]; c.params[1] = (float)((D::DistanceFieldCollisionTorus*)co)->m_radii[1]; }
				else if (t == D::DistanceFieldCollisionCylinder::TYPE_ID) { c.shape = PBDX_SHAPE_CYLINDER; c.params[0] = (float)((D::DistanceFieldCollisionCylinder*)co)->m_dim[0]; c.params[1] = (float)((D::DistanceFieldCollisionCylinder*)co)->m_dim[1]; }
				else if (t == D::DistanceFieldCollisionHollowSphere::TYPE_ID) { c.shape = PBDX_SHAPE_HOLLOW_SPHERE; c.params[0] = (float)((D::DistanceFieldCollisionHollowSphere*)co)->m_radius; c.params[1] = (float)((D::DistanceFieldCollisionHollowSphere*)co)->m_thickness; }
				else { c.shape = PBDX_SHAPE_HOLLOW_BOX; for (int k = 0; k < 3; k++) c.params[k] = (float)((D::DistanceFieldCollisionHollowBox*)co)->m_box[k]; c.params[3] = (float)((D::DistanceFieldCollisionHollowBox*)co)->m_thickness; }
				RigidBody *rb = model.getRigidBodies()[co->m_bodyIndex];
				const Matrix3r &R = rb->getTransformationR();
				for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) c.R[3 * r + k] = (float)R(r, k);
				for (int k = 0; k < 3; k++)
				{
					c.com[k] = (float)rb->getPosition()[k];
					c.v1[k] = (float)rb->getTransformationV1()[k];
					c.v2[k] = (float)rb->getTransformationV2()[k];
					c.body_v[k] = (float)rb->getVelocity()[k];
					c.body_omega[k] = (float)rb->getAngularVelocity()[k];
				}
				c.restitution = (float)rb->getRestitutionCoeff();
				c.friction = (float)rb->getFrictionCoeff();
				c.body_index = co->m_bodyIndex;
				cols.push_back(c);
			}
			else if (((D::DistanceFieldCollisionObject*)co)->m_testMesh)
			{
				pbdx_collision_range r;
				if (co->m_bodyType == CollisionDetection::CollisionObject::TriangleModelCollisionObjectType)
				{
					TriangleModel *tm = model.getTriangleModels()[co->m_bodyIndex];
					r.first = tm->getIndexOffset(); r.count = tm->getParticleMesh().numVertices();
					r.restitution = (float)tm->getRestitutionCoeff(); r.friction = (float)tm->getFrictionCoeff();
				}
				else
				{
					TetModel *tm = model.getTetModels()[co->m_bodyIndex];
					r.first = tm->getIndexOffset(); r.count = tm->getParticleMesh().numVertices();
					r.restitution = (float)tm->getRestitutionCoeff(); r.friction = (float)tm->getFrictionCoeff();
				}
				ranges.push_back(r);
			}
		}
	}
	if (pbdx_solver_set_colliders(m_solver, (uint32_t)cols.size(), cols.data()) != PBDX_OK) return false;
	if (pbdx_solver_set_collision_ranges(m_solver, (uint32_t)ranges.size(), ranges.data()) != PBDX_OK) return false;
	return pbdx_solver_set_contact_params(m_solver, tolerance, (float)model.getContactStiffnessParticleRigidBody(), m_maxIterationsV) == PBDX_OK;
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
	if (n != m_numParticles) m_scheduleValid = false;      // the engine drops its schedule with the old particle image
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
		ok = uploadColliders(model);                        // cheap; poses / coefficients are host-mutable between steps
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
	{
		// TimeStepController.cpp:216-223: the reference rebuilds the contact lists every step; the device keeps
		// its contacts to itself, so the host lists are emptied (counts: pbdx_solver_get_num_contacts)
		if (m_collisionDetection != NULL) model.resetContacts();
		m_iterationsV = m_maxIterationsV;
		ok = downloadParticles(model);
	}
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
