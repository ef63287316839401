// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Headless C-ABI driver around the *unmodified* reference sources compiled from
// /root/reference by oracle/Makefile.  It restates only the scene set-up of
// Demos/ClothDemo/main.cpp:117-162 and Demos/BarDemo/main.cpp:116-166 as calls
// into the reference's own SimulationModel, and exposes the reference's own
// TimeStepController::step, colouring and constraint data to the tests.
// All values cross the ABI as double (exact for float builds) so one ctypes
// binding serves the f32, f64 and fast builds.
//
// No arithmetic of the hot path is implemented here: every number returned is
// produced by reference code.

#include "Common/Common.h"
#include "Simulation/Simulation.h"
#include "Simulation/SimulationModel.h"
#include "Simulation/TimeManager.h"
#include "Simulation/TimeStepController.h"
#include "Simulation/Constraints.h"
#include "Utils/Logger.h"
#include "Utils/Timing.h"
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <omp.h>

INIT_LOGGING
INIT_TIMING

using namespace PBD;

namespace {
	SimulationModel *g_model = nullptr;

	SimulationModel *model()
	{
		if (!g_model)
		{
			g_model = new SimulationModel();
			g_model->init();
			Simulation::getCurrent()->setModel(g_model);
		}
		return g_model;
	}

	TimeStepController *tsc() { return static_cast<TimeStepController*>(Simulation::getCurrent()->getTimeStep()); }

	Vector3r v3(const double *p) { return Vector3r((Real)p[0], (Real)p[1], (Real)p[2]); }

	// map the reference's run-time TYPE_IDs (Constraints.cpp:17-49) to pbdx_constraint_type
	int mapType(Constraint *c)
	{
		const int t = c->getTypeId();
		if (t == DistanceConstraint::TYPE_ID) return 0;
		if (t == DistanceConstraint_XPBD::TYPE_ID) return 1;
		if (t == DihedralConstraint::TYPE_ID) return 2;
		if (t == IsometricBendingConstraint::TYPE_ID) return 3;
		if (t == IsometricBendingConstraint_XPBD::TYPE_ID) return 4;
		if (t == FEMTriangleConstraint::TYPE_ID) return 5;
		if (t == StrainTriangleConstraint::TYPE_ID) return 6;
		if (t == VolumeConstraint::TYPE_ID) return 7;
		if (t == VolumeConstraint_XPBD::TYPE_ID) return 8;
		if (t == FEMTetConstraint::TYPE_ID) return 9;
		if (t == XPBD_FEMTetConstraint::TYPE_ID) return 10;
		if (t == StrainTetConstraint::TYPE_ID) return 11;
		if (t == ShapeMatchingConstraint::TYPE_ID) return 12;
		return -1;
	}
}

extern "C" {

int refdrv_real_size() { return (int)sizeof(Real); }

// fresh Simulation + SimulationModel + TimeStepController
void refdrv_reset_all()
{
	if (Simulation::hasCurrent())
	{
		Simulation *sim = Simulation::getCurrent();
		delete sim;                      // deletes the time step and the TimeManager
		Simulation::setCurrent(nullptr);
	}
	delete g_model;
	g_model = nullptr;
	model();
	TimeManager::getCurrent()->setTimeStepSize(static_cast<Real>(0.005));
	TimeManager::getCurrent()->setTime(static_cast<Real>(0.0));
}

void refdrv_set_num_threads(int n) { omp_set_num_threads(n); }
int refdrv_max_threads() { return omp_get_max_threads(); }

void refdrv_set_time_step_size(double h) { model(); TimeManager::getCurrent()->setTimeStepSize((Real)h); }
double refdrv_get_time_step_size() { model(); return (double)TimeManager::getCurrent()->getTimeStepSize(); }
double refdrv_get_time() { model(); return (double)TimeManager::getCurrent()->getTime(); }

void refdrv_set_gravity(double gx, double gy, double gz)
{
	model();
	Real g[3] = { (Real)gx, (Real)gy, (Real)gz };
	Simulation::getCurrent()->setVecValue<Real>(Simulation::GRAVITATION, g);
}

void refdrv_set_params(unsigned subSteps, unsigned maxIter, int velMethod)
{
	model();
	TimeStepController *ts = tsc();
	ts->setValue<unsigned int>(TimeStepController::NUM_SUB_STEPS, subSteps);
	ts->setValue<unsigned int>(TimeStepController::MAX_ITERATIONS, maxIter);
	ts->setValue<int>(TimeStepController::VELOCITY_UPDATE_METHOD, velMethod);
}

int refdrv_add_regular_triangle_model(int width, int height, const double *T, const double *R /*row-major*/, const double *scale)
{
	Matrix3r rot;
	for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rot(r, c) = (Real)R[3 * r + c];
	const int idx = (int)model()->getTriangleModels().size();
	model()->addRegularTriangleModel(width, height, v3(T), rot, Vector2r((Real)scale[0], (Real)scale[1]));
	return idx;
}

int refdrv_add_triangle_model(unsigned nPoints, unsigned nFaces, const double *points, const unsigned *indices)
{
	std::vector<Vector3r> pts(nPoints);
	for (unsigned i = 0; i < nPoints; i++) pts[i] = v3(points + 3 * i);
	std::vector<unsigned> idx(indices, indices + 3 * nFaces);
	TriangleModel::ParticleMesh::UVIndices uvi; TriangleModel::ParticleMesh::UVs uvs;
	const int k = (int)model()->getTriangleModels().size();
	model()->addTriangleModel(nPoints, nFaces, pts.data(), idx.data(), uvi, uvs);
	return k;
}

int refdrv_add_regular_tet_model(int width, int height, int depth, const double *T, const double *R, const double *scale)
{
	Matrix3r rot;
	for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rot(r, c) = (Real)R[3 * r + c];
	const int idx = (int)model()->getTetModels().size();
	model()->addRegularTetModel(width, height, depth, v3(T), rot, v3(scale));
	return idx;
}

int refdrv_add_tet_model(unsigned nPoints, unsigned nTets, const double *points, const unsigned *indices)
{
	std::vector<Vector3r> pts(nPoints);
	for (unsigned i = 0; i < nPoints; i++) pts[i] = v3(points + 3 * i);
	std::vector<unsigned> idx(indices, indices + 4 * nTets);
	const int k = (int)model()->getTetModels().size();
	model()->addTetModel(nPoints, nTets, pts.data(), idx.data());
	return k;
}

int refdrv_add_vertex(const double *x) { model()->getParticles().addVertex(v3(x)); return (int)model()->getParticles().size() - 1; }
void refdrv_set_mass(unsigned i, double m) { model()->getParticles().setMass(i, (Real)m); }

void refdrv_add_cloth_constraints(unsigned tm, unsigned method, double k, double xx, double yy, double xy,
	double xyP, double yxP, int normStretch, int normShear)
{
	model()->addClothConstraints(model()->getTriangleModels()[tm], method, (Real)k, (Real)xx, (Real)yy, (Real)xy,
		(Real)xyP, (Real)yxP, normStretch != 0, normShear != 0);
}
void refdrv_add_bending_constraints(unsigned tm, unsigned method, double k)
{
	model()->addBendingConstraints(model()->getTriangleModels()[tm], method, (Real)k);
}
void refdrv_add_solid_constraints(unsigned tm, unsigned method, double k, double poisson, double kv, int normStretch, int normShear)
{
	model()->addSolidConstraints(model()->getTetModels()[tm], method, (Real)k, (Real)poisson, (Real)kv, normStretch != 0, normShear != 0);
}

int refdrv_add_distance_constraint(unsigned a, unsigned b, double k) { return model()->addDistanceConstraint(a, b, (Real)k); }
int refdrv_add_distance_constraint_xpbd(unsigned a, unsigned b, double k) { return model()->addDistanceConstraint_XPBD(a, b, (Real)k); }
int refdrv_add_dihedral_constraint(unsigned a, unsigned b, unsigned c, unsigned d, double k) { return model()->addDihedralConstraint(a, b, c, d, (Real)k); }
int refdrv_add_isometric_bending_constraint(unsigned a, unsigned b, unsigned c, unsigned d, double k) { return model()->addIsometricBendingConstraint(a, b, c, d, (Real)k); }
int refdrv_add_isometric_bending_constraint_xpbd(unsigned a, unsigned b, unsigned c, unsigned d, double k) { return model()->addIsometricBendingConstraint_XPBD(a, b, c, d, (Real)k); }
int refdrv_add_fem_triangle_constraint(unsigned a, unsigned b, unsigned c, double xx, double yy, double xy, double xyP, double yxP)
{ return model()->addFEMTriangleConstraint(a, b, c, (Real)xx, (Real)yy, (Real)xy, (Real)xyP, (Real)yxP); }
int refdrv_add_strain_triangle_constraint(unsigned a, unsigned b, unsigned c, double xx, double yy, double xy, int ns, int nsh)
{ return model()->addStrainTriangleConstraint(a, b, c, (Real)xx, (Real)yy, (Real)xy, ns != 0, nsh != 0); }
int refdrv_add_volume_constraint(unsigned a, unsigned b, unsigned c, unsigned d, double k) { return model()->addVolumeConstraint(a, b, c, d, (Real)k); }
int refdrv_add_volume_constraint_xpbd(unsigned a, unsigned b, unsigned c, unsigned d, double k) { return model()->addVolumeConstraint_XPBD(a, b, c, d, (Real)k); }
int refdrv_add_fem_tet_constraint(unsigned a, unsigned b, unsigned c, unsigned d, double k, double nu) { return model()->addFEMTetConstraint(a, b, c, d, (Real)k, (Real)nu); }
int refdrv_add_fem_tet_constraint_xpbd(unsigned a, unsigned b, unsigned c, unsigned d, double k, double nu) { return model()->addFEMTetConstraint_XPBD(a, b, c, d, (Real)k, (Real)nu); }
int refdrv_add_strain_tet_constraint(unsigned a, unsigned b, unsigned c, unsigned d, double ks, double ksh, int ns, int nsh)
{ return model()->addStrainTetConstraint(a, b, c, d, (Real)ks, (Real)ksh, ns != 0, nsh != 0); }
int refdrv_add_shape_matching_constraint(unsigned n, const unsigned *p, const unsigned *nc, double k)
{ return model()->addShapeMatchingConstraint(n, p, nc, (Real)k); }

// ---- particle state ------------------------------------------------------
unsigned refdrv_num_particles() { return model()->getParticles().size(); }

// which: 0=x 1=x0 2=v 3=a 4=oldX 5=lastX (xyz), 6=mass 7=invMass
void refdrv_get_array(int which, double *out)
{
	ParticleData &pd = model()->getParticles();
	const unsigned n = pd.size();
	for (unsigned i = 0; i < n; i++)
	{
		if (which >= 6) { out[i] = which == 6 ? (double)pd.getMass(i) : (double)pd.getInvMass(i); continue; }
		const Vector3r *p;
		switch (which)
		{
		case 0: p = &pd.getPosition(i); break;
		case 1: p = &pd.getPosition0(i); break;
		case 2: p = &pd.getVelocity(i); break;
		case 3: p = &pd.getAcceleration(i); break;
		case 4: p = &pd.getOldPosition(i); break;
		default: p = &pd.getLastPosition(i); break;
		}
		out[3 * i] = (double)(*p)[0]; out[3 * i + 1] = (double)(*p)[1]; out[3 * i + 2] = (double)(*p)[2];
	}
}

void refdrv_set_array(int which, const double *in)
{
	ParticleData &pd = model()->getParticles();
	const unsigned n = pd.size();
	for (unsigned i = 0; i < n; i++)
	{
		if (which == 6) { pd.setMass(i, (Real)in[i]); continue; }
		if (which == 7) continue;
		Vector3r *p;
		switch (which)
		{
		case 0: p = &pd.getPosition(i); break;
		case 1: p = &pd.getPosition0(i); break;
		case 2: p = &pd.getVelocity(i); break;
		case 3: p = &pd.getAcceleration(i); break;
		case 4: p = &pd.getOldPosition(i); break;
		default: p = &pd.getLastPosition(i); break;
		}
		*p = v3(in + 3 * i);
	}
}

// ---- mesh topology -------------------------------------------------------
unsigned refdrv_triangle_model_num_edges(unsigned tm) { return model()->getTriangleModels()[tm]->getParticleMesh().numEdges(); }
unsigned refdrv_triangle_model_index_offset(unsigned tm) { return model()->getTriangleModels()[tm]->getIndexOffset(); }
void refdrv_triangle_model_get_edges(unsigned tm, unsigned *out)
{
	const auto &edges = model()->getTriangleModels()[tm]->getParticleMesh().getEdges();
	for (size_t i = 0; i < edges.size(); i++)
	{
		out[4 * i] = edges[i].m_vert[0]; out[4 * i + 1] = edges[i].m_vert[1];
		out[4 * i + 2] = edges[i].m_face[0]; out[4 * i + 3] = edges[i].m_face[1];
	}
}
unsigned refdrv_tet_model_num_edges(unsigned tm) { return model()->getTetModels()[tm]->getParticleMesh().numEdges(); }
unsigned refdrv_tet_model_index_offset(unsigned tm) { return model()->getTetModels()[tm]->getIndexOffset(); }
void refdrv_tet_model_get_edges(unsigned tm, unsigned *out)
{
	const auto &edges = model()->getTetModels()[tm]->getParticleMesh().getEdges();
	for (size_t i = 0; i < edges.size(); i++) { out[2 * i] = edges[i].m_vert[0]; out[2 * i + 1] = edges[i].m_vert[1]; }
}

// ---- constraints + colouring --------------------------------------------
unsigned refdrv_num_constraints() { return (unsigned)model()->getConstraints().size(); }
int refdrv_constraint_type(unsigned c) { return mapType(model()->getConstraints()[c]); }
unsigned refdrv_constraint_num_bodies(unsigned c) { return model()->getConstraints()[c]->numberOfBodies(); }
void refdrv_constraint_bodies(unsigned c, unsigned *out)
{
	Constraint *k = model()->getConstraints()[c];
	for (unsigned i = 0; i < k->numberOfBodies(); i++) out[i] = k->m_bodies[i];
}

// parameter record in the layout documented in include/pbdx.h; returns #values
int refdrv_constraint_params(unsigned ci, double *o)
{
	Constraint *c = model()->getConstraints()[ci];
	int n = 0;
	switch (mapType(c))
	{
	case 0: { auto *k = (DistanceConstraint*)c; o[n++] = k->m_restLength; o[n++] = k->m_stiffness; break; }
	case 1: { auto *k = (DistanceConstraint_XPBD*)c; o[n++] = k->m_restLength; o[n++] = k->m_stiffness; break; }
	case 2: { auto *k = (DihedralConstraint*)c; o[n++] = k->m_restAngle; o[n++] = k->m_stiffness; break; }
	case 3: { auto *k = (IsometricBendingConstraint*)c; o[n++] = k->m_stiffness;
		for (int col = 0; col < 4; col++) for (int r = 0; r < 4; r++) o[n++] = k->m_Q(r, col); break; }
	case 4: { auto *k = (IsometricBendingConstraint_XPBD*)c; o[n++] = k->m_stiffness;
		for (int col = 0; col < 4; col++) for (int r = 0; r < 4; r++) o[n++] = k->m_Q(r, col); break; }
	case 5: { auto *k = (FEMTriangleConstraint*)c; o[n++] = k->m_area;
		for (int col = 0; col < 2; col++) for (int r = 0; r < 2; r++) o[n++] = k->m_invRestMat(r, col);
		o[n++] = k->m_xxStiffness; o[n++] = k->m_yyStiffness; o[n++] = k->m_xyStiffness; o[n++] = k->m_xyPoissonRatio; o[n++] = k->m_yxPoissonRatio; break; }
	case 6: { auto *k = (StrainTriangleConstraint*)c;
		for (int col = 0; col < 2; col++) for (int r = 0; r < 2; r++) o[n++] = k->m_invRestMat(r, col);
		o[n++] = k->m_xxStiffness; o[n++] = k->m_yyStiffness; o[n++] = k->m_xyStiffness; o[n++] = k->m_normalizeStretch; o[n++] = k->m_normalizeShear; break; }
	case 7: { auto *k = (VolumeConstraint*)c; o[n++] = k->m_restVolume; o[n++] = k->m_stiffness; break; }
	case 8: { auto *k = (VolumeConstraint_XPBD*)c; o[n++] = k->m_restVolume; o[n++] = k->m_stiffness; break; }
	case 9: { auto *k = (FEMTetConstraint*)c; o[n++] = k->m_volume;
		for (int col = 0; col < 3; col++) for (int r = 0; r < 3; r++) o[n++] = k->m_invRestMat(r, col);
		o[n++] = k->m_stiffness; o[n++] = k->m_poissonRatio; break; }
	case 10: { auto *k = (XPBD_FEMTetConstraint*)c; o[n++] = k->m_volume;
		for (int col = 0; col < 3; col++) for (int r = 0; r < 3; r++) o[n++] = k->m_invRestMat(r, col);
		o[n++] = k->m_stiffness; o[n++] = k->m_poissonRatio; break; }
	case 11: { auto *k = (StrainTetConstraint*)c;
		for (int col = 0; col < 3; col++) for (int r = 0; r < 3; r++) o[n++] = k->m_invRestMat(r, col);
		o[n++] = k->m_stretchStiffness; o[n++] = k->m_shearStiffness; o[n++] = k->m_normalizeStretch; o[n++] = k->m_normalizeShear; break; }
	case 12: { auto *k = (ShapeMatchingConstraint*)c; o[n++] = k->m_stiffness;
		for (int j = 0; j < 3; j++) o[n++] = k->m_restCm[j];
		for (unsigned i = 0; i < k->numberOfBodies(); i++) for (int j = 0; j < 3; j++) o[n++] = k->m_x0[i][j];
		for (unsigned i = 0; i < k->numberOfBodies(); i++) o[n++] = k->m_w[i];
		for (unsigned i = 0; i < k->numberOfBodies(); i++) o[n++] = k->m_numClusters[i];
		break; }
	default: break;
	}
	return n;
}

// XPBD multiplier of constraint ci (NaN-free 0 for non-XPBD types)
double refdrv_constraint_lambda(unsigned ci)
{
	Constraint *c = model()->getConstraints()[ci];
	switch (mapType(c))
	{
	case 1: return ((DistanceConstraint_XPBD*)c)->m_lambda;
	case 4: return ((IsometricBendingConstraint_XPBD*)c)->m_lambda;
	case 8: return ((VolumeConstraint_XPBD*)c)->m_lambda;
	case 10: return ((XPBD_FEMTetConstraint*)c)->m_lambda;
	default: return 0.0;
	}
}

void refdrv_init_constraint_groups() { model()->initConstraintGroups(); }
unsigned refdrv_num_groups() { model()->initConstraintGroups(); return (unsigned)model()->getConstraintGroups().size(); }
unsigned refdrv_group_size(unsigned g) { return (unsigned)model()->getConstraintGroups()[g].size(); }
void refdrv_get_group(unsigned g, unsigned *out)
{
	const auto &grp = model()->getConstraintGroups()[g];
	memcpy(out, grp.data(), grp.size() * sizeof(unsigned));
}

// ---- stepping --------------------------------------------------------------
// n calls of the reference's own TimeStepController::step (or whatever TimeStep is installed)
void refdrv_step(unsigned n)
{
	SimulationModel *m = model();
	TimeStep *ts = Simulation::getCurrent()->getTimeStep();
	for (unsigned i = 0; i < n; i++) ts->step(*m);
}

// wall-clock seconds of n steps (std::chrono around step())
double refdrv_time_steps(unsigned n)
{
	SimulationModel *m = model();
	TimeStep *ts = Simulation::getCurrent()->getTimeStep();
	const auto t0 = std::chrono::steady_clock::now();
	for (unsigned i = 0; i < n; i++) ts->step(*m);
	const auto t1 = std::chrono::steady_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

// Known-answer path: one sweep of every constraint's solvePositionConstraint in
// creation order with the given iteration index (lambda reset when iter==0) --
// exactly the body of the hot loop (TimeStepController.cpp:281-284) without colouring.
void refdrv_solve_position_constraints(unsigned iter)
{
	SimulationModel *m = model();
	for (Constraint *c : m->getConstraints())
	{
		c->updateConstraint(*m);
		c->solvePositionConstraint(*m, iter);
	}
}

// Same sweep but in colour-group order (the order TimeStepController uses).
void refdrv_solve_position_constraints_grouped(unsigned iter)
{
	SimulationModel *m = model();
	m->initConstraintGroups();
	auto &groups = m->getConstraintGroups();
	auto &cs = m->getConstraints();
	for (auto &g : groups)
		for (unsigned ci : g)
		{
			cs[ci]->updateConstraint(*m);
			cs[ci]->solvePositionConstraint(*m, iter);
		}
}

void refdrv_model_reset() { Simulation::getCurrent()->reset(); }

// The currently installed TimeStep object (opaque; for plug-in side counters).
void *refdrv_get_timestep() { model(); return (void*)Simulation::getCurrent()->getTimeStep(); }

// Install a TimeStep plug-in from a shared library: the library must export
//   extern "C" PBD::TimeStep *<symbol>();
// This is the pattern of Demos/PositionBasedElasticRodsDemo/PositionBasedElasticRodsDemo.cpp:51-54:
// delete the current time step, setTimeStep(new), init().  Returns 0 on success.
int refdrv_install_timestep_plugin(const char *path, const char *symbol)
{
	model();
	void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);   // LOCAL: float and double hosts may coexist in one test process
	if (!h) { fprintf(stderr, "refdrv: dlopen failed: %s\n", dlerror()); return 1; }
	typedef TimeStep *(*factory_t)();
	factory_t f = (factory_t)dlsym(h, symbol);
	if (!f) { fprintf(stderr, "refdrv: dlsym failed: %s\n", dlerror()); return 2; }
	TimeStep *ts = f();
	if (!ts) return 3;
	Simulation *sim = Simulation::getCurrent();
	delete sim->getTimeStep();
	sim->setTimeStep(ts);
	ts->init();
	return 0;
}

} // extern "C"
