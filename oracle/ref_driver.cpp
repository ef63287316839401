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
#include "Simulation/DistanceFieldCollisionDetection.h"
#include "Simulation/RigidBody.h"
#include "GenericConstraints.h"          // Demos/GenericConstraintsDemos (mixed-model tests)
#include "PositionBasedDynamics/PositionBasedDynamics.h"
#include "PositionBasedDynamics/PositionBasedRigidBodyDynamics.h"
#include "Utils/IndexedFaceMesh.h"
#include "Utils/TetGenLoader.h"
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

	// collision detection of the scene (Demos/DistanceFieldDemos/ClothCollisionDemo.cpp:40,163-180)
	DistanceFieldCollisionDetection *g_cd = nullptr;
	DistanceFieldCollisionDetection &cd()
	{
		if (!g_cd) g_cd = new DistanceFieldCollisionDetection();
		return *g_cd;
	}

	// unit cube surface mesh (data/models/cube.obj: +-0.5), used as the geometry of every static collider
	// (scaled to the bounding box of its analytic distance field, like the demo scales cube.obj / torus.obj)
	void cubeMesh(VertexData &vd, Utilities::IndexedFaceMesh &mesh)
	{
		static const Real V[8][3] = { {-0.5,-0.5,0.5}, {0.5,-0.5,0.5}, {-0.5,0.5,0.5}, {0.5,0.5,0.5}, {-0.5,0.5,-0.5}, {0.5,0.5,-0.5}, {-0.5,-0.5,-0.5}, {0.5,-0.5,-0.5} };
		static const int F[12][3] = { {0,1,2}, {2,1,3}, {2,3,4}, {4,3,5}, {4,5,6}, {6,5,7}, {6,7,0}, {0,7,1}, {1,7,3}, {3,7,5}, {6,0,4}, {4,0,2} };
		mesh.release();
		mesh.initMesh(8, 24, 12);
		vd.reserve(8);
		for (int i = 0; i < 8; i++) vd.addVertex(Vector3r(V[i][0], V[i][1], V[i][2]));
		for (int i = 0; i < 12; i++) { int f[3] = { F[i][0], F[i][1], F[i][2] }; mesh.addFace(&f[0]); }
		mesh.buildNeighbors();
		mesh.updateNormals(vd, 0);
		mesh.updateVertexNormals(vd);
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

// nodes of a bounding-sphere hierarchy are numbered in creation order; their number is found by walking the tree
template <class BVH> static unsigned dumpBvh(const BVH &bvh, unsigned ne, unsigned *lst, unsigned lstCap, int *nodes, double *hulls, unsigned nodeCap)
{
	if (!ne) return 0;
	std::vector<unsigned> stack; stack.push_back(0);
	unsigned maxIdx = 0;
	while (!stack.empty())
	{
		const unsigned n = stack.back(); stack.pop_back();
		if (n > maxIdx) maxIdx = n;
		if (!bvh.node(n).is_leaf()) { stack.push_back((unsigned)bvh.node(n).children[0]); stack.push_back((unsigned)bvh.node(n).children[1]); }
	}
	const unsigned nNodes = maxIdx + 1;
	if (lst && ne <= lstCap) for (unsigned i = 0; i < ne; i++) lst[i] = bvh.entity(i);
	if (nodes && hulls && nNodes <= nodeCap)
		for (unsigned i = 0; i < nNodes; i++)
		{
			nodes[4 * i] = bvh.node(i).children[0]; nodes[4 * i + 1] = bvh.node(i).children[1]; nodes[4 * i + 2] = (int)bvh.node(i).begin; nodes[4 * i + 3] = (int)bvh.node(i).n;
			hulls[4 * i] = (double)bvh.hull(i).x()[0]; hulls[4 * i + 1] = (double)bvh.hull(i).x()[1]; hulls[4 * i + 2] = (double)bvh.hull(i).x()[2]; hulls[4 * i + 3] = (double)bvh.hull(i).r();
		}
	return nNodes;
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
	delete g_cd;
	g_cd = nullptr;
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

// A tet model from TetGen files placed as Demos/SceneLoaderDemo/SceneLoaderDemo.cpp:606-655 places the models of a scene file: the reference's own
// loader (Utils/TetGenLoader.cpp), vertices[j] = R * (vertices[j] .* scale) + x with R = Quaternionr(AngleAxisr(angle, axis)).matrix()
// (Utils/SceneLoader.cpp:366-370), addTetModel, setInitialX / setInitialR / setInitialScale.  Returns the tet model index, -1 if the files are missing.
int refdrv_add_tetgen_model(const char *nodeFile, const char *eleFile, const double *x, const double *axis, double angle, const double *scale)
{
	std::vector<Vector3r> vertices;
	std::vector<unsigned int> tets;
	Utilities::TetGenLoader::loadTetgenModel(nodeFile, eleFile, vertices, tets);
	if (vertices.empty() || tets.empty()) return -1;
	const Quaternionr q = Quaternionr(AngleAxisr((Real)angle, v3(axis)));
	const Matrix3r R = q.matrix();
	const Vector3r X = v3(x), S = v3(scale);
	for (unsigned int j = 0; j < vertices.size(); j++)
		vertices[j] = R * (vertices[j].cwiseProduct(S)) + X;
	SimulationModel *m = model();
	m->addTetModel((unsigned int)vertices.size(), (unsigned int)tets.size() / 4, vertices.data(), tets.data());
	TetModel *tm = m->getTetModels()[m->getTetModels().size() - 1];
	tm->setInitialX(X);
	tm->setInitialR(R);
	tm->setInitialScale(S);
	return (int)m->getTetModels().size() - 1;
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

// ---- static colliders + distance-field collision detection (particle vs rigid body contacts) -------------
// A static rigid body (mass 0) at pos / quaternion (w,x,y,z) whose geometry is the unit cube scaled to
// `bbox`, with an analytic distance field of the given shape attached:
//   shape 0 box(dims = p[0..2]); 1 sphere(radius p[0]); 2 torus(radii p[0], p[1]); 3 cylinder(radius p[0], height p[1]);
//   4 hollow sphere(radius p[0], thickness p[1]); 5 hollow box(dims p[0..2], thickness p[3])
// Returns the rigid body index.
int refdrv_add_static_collider(int shape, const double *pos, const double *quat, const double *bbox, const double *p,
	double restitution, double friction, int invertSDF)
{
	SimulationModel *m = model();
	VertexData vd; Utilities::IndexedFaceMesh mesh;
	cubeMesh(vd, mesh);
	RigidBody *rb = new RigidBody();
	rb->initBody(static_cast<Real>(1.0), v3(pos), Quaternionr((Real)quat[0], (Real)quat[1], (Real)quat[2], (Real)quat[3]), vd, mesh, v3(bbox));
	rb->setMass(0.0);
	rb->setRestitutionCoeff((Real)restitution);
	rb->setFrictionCoeff((Real)friction);
	SimulationModel::RigidBodyVector &rbs = m->getRigidBodies();
	rbs.push_back(rb);
	const unsigned int idx = (unsigned int)rbs.size() - 1;
	const std::vector<Vector3r> &verts = rb->getGeometry().getVertexDataLocal().getVertices();
	const unsigned int nv = (unsigned int)verts.size();
	const unsigned int T = CollisionDetection::CollisionObject::RigidBodyCollisionObjectType;
	switch (shape)
	{
	case 0: cd().addCollisionBox(idx, T, verts.data(), nv, Vector3r((Real)p[0], (Real)p[1], (Real)p[2]), true, invertSDF != 0); break;
	case 1: cd().addCollisionSphere(idx, T, verts.data(), nv, (Real)p[0], true, invertSDF != 0); break;
	case 2: cd().addCollisionTorus(idx, T, verts.data(), nv, Vector2r((Real)p[0], (Real)p[1]), true, invertSDF != 0); break;
	case 3: cd().addCollisionCylinder(idx, T, verts.data(), nv, Vector2r((Real)p[0], (Real)p[1]), true, invertSDF != 0); break;
	case 4: cd().addCollisionHollowSphere(idx, T, verts.data(), nv, (Real)p[0], (Real)p[1], true, invertSDF != 0); break;
	case 5: cd().addCollisionHollowBox(idx, T, verts.data(), nv, Vector3r((Real)p[0], (Real)p[1], (Real)p[2]), (Real)p[3], true, invertSDF != 0); break;
	default: return -1;
	}
	return (int)idx;
}

// A rigid body of FINITE mass (density * volume of `bbox`) with an analytic distance field: the impulse sink of particle contacts
// (ParticleRigidBodyContactConstraint with a dynamic body, Constraints.cpp:2148-2189).  testMesh = 0: the body's own vertices are not tested against other
// objects.  Returns the rigid body index.
int refdrv_add_dynamic_collider(int shape, const double *pos, const double *quat, const double *bbox, const double *p,
	double density, double restitution, double friction, int testMesh)
{
	SimulationModel *m = model();
	VertexData vd; Utilities::IndexedFaceMesh mesh;
	cubeMesh(vd, mesh);
	RigidBody *rb = new RigidBody();
	rb->initBody(static_cast<Real>(density), v3(pos), Quaternionr((Real)quat[0], (Real)quat[1], (Real)quat[2], (Real)quat[3]), vd, mesh, v3(bbox));
	rb->setRestitutionCoeff((Real)restitution);
	rb->setFrictionCoeff((Real)friction);
	SimulationModel::RigidBodyVector &rbs = m->getRigidBodies();
	rbs.push_back(rb);
	const unsigned int idx = (unsigned int)rbs.size() - 1;
	const std::vector<Vector3r> &verts = rb->getGeometry().getVertexDataLocal().getVertices();
	const unsigned int nv = (unsigned int)verts.size();
	const unsigned int T = CollisionDetection::CollisionObject::RigidBodyCollisionObjectType;
	switch (shape)
	{
	case 0: cd().addCollisionBox(idx, T, verts.data(), nv, Vector3r((Real)p[0], (Real)p[1], (Real)p[2]), testMesh != 0, false); break;
	case 1: cd().addCollisionSphere(idx, T, verts.data(), nv, (Real)p[0], testMesh != 0, false); break;
	case 2: cd().addCollisionTorus(idx, T, verts.data(), nv, Vector2r((Real)p[0], (Real)p[1]), testMesh != 0, false); break;
	case 3: cd().addCollisionCylinder(idx, T, verts.data(), nv, Vector2r((Real)p[0], (Real)p[1]), testMesh != 0, false); break;
	default: return -1;
	}
	return (int)idx;
}
// state of a rigid body: position (3), rotation quaternion w x y z (4), velocity (3), angular velocity (3), mass (1)
int refdrv_get_rigid_body_state(unsigned index, double *out)
{
	SimulationModel *m = model();
	if (index >= m->getRigidBodies().size()) return -1;
	RigidBody *rb = m->getRigidBodies()[index];
	for (int k = 0; k < 3; k++) { out[k] = (double)rb->getPosition()[k]; out[7 + k] = (double)rb->getVelocity()[k]; out[10 + k] = (double)rb->getAngularVelocity()[k]; }
	out[3] = (double)rb->getRotation().w(); out[4] = (double)rb->getRotation().x(); out[5] = (double)rb->getRotation().y(); out[6] = (double)rb->getRotation().z();
	out[13] = (double)rb->getMass();
	return 0;
}
// The reference's own functions for ONE contact of a particle with a rigid body (any mass): init_ParticleRigidBodyContactConstraint, then `sweeps` times
// velocitySolve_ParticleRigidBodyContactConstraint applied as ParticleRigidBodyContactConstraint::solveVelocityConstraint applies it (Constraints.cpp:2148-2189).
// in / out: the layout of pbdx_debug_dyn_contact_kat (csrc/pbdx_tetcontact.cpp), as doubles.
void refdrv_dyn_contact_kat(const double *in, double *out)
{
	const Real w0 = (Real)in[0], m0 = (Real)in[1], w1 = (Real)in[5];
	Vector3r v0 = v3(in + 2), x1 = v3(in + 6), v1 = v3(in + 9), om = v3(in + 21);
	Matrix3r Ji;
	for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ji(r, c) = (Real)in[12 + 3 * r + c];
	const Vector3r cp0 = v3(in + 24), cp1 = v3(in + 27), n = v3(in + 30);
	const Vector3r x0 = cp0;
	Eigen::Matrix<Real, 3, 5, Eigen::DontAlign> info;
	PositionBasedRigidBodyDynamics::init_ParticleRigidBodyContactConstraint(w0, x0, v0, w1, x1, v1, Ji, Quaternionr(1, 0, 0, 0), om, cp0, cp1, n, (Real)in[33], info);
	for (int k = 0; k < 3; k++) out[k] = (double)info(k, 3);
	out[3] = (double)info(0, 4); out[4] = (double)info(1, 4); out[5] = (double)info(2, 4);
	Real sum = 0.0;
	for (int it = 0; it < (int)in[36]; it++)
	{
		Vector3r c0, c1, cw;
		if (PositionBasedRigidBodyDynamics::velocitySolve_ParticleRigidBodyContactConstraint(w0, x0, v0, w1, x1, v1, Ji, om, (Real)in[34], (Real)in[35], sum, info, c0, c1, cw))
		{
			if (m0 != 0.0) v0 += c0;
			if (w1 != 0.0) { v1 += c1; om += cw; }
		}
	}
	for (int k = 0; k < 3; k++) { out[6 + k] = (double)v0[k]; out[9 + k] = (double)v1[k]; out[12 + k] = (double)om[k]; }
	out[15] = (double)sum; out[16] = out[17] = out[18] = out[19] = 0.0;
}

// mass 0 = static (RigidBody::setMass also clears the inverse mass); with refdrv_add_dynamic_collider(..., testMesh = 0): a static collider whose own mesh is
// not tested against other rigid bodies
void refdrv_set_rigid_body_mass(unsigned index, double mass)
{
	SimulationModel *m = model();
	if (index < m->getRigidBodies().size()) m->getRigidBodies()[index]->setMass((Real)mass);
}
void refdrv_set_rigid_body_velocity(unsigned index, const double *v, const double *omega)
{
	SimulationModel *m = model();
	if (index >= m->getRigidBodies().size()) return;
	RigidBody *rb = m->getRigidBodies()[index];
	rb->getVelocity() = v3(v); rb->getAngularVelocity() = v3(omega);
}

// Register every triangle / tet model as a collision object without geometry (its particles are tested
// against the colliders) and attach the collision detection to the time step
// (ClothCollisionDemo.cpp:163-180).
void refdrv_enable_collisions(double tolerance, double modelRestitution, double modelFriction)
{
	SimulationModel *m = model();
	cd().setTolerance((Real)tolerance);
	ParticleData &pd = m->getParticles();
	SimulationModel::TriangleModelVector &tms = m->getTriangleModels();
	for (unsigned int i = 0; i < tms.size(); i++)
	{
		tms[i]->setRestitutionCoeff((Real)modelRestitution);
		tms[i]->setFrictionCoeff((Real)modelFriction);
		cd().addCollisionObjectWithoutGeometry(i, CollisionDetection::CollisionObject::TriangleModelCollisionObjectType,
			&pd.getPosition(tms[i]->getIndexOffset()), tms[i]->getParticleMesh().numVertices(), true);
	}
	SimulationModel::TetModelVector &tets = m->getTetModels();
	for (unsigned int i = 0; i < tets.size(); i++)
	{
		tets[i]->setRestitutionCoeff((Real)modelRestitution);
		tets[i]->setFrictionCoeff((Real)modelFriction);
		cd().addCollisionObjectWithoutGeometry(i, CollisionDetection::CollisionObject::TetModelCollisionObjectType,
			&pd.getPosition(tets[i]->getIndexOffset()), tets[i]->getParticleMesh().numVertices(), true);
	}
	Simulation::getCurrent()->getTimeStep()->setCollisionDetection(*m, &cd());
}

// Collision object i as numbers (for feeding the product's raw collider API in tests):
// out[0] body type (0 rigid body, 1 triangle model, 2 tet model), [1] shape (0..5, -1 = without geometry), [2] invertSDF,
// [3..6] shape parameters AS STORED (m_box, m_radius, m_radii, m_dim, thickness), [7..9] body position, [10..18] transformation R
// (row-major), [19..21] v1, [22..24] v2, [25] restitution, [26] friction, [27] body index, [28] first particle, [29] particle count,
// [30] body mass, [31] tolerance
// (re)attach the scene's collision detection to the CURRENT time step (after installing a plug-in)
void refdrv_attach_collision_detection() { Simulation::getCurrent()->getTimeStep()->setCollisionDetection(*model(), &cd()); }
unsigned refdrv_num_collision_objects() { return (unsigned)cd().getCollisionObjects().size(); }
void refdrv_get_collision_object(unsigned i, double *out)
{
	typedef DistanceFieldCollisionDetection D;
	SimulationModel *m = model();
	CollisionDetection::CollisionObject *co = cd().getCollisionObjects()[i];
	for (int k = 0; k < 32; k++) out[k] = 0.0;
	out[27] = co->m_bodyIndex;
	out[31] = (double)cd().getTolerance();
	const int t = co->getTypeId();
	int shape = -1;
	if (t == D::DistanceFieldCollisionBox::TYPE_ID) { shape = 0; auto *c = (D::DistanceFieldCollisionBox*)co; for (int k = 0; k < 3; k++) out[3 + k] = (double)c->m_box[k]; }
	else if (t == D::DistanceFieldCollisionSphere::TYPE_ID) { shape = 1; out[3] = (double)((D::DistanceFieldCollisionSphere*)co)->m_radius; }
	else if (t == D::DistanceFieldCollisionTorus::TYPE_ID) { shape = 2; auto *c = (D::DistanceFieldCollisionTorus*)co; out[3] = (double)c->m_radii[0]; out[4] = (double)c->m_radii[1]; }
	else if (t == D::DistanceFieldCollisionCylinder::TYPE_ID) { shape = 3; auto *c = (D::DistanceFieldCollisionCylinder*)co; out[3] = (double)c->m_dim[0]; out[4] = (double)c->m_dim[1]; }
	else if (t == D::DistanceFieldCollisionHollowSphere::TYPE_ID) { shape = 4; auto *c = (D::DistanceFieldCollisionHollowSphere*)co; out[3] = (double)c->m_radius; out[4] = (double)c->m_thickness; }
	else if (t == D::DistanceFieldCollisionHollowBox::TYPE_ID) { shape = 5; auto *c = (D::DistanceFieldCollisionHollowBox*)co; for (int k = 0; k < 3; k++) out[3 + k] = (double)c->m_box[k]; out[6] = (double)c->m_thickness; }
	out[1] = shape;
	if (shape >= 0) out[2] = (((D::DistanceFieldCollisionObject*)co)->m_invertSDF < 0) ? 1.0 : 0.0;
	if (co->m_bodyType == CollisionDetection::CollisionObject::RigidBodyCollisionObjectType)
	{
		out[0] = 0;
		RigidBody *rb = m->getRigidBodies()[co->m_bodyIndex];
		for (int k = 0; k < 3; k++) out[7 + k] = (double)rb->getPosition()[k];
		const Matrix3r &R = rb->getTransformationR();
		for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[10 + 3 * r + c] = (double)R(r, c);
		for (int k = 0; k < 3; k++) { out[19 + k] = (double)rb->getTransformationV1()[k]; out[22 + k] = (double)rb->getTransformationV2()[k]; }
		out[25] = (double)rb->getRestitutionCoeff(); out[26] = (double)rb->getFrictionCoeff(); out[30] = (double)rb->getMass();
	}
	else if (co->m_bodyType == CollisionDetection::CollisionObject::TriangleModelCollisionObjectType)
	{
		out[0] = 1;
		TriangleModel *tm = m->getTriangleModels()[co->m_bodyIndex];
		out[25] = (double)tm->getRestitutionCoeff(); out[26] = (double)tm->getFrictionCoeff();
		out[28] = tm->getIndexOffset(); out[29] = tm->getParticleMesh().numVertices();
	}
	else
	{
		out[0] = 2;
		TetModel *tm = m->getTetModels()[co->m_bodyIndex];
		out[25] = (double)tm->getRestitutionCoeff(); out[26] = (double)tm->getFrictionCoeff();
		out[28] = tm->getIndexOffset(); out[29] = tm->getParticleMesh().numVertices();
	}
}
// ---- deformable vs deformable: tet models with an ANALYTIC box distance field in their rest frame ----------------------------
// (what the demos do with Discregrid SDFs -- SceneLoaderDemo.cpp:740-760, pyPBD/SimulationModelModule.cpp:175-200 -- with
// DistanceFieldCollisionDetection::addCollisionBox on a TetModelCollisionObjectType body + initTetBVH; the box is evaluated at
// R0^T (X - X0) with X the point mapped into the tet model's rest configuration, DistanceFieldCollisionDetection.cpp:419-430.)
// box = full side lengths; the tet model's initial transform must describe where that box sits: set_tet_model_initial_transform.
int refdrv_add_tet_collision_box(unsigned tetModel, const double *box, int testMesh, double restitution, double friction)
{
	SimulationModel *m = model();
	if (tetModel >= m->getTetModels().size()) return -1;
	TetModel *tm = m->getTetModels()[tetModel];
	tm->setRestitutionCoeff((Real)restitution);
	tm->setFrictionCoeff((Real)friction);
	ParticleData &pd = m->getParticles();
	const unsigned int offset = tm->getIndexOffset();
	const Utilities::IndexedTetMesh &mesh = tm->getParticleMesh();
	cd().addCollisionBox(tetModel, CollisionDetection::CollisionObject::TetModelCollisionObjectType, &pd.getPosition(offset), mesh.numVertices(),
		Vector3r((Real)box[0], (Real)box[1], (Real)box[2]), testMesh != 0, false);
	const unsigned int index = (unsigned int)cd().getCollisionObjects().size() - 1;
	((DistanceFieldCollisionDetection::DistanceFieldCollisionObject*)cd().getCollisionObjects()[index])->initTetBVH(&pd.getPosition(offset), mesh.numVertices(),
		mesh.getTets().data(), mesh.numTets(), cd().getTolerance());
	return (int)index;
}
// the same with any of the analytic shapes (0 box, 1 sphere, 2 torus, 3 cylinder, 4 hollow sphere, 5 hollow box; p as in refdrv_add_static_collider)
int refdrv_add_tet_collision_shape(unsigned tetModel, int shape, const double *p, int testMesh, int invertSDF, double restitution, double friction)
{
	SimulationModel *m = model();
	if (tetModel >= m->getTetModels().size()) return -1;
	TetModel *tm = m->getTetModels()[tetModel];
	tm->setRestitutionCoeff((Real)restitution);
	tm->setFrictionCoeff((Real)friction);
	ParticleData &pd = m->getParticles();
	const unsigned int offset = tm->getIndexOffset();
	const Utilities::IndexedTetMesh &mesh = tm->getParticleMesh();
	const Vector3r *verts = &pd.getPosition(offset);
	const unsigned int nv = mesh.numVertices();
	const unsigned int T = CollisionDetection::CollisionObject::TetModelCollisionObjectType;
	switch (shape)
	{
	case 0: cd().addCollisionBox(tetModel, T, verts, nv, Vector3r((Real)p[0], (Real)p[1], (Real)p[2]), testMesh != 0, invertSDF != 0); break;
	case 1: cd().addCollisionSphere(tetModel, T, verts, nv, (Real)p[0], testMesh != 0, invertSDF != 0); break;
	case 2: cd().addCollisionTorus(tetModel, T, verts, nv, Vector2r((Real)p[0], (Real)p[1]), testMesh != 0, invertSDF != 0); break;
	case 3: cd().addCollisionCylinder(tetModel, T, verts, nv, Vector2r((Real)p[0], (Real)p[1]), testMesh != 0, invertSDF != 0); break;
	case 4: cd().addCollisionHollowSphere(tetModel, T, verts, nv, (Real)p[0], (Real)p[1], testMesh != 0, invertSDF != 0); break;
	case 5: cd().addCollisionHollowBox(tetModel, T, verts, nv, Vector3r((Real)p[0], (Real)p[1], (Real)p[2]), (Real)p[3], testMesh != 0, invertSDF != 0); break;
	default: return -1;
	}
	const unsigned int index = (unsigned int)cd().getCollisionObjects().size() - 1;
	((DistanceFieldCollisionDetection::DistanceFieldCollisionObject*)cd().getCollisionObjects()[index])->initTetBVH(verts, nv, mesh.getTets().data(), mesh.numTets(), cd().getTolerance());
	return (int)index;
}
// what a collision object stores about its distance field: shape id as above (-1: none / unknown), invertSDF, the members the field is evaluated from
int refdrv_get_collision_object_shape(unsigned co, int *invert, double *p)
{
	typedef DistanceFieldCollisionDetection D;
	if (co >= cd().getCollisionObjects().size()) return -1;
	CollisionDetection::CollisionObject *o = cd().getCollisionObjects()[co];
	const int t = o->getTypeId();
	for (int k = 0; k < 4; k++) p[k] = 0.0;
	*invert = ((D::DistanceFieldCollisionObject*)o)->m_invertSDF < 0 ? 1 : 0;
	if (t == D::DistanceFieldCollisionBox::TYPE_ID) { for (int k = 0; k < 3; k++) p[k] = (double)((D::DistanceFieldCollisionBox*)o)->m_box[k]; return 0; }
	if (t == D::DistanceFieldCollisionSphere::TYPE_ID) { p[0] = (double)((D::DistanceFieldCollisionSphere*)o)->m_radius; return 1; }
	if (t == D::DistanceFieldCollisionTorus::TYPE_ID) { p[0] = (double)((D::DistanceFieldCollisionTorus*)o)->m_radii[0]; p[1] = (double)((D::DistanceFieldCollisionTorus*)o)->m_radii[1]; return 2; }
	if (t == D::DistanceFieldCollisionCylinder::TYPE_ID) { p[0] = (double)((D::DistanceFieldCollisionCylinder*)o)->m_dim[0]; p[1] = (double)((D::DistanceFieldCollisionCylinder*)o)->m_dim[1]; return 3; }
	if (t == D::DistanceFieldCollisionHollowSphere::TYPE_ID) { p[0] = (double)((D::DistanceFieldCollisionHollowSphere*)o)->m_radius; p[1] = (double)((D::DistanceFieldCollisionHollowSphere*)o)->m_thickness; return 4; }
	if (t == D::DistanceFieldCollisionHollowBox::TYPE_ID) { for (int k = 0; k < 3; k++) p[k] = (double)((D::DistanceFieldCollisionHollowBox*)o)->m_box[k]; p[3] = (double)((D::DistanceFieldCollisionHollowBox*)o)->m_thickness; return 5; }
	return -1;
}
void refdrv_set_tet_model_initial_transform(unsigned tetModel, const double *x, const double *R /*row-major*/)
{
	TetModel *tm = model()->getTetModels()[tetModel];
	tm->setInitialX(v3(x));
	Matrix3r r;
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = (Real)R[3 * i + j];
	tm->setInitialR(r);
}
// tet model tm: out[0] index offset, [1] #vertices, [2] #tets, [3..5] initialX, [6..14] initialR (row-major); tets: 4 per tet
void refdrv_tet_model_info(unsigned tmIdx, double *out, unsigned *tets, unsigned tetCap)
{
	TetModel *tm = model()->getTetModels()[tmIdx];
	const Utilities::IndexedTetMesh &mesh = tm->getParticleMesh();
	out[0] = tm->getIndexOffset(); out[1] = mesh.numVertices(); out[2] = mesh.numTets();
	for (int k = 0; k < 3; k++) out[3 + k] = (double)tm->getInitialX()[k];
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[6 + 3 * i + j] = (double)tm->getInitialR()(i, j);
	if (tets && 4 * mesh.numTets() <= tetCap) for (unsigned i = 0; i < 4 * mesh.numTets(); i++) tets[i] = mesh.getTets()[i];
}
void refdrv_set_collision_tolerance(double t) { cd().setTolerance((Real)t); }
unsigned refdrv_num_particle_solid_contacts() { return (unsigned)model()->getParticleSolidContactConstraints().size(); }
// contact i: out[0] particle, [1] solid (tet model), [2] tet, [3..5] bary, [6..14] constraintInfo (3x3, column-major), [15] friction,
// [16..27] m_x (4 x 3), [28..31] m_invMasses, [32] m_lambda
void refdrv_get_particle_solid_contact(unsigned i, double *out)
{
	ParticleTetContactConstraint &c = model()->getParticleSolidContactConstraints()[i];
	out[0] = c.m_bodies[0]; out[1] = c.m_solidIndex; out[2] = c.m_tetIndex;
	for (int k = 0; k < 3; k++) out[3 + k] = (double)c.m_bary[k];
	for (int col = 0; col < 3; col++) for (int r = 0; r < 3; r++) out[6 + 3 * col + r] = (double)c.m_constraintInfo(r, col);
	out[15] = (double)c.m_frictionCoeff;
	for (int v = 0; v < 4; v++) for (int k = 0; k < 3; k++) out[16 + 3 * v + k] = (double)c.m_x[v][k];
	for (int v = 0; v < 4; v++) out[28 + v] = (double)c.m_invMasses[v];
	out[32] = (double)c.m_lambda;
}
// the bounding-sphere hierarchies of collision object `co` as the reference constructed them (kd-tree construction uses
// std::sort on tied coordinates: the structure is taken from the reference, never rebuilt):
//   which = 0 point cloud (m_bvh), 1 tets (m_bvhTets), 2 tets in the rest pose (m_bvhTets0)
// returns the number of nodes; lst: entity order (numEntities), nodes: 4 ints per node (child0, child1, begin, n), hulls: 4 doubles per node (x, y, z, r)
unsigned refdrv_get_bvh(unsigned co, int which, unsigned *lst, unsigned lstCap, int *nodes, double *hulls, unsigned nodeCap, unsigned *numEntities)
{
	typedef DistanceFieldCollisionDetection::DistanceFieldCollisionObject DCO;
	DCO *o = (DCO*)cd().getCollisionObjects()[co];
	SimulationModel *m = model();
	unsigned nv = 0, nt = 0;
	if (o->m_bodyType == CollisionDetection::CollisionObject::TetModelCollisionObjectType)
	{
		TetModel *tm = m->getTetModels()[o->m_bodyIndex];
		nv = tm->getParticleMesh().numVertices(); nt = tm->getParticleMesh().numTets();
	}
	else if (o->m_bodyType == CollisionDetection::CollisionObject::TriangleModelCollisionObjectType)
		nv = m->getTriangleModels()[o->m_bodyIndex]->getParticleMesh().numVertices();
	unsigned nNodes = 0;
	if (which == 0) { nNodes = dumpBvh(o->m_bvh, nv, lst, lstCap, nodes, hulls, nodeCap); if (numEntities) *numEntities = nv; }
	else if (which == 1) { nNodes = dumpBvh(o->m_bvhTets, nt, lst, lstCap, nodes, hulls, nodeCap); if (numEntities) *numEntities = nt; }
	else { nNodes = dumpBvh(o->m_bvhTets0, nt, lst, lstCap, nodes, hulls, nodeCap); if (numEntities) *numEntities = nt; }
	return nNodes;
}
// Known-answer entry: the velocity part of ONE particle-tet contact through the reference's own functions
// (PositionBasedDynamics::init_ParticleTetContactConstraint, velocitySolve_ParticleTetContactConstraint).
// in (26): invMass0, v0[3], invMass[4], v[4][3], bary[3], normal[3]; `lambda` stands for the multiplier the reference reads unset (any finite value
// gives the same result when friction == 0); out (20): tangent[3], pMax, result flag, corr_v0[3], corr_v[4][3] (corrections the wrapper would not
// apply -- static particles -- are reported as 0, like ParticleTetContactConstraint::solveVelocityConstraint skips them)
void refdrv_kat_tet_contact_velocity(const double *in, double friction, double lambda, double *out)
{
	const Real invMass0 = (Real)in[0];
	const Vector3r v0((Real)in[1], (Real)in[2], (Real)in[3]);
	Real invMass[4];
	Vector3r x[4], v[4];
	for (int k = 0; k < 4; k++) { invMass[k] = (Real)in[4 + k]; v[k] = Vector3r((Real)in[8 + 3 * k], (Real)in[9 + 3 * k], (Real)in[10 + 3 * k]); x[k].setZero(); }
	const Vector3r bary((Real)in[20], (Real)in[21], (Real)in[22]), normal((Real)in[23], (Real)in[24], (Real)in[25]);
	const Vector3r x0(0, 0, 0);
	Eigen::Matrix<Real, 3, 3, Eigen::DontAlign> info;
	info.setZero();
	PositionBasedDynamics::init_ParticleTetContactConstraint(invMass0, x0, v0, invMass, x, v, bary, normal, info);
	for (int k = 0; k < 20; k++) out[k] = 0.0;
	for (int k = 0; k < 3; k++) out[k] = (double)info(k, 1);
	out[3] = (double)info(1, 2);
	Vector3r corr0(0, 0, 0), corr[4];
	for (int k = 0; k < 4; k++) corr[k].setZero();
	const bool res = PositionBasedDynamics::velocitySolve_ParticleTetContactConstraint(invMass0, x0, v0, invMass, x, v, bary, (Real)lambda, (Real)friction, info, corr0, corr);
	out[4] = res ? 1.0 : 0.0;
	if (res)
	{
		if (invMass0 != 0.0) for (int k = 0; k < 3; k++) out[5 + k] = (double)corr0[k];
		for (int q = 0; q < 4; q++) if (invMass[q] != 0.0) for (int k = 0; k < 3; k++) out[8 + 3 * q + k] = (double)corr[q][k];
	}
}
// run ONLY the collision detection on the current state (fills the contact lists; no velocity solve)
void refdrv_collision_detection_only() { cd().collisionDetection(*model()); }

double refdrv_contact_stiffness_particle_rigid_body() { return (double)model()->getContactStiffnessParticleRigidBody(); }

unsigned refdrv_num_particle_rigid_body_contacts() { return (unsigned)model()->getParticleRigidBodyContactConstraints().size(); }
// contact i: out[0] particle, out[1] rigid body (as doubles), out[2..16] constraintInfo (3x5, column-major), out[17] sum of impulses
void refdrv_get_particle_rigid_body_contact(unsigned i, double *out)
{
	ParticleRigidBodyContactConstraint &c = model()->getParticleRigidBodyContactConstraints()[i];
	out[0] = c.m_bodies[0]; out[1] = c.m_bodies[1];
	for (int col = 0; col < 5; col++) for (int r = 0; r < 3; r++) out[2 + 3 * col + r] = (double)c.m_constraintInfo(r, col);
	out[17] = (double)c.m_sum_impulses;
}
void refdrv_set_max_iterations_v(unsigned n) { model(); tsc()->setValue<unsigned int>(TimeStepController::MAX_ITERATIONS_V, n); }

// The currently installed TimeStep object (opaque; for plug-in side counters).
void *refdrv_get_timestep() { model(); return (void*)Simulation::getCurrent()->getTimeStep(); }
void *refdrv_get_model() { return (void*)model(); }
// the run-time parameter edits a demo GUI makes between steps (SimulationModel::setClothStiffness / setClothBendingStiffness,
// SimulationModel.cpp -> setConstraintValue<>): every constraint of the scene keeps its topology, only m_stiffness changes
void refdrv_set_cloth_stiffness(double k) { model()->setClothStiffness((Real)k); }
void refdrv_set_cloth_bending_stiffness(double k) { model()->setClothBendingStiffness((Real)k); }
// constraint classes outside the engine's scope (mixed models): the reference's generic constraints
// (Demos/GenericConstraintsDemos/GenericConstraints.cpp; GenericConstraintsModel::addGeneric* without the demo's model class)
int refdrv_add_generic_distance_constraint(unsigned p1, unsigned p2, double stiffness)
{
	GenericDistanceConstraint *c = new GenericDistanceConstraint();
	if (!c->initConstraint(*model(), p1, p2, (Real)stiffness)) { delete c; return 1; }
	model()->getConstraints().push_back(c);
	model()->m_groupsInitialized = false;
	return 0;
}
// A USER SUBCLASS that overrides the per-substep hook (Constraint::initConstraintBeforeProjection, called for every constraint at the start of every
// positionConstraintProjection: TimeStepController.cpp:264-268): its rest length follows the state the hook sees -- the integrated velocity and the
// distance travelled since oldX of its first particle -- so a host that calls the hook at another point of the substep, or not at all, or on stale
// velocities / old positions, gets different positions (tests/test_plugin.py: mixed models).
static unsigned g_hooked_calls = 0;
namespace {
class HookedDistanceConstraint : public GenericDistanceConstraint
{
public:
	Real m_base = 0;
	bool initConstraintBeforeProjection(SimulationModel &m) override
	{
		ParticleData &pd = m.getParticles();
		const Vector3r d = pd.getPosition(m_bodies[0]) - pd.getOldPosition(m_bodies[0]);
		// (bounded: the rest length stays within [1, 1.15] x its initial value, the scene stays stable)
		Real sv = pd.getVelocity(m_bodies[0]).norm(), sd = d.norm();
		if (sv > (Real)1.0) sv = (Real)1.0;
		if (sd > (Real)0.1) sd = (Real)0.1;
		m_restLength = m_base * ((Real)1.0 + (Real)0.05 * sv + sd);
		g_hooked_calls++;
		return true;
	}
};
}
int refdrv_add_hooked_distance_constraint(unsigned p1, unsigned p2, double stiffness)
{
	HookedDistanceConstraint *c = new HookedDistanceConstraint();
	if (!c->initConstraint(*model(), p1, p2, (Real)stiffness)) { delete c; return 1; }
	c->m_base = c->m_restLength;
	model()->getConstraints().push_back(c);
	model()->m_groupsInitialized = false;
	return 0;
}
unsigned refdrv_hooked_calls() { return g_hooked_calls; }
void refdrv_reset_hooked_calls() { g_hooked_calls = 0; }
int refdrv_add_generic_isometric_bending_constraint(unsigned p1, unsigned p2, unsigned p3, unsigned p4, double stiffness)
{
	GenericIsometricBendingConstraint *c = new GenericIsometricBendingConstraint();
	if (!c->initConstraint(*model(), p1, p2, p3, p4, (Real)stiffness)) { delete c; return 1; }
	model()->getConstraints().push_back(c);
	model()->m_groupsInitialized = false;
	return 0;
}
// edit of ONE constraint's stiffness behind the model's back (python: constraint.stiffness = ...)
void refdrv_set_constraint_stiffness(unsigned ci, double k)
{
	Constraint *c = model()->getConstraints()[ci];
	switch (mapType(c))
	{
	case 0: ((DistanceConstraint*)c)->m_stiffness = (Real)k; break;
	case 1: ((DistanceConstraint_XPBD*)c)->m_stiffness = (Real)k; break;
	case 3: ((IsometricBendingConstraint*)c)->m_stiffness = (Real)k; break;
	case 4: ((IsometricBendingConstraint_XPBD*)c)->m_stiffness = (Real)k; break;
	default: break;
	}
}

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
