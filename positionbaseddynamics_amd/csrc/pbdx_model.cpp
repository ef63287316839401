// pbdx_model.cpp -- host-side mirror of PBD::SimulationModel for particle scenes.
//
// Written from scratch against the behaviour of the reference:
//   mesh builders            Simulation/SimulationModel.cpp:806-1005
//   edge enumeration order   Utils/IndexedFaceMesh.cpp:118-226, Utils/IndexedTetMesh.cpp:55-182
//   per-constraint init      Simulation/Constraints.cpp:1166-2000 (+ init_* in PositionBasedDynamics.cpp)
//   bulk builders            Simulation/SimulationModel.cpp:1125-1349
//   greedy colouring         Simulation/SimulationModel.cpp:1033-1094
// Constraint creation order and the colouring are integer-exact replicas (the
// Gauss-Seidel order of the solver depends on them); rest data is computed in
// fp32 with the reference's operation order.  The colouring uses per-particle
// group bitsets (O(#constraints)) instead of the reference's per-group byte
// maps (O(#constraints x #groups)); the first-fit result is identical.
#include "pbdx_internal.h"
#include "pbdx_vec.h"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>

using namespace pbdx;

namespace {

const float kEps = 1e-6f;

// reserve room for `extra` more elements WITHOUT defeating the vector's geometric growth (an exact
// reserve per bulk builder call made a model of K instances cost O(K^2) to build)
template <class T> inline void grow_for(std::vector<T> &v, size_t extra)
{
	const size_t need = v.size() + extra;
	if (need > v.capacity()) v.reserve(need > 2 * v.capacity() ? need : 2 * v.capacity());
}

template <class A> inline V3 ld(const A &a, uint32_t i) { return mk(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
template <class A> inline void st(A &a, uint32_t i, V3 v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }

uint32_t add_vertex(pbdx_model *m, V3 p)
{
	// ParticleData::addVertex  ParticleData.h:128-138
	for (ParticleArray *arr : { &m->x0, &m->x, &m->old_x, &m->last_x })
	{
		arr->push_back(p.x); arr->push_back(p.y); arr->push_back(p.z);
	}
	for (ParticleArray *arr : { &m->v, &m->a })
	{
		arr->push_back(0.0f); arr->push_back(0.0f); arr->push_back(0.0f);
	}
	m->mass.push_back(1.0f);
	m->inv_mass.push_back(1.0f);
	return m->size() - 1;
}

void set_mass(pbdx_model *m, uint32_t i, float mass)
{
	m->mass[i] = mass;
	m->inv_mass[i] = (mass != 0.0f) ? 1.0f / mass : 0.0f;
}

// IndexedFaceMesh::buildNeighbors: faces in order, per face the edges
// (v0,v1),(v1,v2),(v2,v0); an edge is created the first time its vertex pair is
// seen (keeping that orientation and face[0]); later faces overwrite face[1].
void build_tri_edges(TriMesh &tm)
{
	std::unordered_map<uint64_t, uint32_t> lookup;
	const uint32_t nf = (uint32_t)tm.faces.size() / 3;
	lookup.reserve((size_t)nf * 2);
	tm.edges.clear();
	tm.edges.reserve((size_t)nf * 2);
	for (uint32_t f = 0; f < nf; f++)
	{
		const uint32_t *v = &tm.faces[3 * f];
		for (int j = 0; j < 3; j++)
		{
			const uint32_t a = v[j], b = v[(j + 1) % 3];
			const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
			auto it = lookup.find(key);
			if (it == lookup.end())
			{
				TriMesh::Edge e;
				e.vert[0] = a; e.vert[1] = b; e.face[0] = f; e.face[1] = 0xffffffffu;
				lookup.emplace(key, (uint32_t)tm.edges.size());
				tm.edges.push_back(e);
			}
			else
				tm.edges[it->second].face[1] = f;
		}
	}
}

// IndexedTetMesh::buildNeighbors: per tet the edges {01,02,03,12,13,23}, first seen wins.
void build_tet_edges(TetMesh &tm)
{
	std::unordered_map<uint64_t, uint32_t> lookup;
	const uint32_t nt = (uint32_t)tm.tets.size() / 4;
	lookup.reserve((size_t)nt * 2);
	tm.edges.clear();
	tm.vertex_tet_count.assign(tm.num_vertices, 0);
	static const int E[6][2] = { { 0, 1 }, { 0, 2 }, { 0, 3 }, { 1, 2 }, { 1, 3 }, { 2, 3 } };
	for (uint32_t t = 0; t < nt; t++)
	{
		const uint32_t *v = &tm.tets[4 * t];
		for (int j = 0; j < 4; j++) tm.vertex_tet_count[v[j]]++;
		for (int j = 0; j < 6; j++)
		{
			const uint32_t a = v[E[j][0]], b = v[E[j][1]];
			const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
			if (lookup.find(key) == lookup.end())
			{
				TetMesh::Edge e; e.vert[0] = a; e.vert[1] = b;
				lookup.emplace(key, (uint32_t)tm.edges.size());
				tm.edges.push_back(e);
			}
		}
	}
}

void push_constraint(pbdx_model *m, const HostConstraint &c)
{
	m->constraints.push_back(c);
	m->groups_initialized = false;
	m->topology_version++;
}

bool check_particles(pbdx_model *m, const uint32_t *p, uint32_t n)
{
	for (uint32_t i = 0; i < n; i++)
		if (p[i] >= m->size()) { set_error("particle index %u out of range (%u particles)", p[i], m->size()); return false; }
	return true;
}

float cot_theta(V3 v, V3 w)
{
	// MathFunctions::cotTheta  MathFunctions.cpp:391-396
	const float cosTheta = dot(v, w);
	const float sinTheta = norm(cross(v, w));
	return cosTheta / sinTheta;
}

// init_IsometricBendingConstraint  PositionBasedDynamics.cpp:145-183
void init_isometric_Q(V3 p0, V3 p1, V3 p2, V3 p3, float Q[16] /*col-major*/)
{
	const V3 x[4] = { p2, p3, p0, p1 };
	const V3 e0 = x[1] - x[0];
	const V3 e1 = x[2] - x[0];
	const V3 e2 = x[3] - x[0];
	const V3 e3 = x[2] - x[1];
	const V3 e4 = x[3] - x[1];
	const float c01 = cot_theta(e0, e1);
	const float c02 = cot_theta(e0, e2);
	const float c03 = cot_theta(-e0, e3);
	const float c04 = cot_theta(-e0, e4);
	const float A0 = 0.5f * norm(cross(e0, e1));
	const float A1 = 0.5f * norm(cross(e0, e2));
	const float coef = -3.f / (2.f * (A0 + A1));
	const float K[4] = { c03 + c04, c01 + c02, -c01 - c03, -c02 - c04 };
	const float K2[4] = { coef * K[0], coef * K[1], coef * K[2], coef * K[3] };
	for (int j = 0; j < 4; j++)
	{
		for (int k = 0; k < j; k++)
			Q[k * 4 + j] = Q[j * 4 + k] = K[j] * K2[k];
		Q[j * 4 + j] = K[j] * K2[j];
	}
}

// rest positions of the regular mesh builders (SimulationModel.cpp:840-862, 932-958): R * (grid point) + T, in float
void regular_tri_points(int width, int height, const M3 &rot, V3 t, const float scale[2], float *points)
{
	const float dy = scale[1] / (float)(height - 1);
	const float dx = scale[0] / (float)(width - 1);
	for (int i = 0; i < height; i++)
		for (int j = 0; j < width; j++)
		{
			const float y = dy * i;
			const float x = dx * j;
			const V3 p = mul(rot, mk(x, y, 0.0f)) + t;
			const size_t k = (size_t)i * width + j;
			points[3 * k] = p.x; points[3 * k + 1] = p.y; points[3 * k + 2] = p.z;
		}
}
void regular_tet_points(int width, int height, int depth, const M3 &rot, V3 T, const float scale[3], float *points)
{
	const float dx = scale[0] / (float)(width - 1);
	const float dy = scale[1] / (float)(height - 1);
	const float dz = scale[2] / (float)(depth - 1);
	const V3 t = mk(T.x - 0.5f * scale[0], T.y - 0.5f * scale[1], T.z - 0.5f * scale[2]);
	for (int i = 0; i < width; i++)
		for (int j = 0; j < height; j++)
			for (int k = 0; k < depth; k++)
			{
				const float x = dx * i, y = dy * j, z = dz * k;
				const V3 p = mul(rot, mk(x, y, z)) + t;
				const size_t q = (size_t)i * height * depth + (size_t)j * depth + k;
				points[3 * q] = p.x; points[3 * q + 1] = p.y; points[3 * q + 2] = p.z;
			}
}

M3 from_cols(V3 c0, V3 c1, V3 c2)
{
	M3 A;
	A.m[0][0] = c0.x; A.m[1][0] = c0.y; A.m[2][0] = c0.z;
	A.m[0][1] = c1.x; A.m[1][1] = c1.y; A.m[2][1] = c1.z;
	A.m[0][2] = c2.x; A.m[1][2] = c2.y; A.m[2][2] = c2.z;
	return A;
}
void store_colmajor(const M3 &A, float *out)
{
	for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) out[c * 3 + r] = A.m[r][c];
}

} // namespace

uint64_t pbdx::next_model_uid() { static std::atomic<uint64_t> n(1); return n.fetch_add(1); }
// the models alive in this process by their never-reused uid: holders of a model's IDENTITY (pbdx_ensemble: its blocks are copies) ask whether it still
// exists instead of keeping its address
namespace {
std::mutex g_models_mu;
std::unordered_map<uint64_t, const pbdx_model *> &live_models() { static std::unordered_map<uint64_t, const pbdx_model *> m; return m; }
}
const pbdx_model *pbdx::find_model(uint64_t uid)
{
	std::lock_guard<std::mutex> lk(g_models_mu);
	auto it = live_models().find(uid);
	return it == live_models().end() ? nullptr : it->second;
}

extern "C" {

int pbdx_model_create(pbdx_model **out)
{
	if (!out) { set_error("pbdx_model_create: null out"); return PBDX_ERR_INVALID; }
	*out = new (std::nothrow) pbdx_model();
	if (!*out) { set_error("out of memory"); return PBDX_ERR_ALLOC; }
	{ std::lock_guard<std::mutex> lk(g_models_mu); live_models()[(*out)->uid] = *out; }
	return PBDX_OK;
}

void pbdx_model_destroy(pbdx_model *m)
{
	if (!m) return;
	{ std::lock_guard<std::mutex> lk(g_models_mu); live_models().erase(m->uid); }
	delete m;
}

int pbdx_model_cleanup(pbdx_model *m)
{
	if (!m) return PBDX_ERR_INVALID;
	m->mass.clear(); m->inv_mass.clear();
	m->x0.clear(); m->x.clear(); m->v.clear(); m->a.clear(); m->old_x.clear(); m->last_x.clear();
	m->tri_models.clear(); m->tet_models.clear(); m->constraints.clear(); m->groups.clear();
	m->inst_count = 1; m->inst_particles = 0; m->inst_offset.clear();
	m->groups_initialized = false;
	m->topology_version++;
	return PBDX_OK;
}

int pbdx_model_reset(pbdx_model *m)
{
	if (!m) return PBDX_ERR_INVALID;
	m->x = m->x0; m->last_x = m->x0; m->old_x = m->x0;
	std::fill(m->v.begin(), m->v.end(), 0.0f);
	std::fill(m->a.begin(), m->a.end(), 0.0f);
	m->params_version++;
	m->state_version++;              // the host state is authoritative again (a device-resident image is stale)
	m->dirty_arrays |= 0x3fu;
	return PBDX_OK;
}

int pbdx_model_add_triangle_model(pbdx_model *m, uint32_t n_points, uint32_t n_faces, const float *points, const uint32_t *indices)
{
	if (!m || !points || !indices) { set_error("add_triangle_model: null argument"); return -1; }
	if (m->inst_count > 1) { set_error("the model holds instances (pbdx_model_add_instances): nothing can be added any more"); return -1; }
	for (uint32_t i = 0; i < 3 * n_faces; i++)
		if (indices[i] >= n_points) { set_error("add_triangle_model: face index out of range"); return -1; }
	TriMesh tm;
	tm.index_offset = m->size();
	tm.num_vertices = n_points;
	for (uint32_t i = 0; i < n_points; i++) add_vertex(m, mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]));
	tm.faces.assign(indices, indices + 3 * (size_t)n_faces);
	build_tri_edges(tm);
	m->tri_models.push_back(std::move(tm));
	m->topology_version++;
	return (int)m->tri_models.size() - 1;
}

int pbdx_model_add_regular_triangle_model(pbdx_model *m, int width, int height,
	const float T[3], const float R[9], const float scale[2])
{
	if (!m || width < 2 || height < 2) { set_error("add_regular_triangle_model: need width,height >= 2"); return -1; }
	static const float I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
	static const float Z[3] = { 0, 0, 0 };
	static const float O[2] = { 1, 1 };
	if (!R) R = I;
	if (!T) T = Z;
	if (!scale) scale = O;
	M3 rot;
	for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rot.m[r][c] = R[3 * r + c];
	const V3 t = mk(T[0], T[1], T[2]);
	const float dy = scale[1] / (float)(height - 1);
	const float dx = scale[0] / (float)(width - 1);
	std::vector<float> points((size_t)width * height * 3);
	regular_tri_points(width, height, rot, t, scale, points.data());
	(void)dx; (void)dy;
	std::vector<uint32_t> indices((size_t)6 * (height - 1) * (width - 1));
	size_t index = 0;
	for (int i = 0; i < height - 1; i++)
		for (int j = 0; j < width - 1; j++)
		{
			const int helper = (i % 2 == j % 2) ? 1 : 0;
			indices[index] = i * width + j;
			indices[index + 1] = i * width + j + 1;
			indices[index + 2] = (i + 1) * width + j + helper;
			index += 3;
			indices[index] = (i + 1) * width + j + 1;
			indices[index + 1] = (i + 1) * width + j;
			indices[index + 2] = i * width + j + 1 - helper;
			index += 3;
		}
	const int res = pbdx_model_add_triangle_model(m, (uint32_t)(width * height), (uint32_t)(indices.size() / 3), points.data(), indices.data());
	if (res < 0) return res;
	{
		MeshRecipe &rc = m->tri_models[res].recipe;
		rc.regular = 1; rc.dims[0] = width; rc.dims[1] = height;
		memcpy(rc.R, R, sizeof(rc.R)); memcpy(rc.T, T, sizeof(rc.T)); rc.scale[0] = scale[0]; rc.scale[1] = scale[1];
	}
	const uint32_t off = m->tri_models[res].index_offset;
	for (uint32_t i = off; i < off + m->tri_models[res].num_vertices; i++) set_mass(m, i, 1.0f);
	return res;
}

int pbdx_model_add_tet_model(pbdx_model *m, uint32_t n_points, uint32_t n_tets, const float *points, const uint32_t *indices)
{
	if (!m || !points || !indices) { set_error("add_tet_model: null argument"); return -1; }
	if (m->inst_count > 1) { set_error("the model holds instances (pbdx_model_add_instances): nothing can be added any more"); return -1; }
	for (uint32_t i = 0; i < 4 * n_tets; i++)
		if (indices[i] >= n_points) { set_error("add_tet_model: tet index out of range"); return -1; }
	TetMesh tm;
	tm.index_offset = m->size();
	tm.num_vertices = n_points;
	for (uint32_t i = 0; i < n_points; i++) add_vertex(m, mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]));
	tm.tets.assign(indices, indices + 4 * (size_t)n_tets);
	build_tet_edges(tm);
	m->tet_models.push_back(std::move(tm));
	m->topology_version++;
	return (int)m->tet_models.size() - 1;
}

int pbdx_model_add_regular_tet_model(pbdx_model *m, int width, int height, int depth,
	const float T[3], const float R[9], const float scale[3])
{
	if (!m || width < 2 || height < 2 || depth < 2) { set_error("add_regular_tet_model: need all dims >= 2"); return -1; }
	static const float I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
	static const float Z[3] = { 0, 0, 0 };
	static const float O[3] = { 1, 1, 1 };
	if (!R) R = I;
	if (!T) T = Z;
	if (!scale) scale = O;
	M3 rot;
	for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rot.m[r][c] = R[3 * r + c];
	const float dx = scale[0] / (float)(width - 1);
	const float dy = scale[1] / (float)(height - 1);
	const float dz = scale[2] / (float)(depth - 1);
	const V3 t = mk(T[0] - 0.5f * scale[0], T[1] - 0.5f * scale[1], T[2] - 0.5f * scale[2]);
	std::vector<float> points((size_t)width * height * depth * 3);
	regular_tet_points(width, height, depth, rot, mk(T[0], T[1], T[2]), scale, points.data());
	(void)dx; (void)dy; (void)dz; (void)t;
	std::vector<uint32_t> idx;
	idx.reserve((size_t)(width - 1) * (height - 1) * (depth - 1) * 20);
	for (int i = 0; i < width - 1; i++)
		for (int j = 0; j < height - 1; j++)
			for (int k = 0; k < depth - 1; k++)
			{
				const uint32_t p0 = i * height * depth + j * depth + k;
				const uint32_t p1 = p0 + 1;
				const uint32_t p3 = (i + 1) * height * depth + j * depth + k;
				const uint32_t p2 = p3 + 1;
				const uint32_t p7 = (i + 1) * height * depth + (j + 1) * depth + k;
				const uint32_t p6 = p7 + 1;
				const uint32_t p4 = i * height * depth + (j + 1) * depth + k;
				const uint32_t p5 = p4 + 1;
				if ((i + j + k) % 2 == 1)
				{
					const uint32_t t5[20] = { p2, p1, p6, p3,  p6, p3, p4, p7,  p4, p1, p6, p5,  p3, p1, p4, p0,  p6, p1, p4, p3 };
					idx.insert(idx.end(), t5, t5 + 20);
				}
				else
				{
					const uint32_t t5[20] = { p0, p2, p5, p1,  p7, p2, p0, p3,  p5, p2, p7, p6,  p7, p0, p5, p4,  p0, p2, p7, p5 };
					idx.insert(idx.end(), t5, t5 + 20);
				}
			}
	const int res = pbdx_model_add_tet_model(m, (uint32_t)(width * height * depth), (uint32_t)(idx.size() / 4), points.data(), idx.data());
	if (res < 0) return res;
	{
		MeshRecipe &rc = m->tet_models[res].recipe;
		rc.regular = 1; rc.dims[0] = width; rc.dims[1] = height; rc.dims[2] = depth;
		memcpy(rc.R, R, sizeof(rc.R)); memcpy(rc.T, T, sizeof(rc.T)); memcpy(rc.scale, scale, sizeof(rc.scale));
	}
	const uint32_t off = m->tet_models[res].index_offset;
	for (uint32_t i = off; i < off + m->tet_models[res].num_vertices; i++) set_mass(m, i, 1.0f);
	return res;
}

// Mesh model `tm` of a (possibly instanced) model: instance k's models are numbered k * (#prototype models) + j and share the
// prototype's topology; their particles start k * inst_particles later.
static const TriMesh *tri_of(const pbdx_model *m, uint32_t tm, uint32_t *offset)
{
	if (!m || tm >= m->num_tri_models()) return nullptr;
	const uint32_t np = (uint32_t)m->tri_models.size();
	const TriMesh &t = m->tri_models[tm % np];
	if (offset) *offset = t.index_offset + (tm / np) * m->inst_particles;
	return &t;
}
static const TetMesh *tet_of(const pbdx_model *m, uint32_t tm, uint32_t *offset)
{
	if (!m || tm >= m->num_tet_models()) return nullptr;
	const uint32_t np = (uint32_t)m->tet_models.size();
	const TetMesh &t = m->tet_models[tm % np];
	if (offset) *offset = t.index_offset + (tm / np) * m->inst_particles;
	return &t;
}
uint32_t pbdx_model_num_triangle_models(const pbdx_model *m) { return m ? m->num_tri_models() : 0; }
uint32_t pbdx_model_num_tet_models(const pbdx_model *m) { return m ? m->num_tet_models() : 0; }
uint32_t pbdx_model_triangle_model_index_offset(const pbdx_model *m, uint32_t tm) { uint32_t o = 0; return tri_of(m, tm, &o) ? o : 0; }
uint32_t pbdx_model_tet_model_index_offset(const pbdx_model *m, uint32_t tm) { uint32_t o = 0; return tet_of(m, tm, &o) ? o : 0; }
uint32_t pbdx_model_triangle_model_num_edges(const pbdx_model *m, uint32_t tm) { const TriMesh *t = tri_of(m, tm, nullptr); return t ? (uint32_t)t->edges.size() : 0; }
int pbdx_model_triangle_model_get_edges(const pbdx_model *m, uint32_t tm, uint32_t *out)
{
	const TriMesh *t = tri_of(m, tm, nullptr);
	if (!t || !out) { set_error("triangle_model_get_edges: bad argument"); return PBDX_ERR_INVALID; }
	const auto &e = t->edges;
	for (size_t i = 0; i < e.size(); i++)
	{
		out[4 * i] = e[i].vert[0]; out[4 * i + 1] = e[i].vert[1]; out[4 * i + 2] = e[i].face[0]; out[4 * i + 3] = e[i].face[1];
	}
	return PBDX_OK;
}
// faces / tets / vertex counts (IndexedFaceMesh::numFaces/getFaces, IndexedTetMesh::numTets/getTets)
uint32_t pbdx_model_triangle_model_num_vertices(const pbdx_model *m, uint32_t tm) { const TriMesh *t = tri_of(m, tm, nullptr); return t ? t->num_vertices : 0; }
uint32_t pbdx_model_triangle_model_num_faces(const pbdx_model *m, uint32_t tm) { const TriMesh *t = tri_of(m, tm, nullptr); return t ? (uint32_t)(t->faces.size() / 3) : 0; }
int pbdx_model_triangle_model_get_faces(const pbdx_model *m, uint32_t tm, uint32_t *out)
{
	const TriMesh *t = tri_of(m, tm, nullptr);
	if (!t || !out) { set_error("triangle_model_get_faces: bad argument"); return PBDX_ERR_INVALID; }
	memcpy(out, t->faces.data(), t->faces.size() * sizeof(uint32_t));
	return PBDX_OK;
}
uint32_t pbdx_model_tet_model_num_vertices(const pbdx_model *m, uint32_t tm) { const TetMesh *t = tet_of(m, tm, nullptr); return t ? t->num_vertices : 0; }
uint32_t pbdx_model_tet_model_num_tets(const pbdx_model *m, uint32_t tm) { const TetMesh *t = tet_of(m, tm, nullptr); return t ? (uint32_t)(t->tets.size() / 4) : 0; }
int pbdx_model_tet_model_get_tets(const pbdx_model *m, uint32_t tm, uint32_t *out)
{
	const TetMesh *t = tet_of(m, tm, nullptr);
	if (!t || !out) { set_error("tet_model_get_tets: bad argument"); return PBDX_ERR_INVALID; }
	memcpy(out, t->tets.data(), t->tets.size() * sizeof(uint32_t));
	return PBDX_OK;
}

uint32_t pbdx_model_tet_model_num_edges(const pbdx_model *m, uint32_t tm) { const TetMesh *t = tet_of(m, tm, nullptr); return t ? (uint32_t)t->edges.size() : 0; }
int pbdx_model_tet_model_get_edges(const pbdx_model *m, uint32_t tm, uint32_t *out)
{
	const TetMesh *t = tet_of(m, tm, nullptr);
	if (!t || !out) { set_error("tet_model_get_edges: bad argument"); return PBDX_ERR_INVALID; }
	const auto &e = t->edges;
	for (size_t i = 0; i < e.size(); i++) { out[2 * i] = e[i].vert[0]; out[2 * i + 1] = e[i].vert[1]; }
	return PBDX_OK;
}

// ---- instances -------------------------------------------------------------------------------------
// Append `count` congruent copies of everything the model holds (the 512-instance ensemble is 512
// calls of the same builders with a different translation).  What the reference would have done for copy k -- call the
// mesh builders with translation T + offset_k, then the constraint builders, then re-colour everything -- is reproduced
// exactly, but stored once: rest positions are re-evaluated per instance with the builders' own formula (explicit-point
// models and loose particles: prototype position + offset), constraints and colour groups stay the prototype's and are
// expanded on demand (model_constraint, pbdx_model_get_group).  Set-up cost and memory no longer grow with the instance
// count except for the particle arrays.  Afterwards the model is sealed: no further add*.
int pbdx_model_add_instances(pbdx_model *m, uint32_t count, const float *offsets)
{
	if (!m || (count && !offsets)) { set_error("add_instances: null argument"); return PBDX_ERR_INVALID; }
	if (m->inst_count > 1) { set_error("add_instances: the model already holds instances"); return PBDX_ERR_INVALID; }
	if (!count) return PBDX_OK;
	const uint32_t np = m->size();
	if (!np) { set_error("add_instances: empty model"); return PBDX_ERR_INVALID; }
	if ((uint64_t)np * (count + 1) > 0xffffffffull || (uint64_t)m->constraints.size() * (count + 1) > 0xffffffffull)
	{ set_error("add_instances: more than 2^32 particles or constraints"); return PBDX_ERR_INVALID; }
	const uint32_t K = count + 1;
	ParticleArray x0((size_t)3 * np * K);
	memcpy(x0.data(), m->x0.data(), (size_t)3 * np * sizeof(float));
	for (uint32_t k = 1; k < K; k++)
	{
		const float *off = offsets + 3 * (size_t)(k - 1);
		float *dst = x0.data() + (size_t)3 * np * k;
		for (uint32_t i = 0; i < np; i++)
			for (int d = 0; d < 3; d++) dst[3 * i + d] = m->x0[3 * i + d] + off[d];
		for (const TriMesh &t : m->tri_models)
			if (t.recipe.regular)
			{
				M3 rot;
				for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rot.m[r][c] = t.recipe.R[3 * r + c];
				regular_tri_points(t.recipe.dims[0], t.recipe.dims[1], rot, mk(t.recipe.T[0] + off[0], t.recipe.T[1] + off[1], t.recipe.T[2] + off[2]),
					t.recipe.scale, dst + (size_t)3 * t.index_offset);
			}
		for (const TetMesh &t : m->tet_models)
			if (t.recipe.regular)
			{
				M3 rot;
				for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rot.m[r][c] = t.recipe.R[3 * r + c];
				regular_tet_points(t.recipe.dims[0], t.recipe.dims[1], t.recipe.dims[2], rot, mk(t.recipe.T[0] + off[0], t.recipe.T[1] + off[1], t.recipe.T[2] + off[2]),
					t.recipe.scale, dst + (size_t)3 * t.index_offset);
			}
	}
	// commit: particle state of the copies = their rest state; masses as the prototype's
	ParticleArray mass((size_t)np * K), inv((size_t)np * K);
	for (uint32_t k = 0; k < K; k++)
	{
		memcpy(mass.data() + (size_t)np * k, m->mass.data(), (size_t)np * sizeof(float));
		memcpy(inv.data() + (size_t)np * k, m->inv_mass.data(), (size_t)np * sizeof(float));
	}
	auto extend = [&](ParticleArray &a, bool zero)
	{
		a.resize((size_t)3 * np * K, 0.0f);
		if (!zero) memcpy(a.data() + (size_t)3 * np, x0.data() + (size_t)3 * np, (size_t)3 * np * (K - 1) * sizeof(float));
	};
	extend(m->x, false); extend(m->old_x, false); extend(m->last_x, false); extend(m->v, true); extend(m->a, true);
	m->x0.swap(x0);
	m->mass.swap(mass); m->inv_mass.swap(inv);
	m->inst_count = K;
	m->inst_particles = np;
	m->inst_offset.assign((size_t)3 * K, 0.0f);
	memcpy(m->inst_offset.data() + 3, offsets, (size_t)3 * count * sizeof(float));
	// every instance must be congruent to the prototype: an element that is degenerate in one copy only (the reference
	// would have dropped that one constraint) cannot be represented
	{
		const uint64_t nc = m->constraints.size(), total = m->num_constraints();
		std::atomic<uint64_t> bad(~0ull);
		auto scan = [&](uint64_t a, uint64_t b)
		{
			HostConstraint c;
			for (uint64_t ci = a; ci < b; ci++)
				if (!model_constraint(m, ci, c)) { uint64_t cur = bad.load(); while (ci < cur && !bad.compare_exchange_weak(cur, ci)) {} return; }
		};
		const uint32_t threads = (total - nc > 200000) ? std::min<uint32_t>(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
		if (threads <= 1) scan(nc, total);
		else
		{
			std::vector<std::thread> pool;
			for (uint32_t t = 0; t < threads; t++) pool.emplace_back(scan, nc + (total - nc) * t / threads, nc + (total - nc) * (t + 1) / threads);
			for (std::thread &t : pool) t.join();
		}
		if (bad.load() != ~0ull)
		{
			m->x0.resize((size_t)3 * np); m->x.resize((size_t)3 * np); m->old_x.resize((size_t)3 * np); m->last_x.resize((size_t)3 * np);
			m->v.resize((size_t)3 * np); m->a.resize((size_t)3 * np); m->mass.resize(np); m->inv_mass.resize(np);
			m->inst_count = 1; m->inst_particles = 0; m->inst_offset.clear();
			set_error("add_instances: constraint %llu of the prototype is degenerate in a copy: the instances are not congruent", (unsigned long long)(bad.load() % nc));
			return PBDX_ERR_INVALID;
		}
	}
	m->topology_version++;
	m->state_version++;
	m->dirty_arrays |= 0x3fu;
	return PBDX_OK;
}
uint32_t pbdx_model_num_instances(const pbdx_model *m) { return m ? m->inst_count : 0; }

uint32_t pbdx_model_num_particles(const pbdx_model *m) { return m ? m->size() : 0; }

int pbdx_model_add_vertex(pbdx_model *m, const float x[3])
{
	if (!m || !x) return -1;
	if (m->inst_count > 1) { set_error("the model holds instances (pbdx_model_add_instances): nothing can be added any more"); return -1; }
	m->topology_version++;
	return (int)add_vertex(m, mk(x[0], x[1], x[2]));
}

int pbdx_model_set_mass(pbdx_model *m, uint32_t i, float mass)
{
	if (!m || i >= m->size()) { set_error("set_mass: index out of range"); return PBDX_ERR_INVALID; }
	set_mass(m, i, mass);
	m->params_version++;
	return PBDX_OK;
}

static ParticleArray *model_array(pbdx_model *m, int which)
{
	switch (which)
	{
	case 0: return &m->x; case 1: return &m->x0; case 2: return &m->v; case 3: return &m->a;
	case 4: return &m->old_x; case 5: return &m->last_x; case 6: return &m->mass; case 7: return &m->inv_mass;
	default: return nullptr;
	}
}

int pbdx_model_get_array(const pbdx_model *m, int which, float *out)
{
	if (!m || !out) return PBDX_ERR_INVALID;
	const ParticleArray *a = model_array(const_cast<pbdx_model*>(m), which);
	if (!a) { set_error("get_array: bad selector %d", which); return PBDX_ERR_INVALID; }
	memcpy(out, a->data(), a->size() * sizeof(float));
	return PBDX_OK;
}

int pbdx_model_set_array(pbdx_model *m, int which, const float *in)
{
	if (!m || !in) return PBDX_ERR_INVALID;
	if (which == 6)
	{
		for (uint32_t i = 0; i < m->size(); i++) set_mass(m, i, in[i]);
		m->params_version++;
		return PBDX_OK;
	}
	ParticleArray *a = model_array(m, which);
	if (!a || which == 7) { set_error("set_array: bad selector %d", which); return PBDX_ERR_INVALID; }
	memcpy(a->data(), in, a->size() * sizeof(float));
	m->state_version++;
	m->dirty_arrays |= 1u << which;
	return PBDX_OK;
}

float *pbdx_model_positions_ptr(pbdx_model *m) { return m ? m->x.data() : nullptr; }
int pbdx_model_mark_state_dirty(pbdx_model *m) { if (!m) return PBDX_ERR_INVALID; m->state_version++; m->dirty_arrays |= 1u; return PBDX_OK; }

// ---- per-constraint builders -----------------------------------------------------------
// The rest data of a constraint (Constraints.cpp initConstraint + init_* in PositionBasedDynamics.cpp) is a function of the
// rest positions x0 of its particles only.  init_geometry fills those entries of c.params for c.type / c.bodies, leaving the
// user parameters (stiffness, Poisson ratio, flags ...) that the caller put there; false = the reference's initConstraint
// would have returned false (degenerate element).  The same function serves the add* calls and the instanced model
// (an instance's constraint = the prototype's with offset particles and ITS OWN rest data).
static float rest_volume(const pbdx_model *m, const uint32_t p[4])
{
	const V3 p0 = ld(m->x0, p[0]), p1 = ld(m->x0, p[1]), p2 = ld(m->x0, p[2]), p3 = ld(m->x0, p[3]);
	return fabsf((float)(1.0 / 6.0) * dot(p3 - p0, cross(p2 - p0, p1 - p0)));
}

static bool init_geometry(const pbdx_model *m, HostConstraint &c)
{
	const uint32_t *b = c.bodies;
	switch (c.type)
	{
	case PBDX_DISTANCE: case PBDX_DISTANCE_XPBD:
		c.params[0] = norm(ld(m->x0, b[1]) - ld(m->x0, b[0]));
		return true;
	case PBDX_DIHEDRAL:
	{
		const V3 p0 = ld(m->x0, b[0]), p1 = ld(m->x0, b[1]), p2 = ld(m->x0, b[2]), p3 = ld(m->x0, b[3]);
		const V3 e = p3 - p2;
		const float elen = norm(e);
		if ((double)elen < 1e-6)
			return false;
		V3 n1 = cross(p2 - p0, p3 - p0); n1 = n1 / sqn(n1);
		V3 n2 = cross(p3 - p1, p2 - p1); n2 = n2 / sqn(n2);
		n1 = normalized(n1);
		n2 = normalized(n2);
		float d = dot(n1, n2);
		if (d < -1.0f) d = -1.0f;
		if (d > 1.0f) d = 1.0f;
		c.params[0] = acosf(d);
		return true;
	}
	case PBDX_ISOMETRIC_BENDING: case PBDX_ISOMETRIC_BENDING_XPBD:
		init_isometric_Q(ld(m->x0, b[0]), ld(m->x0, b[1]), ld(m->x0, b[2]), ld(m->x0, b[3]), &c.params[1]);
		return true;
	case PBDX_FEM_TRIANGLE:
	{
		// init_FEMTriangleConstraint  PositionBasedDynamics.cpp:808-841
		const V3 p0 = ld(m->x0, b[0]), p1 = ld(m->x0, b[1]), p2 = ld(m->x0, b[2]);
		const V3 normal0 = cross(p1 - p0, p2 - p0);
		const float area = norm(normal0) * 0.5f;
		const V3 axis0_1 = normalized(p1 - p0);
		const V3 axis0_2 = normalized(cross(normal0, axis0_1));
		const float q[3][2] = { { dot(p0, axis0_2), dot(p0, axis0_1) }, { dot(p1, axis0_2), dot(p1, axis0_1) }, { dot(p2, axis0_2), dot(p2, axis0_1) } };
		const float P00 = q[0][0] - q[2][0], P10 = q[0][1] - q[2][1], P01 = q[1][0] - q[2][0], P11 = q[1][1] - q[2][1];
		const float det = P00 * P11 - P10 * P01;
		if (!(fabsf(det) > kEps))
			return false;
		const float invdet = 1.0f / det;
		c.params[0] = area;
		c.params[1] = P11 * invdet;    // (0,0)
		c.params[2] = -P10 * invdet;   // (1,0)
		c.params[3] = -P01 * invdet;   // (0,1)
		c.params[4] = P00 * invdet;    // (1,1)
		return true;
	}
	case PBDX_STRAIN_TRIANGLE:
	{
		// StrainTriangleConstraint::initConstraint flattens to the x-z plane (Constraints.cpp:1563-1568),
		// then init_StrainTriangleConstraint  PositionBasedDynamics.cpp:562-581
		const V3 x1 = ld(m->x0, b[0]), x2 = ld(m->x0, b[1]), x3 = ld(m->x0, b[2]);
		const float a = x2.x - x1.x, bb = x3.x - x1.x;
		const float cc = x2.z - x1.z, d = x3.z - x1.z;
		const float det = a * d - bb * cc;
		if (fabsf(det) < kEps)
			return false;
		const float s = 1.0f / det;
		c.params[0] = d * s;     // (0,0)
		c.params[1] = -cc * s;   // (1,0)
		c.params[2] = -bb * s;   // (0,1)
		c.params[3] = a * s;     // (1,1)
		return true;
	}
	case PBDX_VOLUME: case PBDX_VOLUME_XPBD:
		c.params[0] = rest_volume(m, b);
		return true;
	case PBDX_FEM_TET: case PBDX_FEM_TET_XPBD:
	{
		// init_FEMTetraConstraint  PositionBasedDynamics.cpp:933-955
		const V3 p0 = ld(m->x0, b[0]), p1 = ld(m->x0, b[1]), p2 = ld(m->x0, b[2]), p3 = ld(m->x0, b[3]);
		const M3 mat = from_cols(p0 - p3, p1 - p3, p2 - p3);
		const float dt = det(mat);
		if (!(fabsf(dt) > kEps))
			return false;
		c.params[0] = rest_volume(m, b);
		store_colmajor(inverse(mat), &c.params[1]);
		return true;
	}
	case PBDX_STRAIN_TET:
	{
		// init_StrainTetraConstraint  PositionBasedDynamics.cpp:691-710
		const V3 p0 = ld(m->x0, b[0]), p1 = ld(m->x0, b[1]), p2 = ld(m->x0, b[2]), p3 = ld(m->x0, b[3]);
		const M3 mat = from_cols(p1 - p0, p2 - p0, p3 - p0);
		const float dt = det(mat);
		if (!(fabsf(dt) > kEps))
			return false;
		store_colmajor(inverse(mat), &c.params[0]);
		return true;
	}
	case PBDX_SHAPE_MATCHING:
	{
		// init_ShapeMatchingConstraint  PositionBasedDynamics.cpp:481-498 (c.params[20..23] = numClusters set by the caller)
		V3 restCm = mk(0.0f, 0.0f, 0.0f);
		float wsum = 0.0f;
		for (int i = 0; i < 4; i++)
		{
			const V3 x0 = ld(m->x0, b[i]);
			const float w = m->inv_mass[b[i]];
			c.params[4 + 3 * i] = x0.x; c.params[5 + 3 * i] = x0.y; c.params[6 + 3 * i] = x0.z;
			c.params[16 + i] = w;
			const float wi = 1.0f / (w + kEps);
			restCm = restCm + x0 * wi;
			wsum += wi;
		}
		if (wsum == 0.0f)
			return false;
		restCm = restCm / wsum;
		c.params[1] = restCm.x; c.params[2] = restCm.y; c.params[3] = restCm.z;
		return true;
	}
	default:
		return false;
	}
}

// common tail of every add*Constraint: range check, rest data, append
static int add_constraint(pbdx_model *m, HostConstraint &c)
{
	if (!m) return 0;
	if (m->inst_count > 1) { set_error("the model holds instances (pbdx_model_add_instances): no constraints can be added any more"); return 0; }
	if (!check_particles(m, c.bodies, type_info(c.type)->num_bodies)) return 0;
	if (!init_geometry(m, c)) return 0;
	push_constraint(m, c);
	return 1;
}

int pbdx_model_add_distance_constraint(pbdx_model *m, uint32_t p1, uint32_t p2, float stiffness)
{
	HostConstraint c = {};
	c.type = PBDX_DISTANCE; c.bodies[0] = p1; c.bodies[1] = p2; c.params[1] = stiffness;
	return add_constraint(m, c);
}

int pbdx_model_add_distance_constraint_xpbd(pbdx_model *m, uint32_t p1, uint32_t p2, float stiffness)
{
	HostConstraint c = {};
	c.type = PBDX_DISTANCE_XPBD; c.bodies[0] = p1; c.bodies[1] = p2; c.params[1] = stiffness;
	return add_constraint(m, c);
}

int pbdx_model_add_dihedral_constraint(pbdx_model *m, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, float stiffness)
{
	HostConstraint c = {};
	c.type = PBDX_DIHEDRAL; c.bodies[0] = i0; c.bodies[1] = i1; c.bodies[2] = i2; c.bodies[3] = i3; c.params[1] = stiffness;
	return add_constraint(m, c);
}

static int add_isometric(pbdx_model *m, int type, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, float stiffness)
{
	HostConstraint c = {};
	c.type = type; c.bodies[0] = i0; c.bodies[1] = i1; c.bodies[2] = i2; c.bodies[3] = i3; c.params[0] = stiffness;
	return add_constraint(m, c);
}
int pbdx_model_add_isometric_bending_constraint(pbdx_model *m, uint32_t a, uint32_t b, uint32_t c, uint32_t d, float k)
{ return add_isometric(m, PBDX_ISOMETRIC_BENDING, a, b, c, d, k); }
int pbdx_model_add_isometric_bending_constraint_xpbd(pbdx_model *m, uint32_t a, uint32_t b, uint32_t c, uint32_t d, float k)
{ return add_isometric(m, PBDX_ISOMETRIC_BENDING_XPBD, a, b, c, d, k); }

int pbdx_model_add_fem_triangle_constraint(pbdx_model *m, uint32_t i0, uint32_t i1, uint32_t i2,
	float xx, float yy, float xy, float xyP, float yxP)
{
	HostConstraint c = {};
	c.type = PBDX_FEM_TRIANGLE; c.bodies[0] = i0; c.bodies[1] = i1; c.bodies[2] = i2;
	c.params[5] = xx; c.params[6] = yy; c.params[7] = xy; c.params[8] = xyP; c.params[9] = yxP;
	return add_constraint(m, c);
}

int pbdx_model_add_strain_triangle_constraint(pbdx_model *m, uint32_t i0, uint32_t i1, uint32_t i2,
	float xx, float yy, float xy, int ns, int nsh)
{
	HostConstraint c = {};
	c.type = PBDX_STRAIN_TRIANGLE; c.bodies[0] = i0; c.bodies[1] = i1; c.bodies[2] = i2;
	c.params[4] = xx; c.params[5] = yy; c.params[6] = xy; c.params[7] = ns ? 1.0f : 0.0f; c.params[8] = nsh ? 1.0f : 0.0f;
	return add_constraint(m, c);
}

static int add_volume(pbdx_model *m, int type, uint32_t a, uint32_t b, uint32_t c_, uint32_t d, float k)
{
	HostConstraint c = {};
	c.type = type; c.bodies[0] = a; c.bodies[1] = b; c.bodies[2] = c_; c.bodies[3] = d; c.params[1] = k;
	return add_constraint(m, c);
}
int pbdx_model_add_volume_constraint(pbdx_model *m, uint32_t a, uint32_t b, uint32_t c, uint32_t d, float k) { return add_volume(m, PBDX_VOLUME, a, b, c, d, k); }
int pbdx_model_add_volume_constraint_xpbd(pbdx_model *m, uint32_t a, uint32_t b, uint32_t c, uint32_t d, float k) { return add_volume(m, PBDX_VOLUME_XPBD, a, b, c, d, k); }

static int add_fem_tet(pbdx_model *m, int type, uint32_t a, uint32_t b, uint32_t c_, uint32_t d, float k, float nu)
{
	HostConstraint c = {};
	c.type = type; c.bodies[0] = a; c.bodies[1] = b; c.bodies[2] = c_; c.bodies[3] = d;
	c.params[10] = k; c.params[11] = nu;
	return add_constraint(m, c);
}
int pbdx_model_add_fem_tet_constraint(pbdx_model *m, uint32_t a, uint32_t b, uint32_t c, uint32_t d, float k, float nu) { return add_fem_tet(m, PBDX_FEM_TET, a, b, c, d, k, nu); }
int pbdx_model_add_fem_tet_constraint_xpbd(pbdx_model *m, uint32_t a, uint32_t b, uint32_t c, uint32_t d, float k, float nu) { return add_fem_tet(m, PBDX_FEM_TET_XPBD, a, b, c, d, k, nu); }

int pbdx_model_add_strain_tet_constraint(pbdx_model *m, uint32_t a, uint32_t b, uint32_t c_, uint32_t d,
	float stretch, float shear, int ns, int nsh)
{
	HostConstraint c = {};
	c.type = PBDX_STRAIN_TET; c.bodies[0] = a; c.bodies[1] = b; c.bodies[2] = c_; c.bodies[3] = d;
	c.params[9] = stretch; c.params[10] = shear; c.params[11] = ns ? 1.0f : 0.0f; c.params[12] = nsh ? 1.0f : 0.0f;
	return add_constraint(m, c);
}

int pbdx_model_add_shape_matching_constraint(pbdx_model *m, uint32_t n, const uint32_t *particles, const uint32_t *num_clusters, float stiffness)
{
	if (!m || !particles || !num_clusters) return 0;
	if (n != 4) { set_error("shape matching: only 4-particle clusters are on the path (addSolidConstraints method 5)"); return 0; }
	HostConstraint c = {};
	c.type = PBDX_SHAPE_MATCHING; memcpy(c.bodies, particles, 4 * sizeof(uint32_t));
	c.params[0] = stiffness;
	for (int i = 0; i < 4; i++) c.params[20 + i] = (float)num_clusters[i];
	return add_constraint(m, c);
}

// ---- bulk builders ---------------------------------------------------------------------
int pbdx_model_add_cloth_constraints(pbdx_model *m, uint32_t tmi, uint32_t method,
	float k, float xx, float yy, float xy, float xyP, float yxP, int ns, int nsh)
{
	if (m && m->inst_count > 1) { set_error("the model holds instances (pbdx_model_add_instances): no constraints can be added any more"); return PBDX_ERR_INVALID; }
	if (!m || tmi >= m->tri_models.size()) { set_error("add_cloth_constraints: bad triangle model"); return PBDX_ERR_INVALID; }
	const uint32_t offset = m->tri_models[tmi].index_offset;
	if (method == 1 || method == 4)
	{
		const size_t ne = m->tri_models[tmi].edges.size();
		grow_for(m->constraints, ne);
		for (size_t i = 0; i < ne; i++)
		{
			const TriMesh::Edge e = m->tri_models[tmi].edges[i];
			if (method == 1) pbdx_model_add_distance_constraint(m, e.vert[0] + offset, e.vert[1] + offset, k);
			else pbdx_model_add_distance_constraint_xpbd(m, e.vert[0] + offset, e.vert[1] + offset, k);
		}
	}
	else if (method == 2 || method == 3)
	{
		const size_t nf = m->tri_models[tmi].faces.size() / 3;
		for (size_t i = 0; i < nf; i++)
		{
			const uint32_t v1 = m->tri_models[tmi].faces[3 * i] + offset;
			const uint32_t v2 = m->tri_models[tmi].faces[3 * i + 1] + offset;
			const uint32_t v3 = m->tri_models[tmi].faces[3 * i + 2] + offset;
			if (method == 2) pbdx_model_add_fem_triangle_constraint(m, v1, v2, v3, xx, yy, xy, xyP, yxP);
			else pbdx_model_add_strain_triangle_constraint(m, v1, v2, v3, xx, yy, xy, ns, nsh);
		}
	}
	return PBDX_OK;
}

int pbdx_model_add_bending_constraints(pbdx_model *m, uint32_t tmi, uint32_t method, float stiffness)
{
	if (!m || tmi >= m->tri_models.size()) { set_error("add_bending_constraints: bad triangle model"); return PBDX_ERR_INVALID; }
	if (method < 1 || method > 3)
		return PBDX_OK;
	const uint32_t offset = m->tri_models[tmi].index_offset;
	const size_t ne = m->tri_models[tmi].edges.size();
	grow_for(m->constraints, ne);
	for (size_t i = 0; i < ne; i++)
	{
		const TriMesh::Edge e = m->tri_models[tmi].edges[i];
		const uint32_t *tris = m->tri_models[tmi].faces.data();
		if (e.face[0] == 0xffffffffu || e.face[1] == 0xffffffffu)
			continue;
		int point1 = -1, point2 = -1;
		for (int j = 0; j < 3; j++)
			if (tris[3 * e.face[0] + j] != e.vert[0] && tris[3 * e.face[0] + j] != e.vert[1]) { point1 = (int)tris[3 * e.face[0] + j]; break; }
		for (int j = 0; j < 3; j++)
			if (tris[3 * e.face[1] + j] != e.vert[0] && tris[3 * e.face[1] + j] != e.vert[1]) { point2 = (int)tris[3 * e.face[1] + j]; break; }
		if (point1 == -1 || point2 == -1)
			continue;
		const uint32_t v1 = point1 + offset, v2 = point2 + offset, v3 = e.vert[0] + offset, v4 = e.vert[1] + offset;
		if (method == 1) pbdx_model_add_dihedral_constraint(m, v1, v2, v3, v4, stiffness);
		else if (method == 2) pbdx_model_add_isometric_bending_constraint(m, v1, v2, v3, v4, stiffness);
		else pbdx_model_add_isometric_bending_constraint_xpbd(m, v1, v2, v3, v4, stiffness);
	}
	return PBDX_OK;
}

int pbdx_model_add_solid_constraints(pbdx_model *m, uint32_t tmi, uint32_t method,
	float stiffness, float poisson, float volume_stiffness, int ns, int nsh)
{
	if (!m || tmi >= m->tet_models.size()) { set_error("add_solid_constraints: bad tet model"); return PBDX_ERR_INVALID; }
	(void)nsh; // the reference passes normalizeStretch twice (SimulationModel.cpp:1308); reproduced below
	const uint32_t offset = m->tet_models[tmi].index_offset;
	const size_t nt = m->tet_models[tmi].tets.size() / 4;
	if (method == 1 || method == 6)
	{
		const size_t ne = m->tet_models[tmi].edges.size();
		for (size_t i = 0; i < ne; i++)
		{
			const TetMesh::Edge e = m->tet_models[tmi].edges[i];
			if (method == 1) pbdx_model_add_distance_constraint(m, e.vert[0] + offset, e.vert[1] + offset, stiffness);
			else pbdx_model_add_distance_constraint_xpbd(m, e.vert[0] + offset, e.vert[1] + offset, stiffness);
		}
	}
	for (size_t i = 0; i < nt; i++)
	{
		uint32_t v[4];
		for (int j = 0; j < 4; j++) v[j] = m->tet_models[tmi].tets[4 * i + j] + offset;
		switch (method)
		{
		case 1: pbdx_model_add_volume_constraint(m, v[0], v[1], v[2], v[3], volume_stiffness); break;
		case 2: pbdx_model_add_fem_tet_constraint(m, v[0], v[1], v[2], v[3], stiffness, poisson); break;
		case 3: pbdx_model_add_fem_tet_constraint_xpbd(m, v[0], v[1], v[2], v[3], stiffness, poisson); break;
		case 4: pbdx_model_add_strain_tet_constraint(m, v[0], v[1], v[2], v[3], stiffness, stiffness, ns, ns); break;
		case 5:
		{
			uint32_t nc[4];
			for (int j = 0; j < 4; j++) nc[j] = m->tet_models[tmi].vertex_tet_count[v[j] - offset];
			pbdx_model_add_shape_matching_constraint(m, 4, v, nc, stiffness);
			break;
		}
		case 6: pbdx_model_add_volume_constraint_xpbd(m, v[0], v[1], v[2], v[3], volume_stiffness); break;
		default: break;
		}
	}
	return PBDX_OK;
}

// ---- inspection ----------------------------------------------------------------------------
uint32_t pbdx_model_num_constraints(const pbdx_model *m) { return m ? (uint32_t)m->num_constraints() : 0; }
int pbdx_model_constraint_type(const pbdx_model *m, uint32_t c) { return (m && c < m->num_constraints()) ? m->constraints[c % m->constraints.size()].type : -1; }
int pbdx_model_constraint_bodies(const pbdx_model *m, uint32_t c, uint32_t *out)
{
	if (!m || !out || c >= m->num_constraints()) return PBDX_ERR_INVALID;
	const HostConstraint &p = m->constraints[c % m->constraints.size()];
	const uint32_t shift = (uint32_t)(c / m->constraints.size()) * m->inst_particles;
	for (uint32_t j = 0; j < type_info(p.type)->num_bodies; j++) out[j] = p.bodies[j] + shift;
	return PBDX_OK;
}
int pbdx_model_constraint_params(const pbdx_model *m, uint32_t c, float *out)
{
	if (!m || !out || c >= m->num_constraints()) return PBDX_ERR_INVALID;
	HostConstraint k;
	if (!model_constraint(m, c, k)) { set_error("constraint %u is degenerate in its instance", c); return PBDX_ERR_INVALID; }
	memcpy(out, k.params, type_info(k.type)->param_stride * sizeof(float));
	return PBDX_OK;
}
// In an instanced model the user parameters (stiffness ...) live in the prototype: an edit of prototype constraint i applies to
// constraint i of every instance; the rest data of the copies is always derived from their rest positions.
int pbdx_model_set_constraint_params(pbdx_model *m, uint32_t c, const float *in)
{
	if (!m || !in || c >= m->num_constraints()) return PBDX_ERR_INVALID;
	if (c >= m->constraints.size()) { set_error("set_constraint_params: in an instanced model edit the prototype's constraint (index %u)", (uint32_t)(c % m->constraints.size())); return PBDX_ERR_UNSUPPORTED; }
	memcpy(m->constraints[c].params, in, type_info(m->constraints[c].type)->param_stride * sizeof(float));
	m->params_version++;
	return PBDX_OK;
}

// ---- colouring -----------------------------------------------------------------------------
// the constraints' bodies as CSR arrays (instanced model: the prototype's)
static void model_bodies_csr(const pbdx_model *m, std::vector<uint32_t> &off, std::vector<uint32_t> &bodies)
{
	const size_t nc = m->constraints.size();
	off.assign(nc + 1, 0);
	bodies.clear();
	bodies.reserve(nc * 4);
	for (size_t ci = 0; ci < nc; ci++)
	{
		const HostConstraint &c = m->constraints[ci];
		const uint32_t nb = type_info(c.type)->num_bodies;
		for (uint32_t k = 0; k < nb; k++) bodies.push_back(c.bodies[k]);
		off[ci + 1] = (uint32_t)bodies.size();
	}
}
static void install_groups(pbdx_model *m, const std::vector<uint32_t> &group_of, uint32_t ngroups)
{
	m->groups.assign(ngroups, std::vector<uint32_t>());
	for (size_t ci = 0; ci < group_of.size(); ci++) m->groups[group_of[ci]].push_back((uint32_t)ci);
	m->groups_initialized = true;
}

int pbdx_colour_constraints_host(uint32_t num_bodies, uint32_t num_constraints, const uint32_t *body_off, const uint32_t *bodies,
	uint32_t *group_of, uint32_t *num_groups)
{
	if ((!body_off || !bodies || !group_of) && num_constraints) { set_error("pbdx_colour_constraints_host: null argument"); return PBDX_ERR_INVALID; }
	// used[p*words + w]: bit g set <=> body p is already touched by a constraint of group g  (the reference keeps one byte map per group and
	// tries the groups one after the other, SimulationModel.cpp:1046-1083: the same first fit, 5-7 times the time at 6 M constraints)
	uint32_t words = 1, ngroups = 0;
	std::vector<uint64_t> used((size_t)num_bodies * words, 0);
	for (uint32_t ci = 0; ci < num_constraints; ci++)
	{
		const uint32_t *b = bodies + body_off[ci];
		const uint32_t nb = body_off[ci + 1] - body_off[ci];
		for (uint32_t k = 0; k < nb; k++) if (b[k] >= num_bodies) { set_error("pbdx_colour_constraints_host: body index out of range"); return PBDX_ERR_INVALID; }
		uint32_t g = 0xffffffffu;
		for (uint32_t w = 0; w < words && g == 0xffffffffu; w++)
		{
			uint64_t occ = 0;
			for (uint32_t k = 0; k < nb; k++) occ |= used[(size_t)b[k] * words + w];
			if (~occ)
			{
				const uint32_t bit = (uint32_t)__builtin_ctzll(~occ);
				if (w * 64 + bit <= ngroups)   // first free existing group, or the next new one
					g = w * 64 + bit;
			}
		}
		if (g == 0xffffffffu || g >= ngroups)
		{
			g = ngroups++;
			if (g >= words * 64)
			{
				// grow bitset stride
				const uint32_t nw = words + 1;
				std::vector<uint64_t> grown((size_t)num_bodies * nw, 0);
				for (uint32_t p = 0; p < num_bodies; p++)
					for (uint32_t w = 0; w < words; w++) grown[(size_t)p * nw + w] = used[(size_t)p * words + w];
				used.swap(grown);
				words = nw;
			}
		}
		group_of[ci] = g;
		for (uint32_t k = 0; k < nb; k++) used[(size_t)b[k] * words + (g >> 6)] |= 1ull << (g & 63);
	}
	if (num_groups) *num_groups = ngroups;
	return PBDX_OK;
}

int pbdx_model_init_constraint_groups(pbdx_model *m)
{
	if (!m) return PBDX_ERR_INVALID;
	if (m->groups_initialized)
		return PBDX_OK;
	// (instanced model: the prototype is coloured; group g of the model = the prototype's group g, instance after instance)
	const uint32_t n = m->inst_count > 1 ? m->inst_particles : m->size();
	std::vector<uint32_t> off, bodies;
	model_bodies_csr(m, off, bodies);
	std::vector<uint32_t> group_of(m->constraints.size(), 0);
	uint32_t ngroups = 0;
	const int r = pbdx_colour_constraints_host(n, (uint32_t)m->constraints.size(), off.data(), bodies.data(), group_of.data(), &ngroups);
	if (r != PBDX_OK) return r;
	install_groups(m, group_of, ngroups);
	return PBDX_OK;
}

int pbdx_model_init_constraint_groups_device(pbdx_model *m, int device)
{
	if (!m) return PBDX_ERR_INVALID;
	if (m->groups_initialized)
		return PBDX_OK;
	// (instanced model: the prototype is coloured, as on the host)
	const uint32_t n = m->inst_count > 1 ? m->inst_particles : m->size();
	std::vector<uint32_t> off, bodies;
	model_bodies_csr(m, off, bodies);
	std::vector<uint32_t> group_of(m->constraints.size(), 0);
	uint32_t ngroups = 0;
	const int r = pbdx_colour_constraints(device, n, (uint32_t)m->constraints.size(), off.data(), bodies.data(), group_of.data(), &ngroups, nullptr);
	if (r != PBDX_OK) return r;
	install_groups(m, group_of, ngroups);
	return PBDX_OK;
}

int pbdx_model_groups_initialized(const pbdx_model *m) { return m && m->groups_initialized; }
uint32_t pbdx_model_num_groups(const pbdx_model *m) { return m ? (uint32_t)m->groups.size() : 0; }
uint32_t pbdx_model_group_size(const pbdx_model *m, uint32_t g) { return (m && g < m->groups.size()) ? (uint32_t)m->groups[g].size() * m->inst_count : 0; }
int pbdx_model_get_group(const pbdx_model *m, uint32_t g, uint32_t *out)
{
	if (!m || !out || g >= m->groups.size()) return PBDX_ERR_INVALID;
	const size_t n = m->groups[g].size();
	memcpy(out, m->groups[g].data(), n * sizeof(uint32_t));
	const uint32_t nc = (uint32_t)m->constraints.size();
	for (uint32_t k = 1; k < m->inst_count; k++)
		for (size_t i = 0; i < n; i++) out[(size_t)k * n + i] = m->groups[g][i] + k * nc;
	return PBDX_OK;
}

} // extern "C"

namespace pbdx {
bool model_constraint(const pbdx_model *m, uint64_t c, HostConstraint &out)
{
	const size_t nc = m->constraints.size();
	out = m->constraints[c % nc];
	if (c < nc) return true;
	const uint32_t shift = (uint32_t)(c / nc) * m->inst_particles;
	for (uint32_t j = 0; j < type_info(out.type)->num_bodies; j++) out.bodies[j] += shift;
	return init_geometry(m, out);
}
}
