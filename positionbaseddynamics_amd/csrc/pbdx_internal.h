// pbdx_internal.h -- shared internals of libpbdx (not part of the ABI).
#ifndef PBDX_INTERNAL_H
#define PBDX_INTERNAL_H

#include "../../include/pbdx.h"
#include "../../include/pbdx_debug.h"
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <new>

namespace pbdx {

void set_error(const char *fmt, ...);
void clear_error();

struct TypeInfo
{
	const char *name;
	uint32_t num_bodies;
	uint32_t param_stride;
	uint32_t algorithmic_bytes;   // SURVEY.md 8d, fp32, 4-byte indices
	bool xpbd;                    // owns a lambda stream
};
const TypeInfo *type_info(int type);
// structural checks of deformable colliders (pbdx_tetcontact.cpp): ranges, hierarchies are trees of depth <= 62, friction 0, indices < 2^24
int validate_tet_colliders(uint32_t n, const pbdx_tet_collider *colliders, uint32_t n_particles);

// ---- host model (pbdx_model.cpp) -------------------------------------------
struct HostConstraint
{
	int type;
	uint32_t bodies[4];
	float params[24];
};

// how a mesh model's rest positions were produced: needed to give an INSTANCE of the model the positions the reference's
// builder would have given it (addRegularTriangleModel / addRegularTetModel evaluate R * p + T in Real; an instance
// re-evaluates that with its own T)
struct MeshRecipe
{
	int regular = 0;             // 0: explicit points (an instance's points = the prototype's + offset), 1: regular grid
	int dims[3] = { 0, 0, 0 };
	float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
	float T[3] = { 0, 0, 0 };
	float scale[3] = { 1, 1, 1 };
};

struct TriMesh
{
	MeshRecipe recipe;
	uint32_t index_offset;
	uint32_t num_vertices;
	std::vector<uint32_t> faces;            // 3 per face
	struct Edge { uint32_t vert[2]; uint32_t face[2]; };
	std::vector<Edge> edges;
};

struct TetMesh
{
	MeshRecipe recipe;
	uint32_t index_offset;
	uint32_t num_vertices;
	std::vector<uint32_t> tets;             // 4 per tet
	struct Edge { uint32_t vert[2]; };
	std::vector<Edge> edges;
	std::vector<uint32_t> vertex_tet_count; // |verticesTets[v]| (shape-matching cluster counts)
};

// stable radix sort of 32-bit (key, value) pairs on a HIP stream (pbdx_colour.hip; temp == nullptr: *temp_bytes receives the scratch size)
int sort_pairs_u32(void *temp, size_t *temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, void *stream);

} // namespace pbdx

struct pbdx_model;
namespace pbdx { uint64_t next_model_uid(); const pbdx_model *find_model(uint64_t uid); /* null: destroyed */ }

// pbdx_hostio.hip (the copies to / from the device are declared in pbdx_device.h)
namespace pbdx {
void host_copy(void *dst, const void *src, size_t bytes);             // memcpy, by several threads when large
// page-locked memory the library owns (malloc without a HIP device), registered by address range: the only host memory whose ADDRESS reaches the GPU
void *pinned_alloc(size_t bytes);
void pinned_free(void *p);
bool is_library_pinned(const void *p, size_t bytes);
template <class T> struct PinnedAllocator
{
	typedef T value_type;
	PinnedAllocator() {}
	template <class U> PinnedAllocator(const PinnedAllocator<U> &) {}
	T *allocate(size_t n) { T *p = static_cast<T *>(pinned_alloc(n * sizeof(T))); if (!p && n) throw std::bad_alloc(); return p; }
	void deallocate(T *p, size_t) { pinned_free(p); }
	template <class U> bool operator==(const PinnedAllocator<U> &) const { return true; }
	template <class U> bool operator!=(const PinnedAllocator<U> &) const { return false; }
};
// a particle array of the host mirror (pbdx_model): std::vector<float> in page-locked memory of the library
typedef std::vector<float, PinnedAllocator<float>> ParticleArray;
}

struct pbdx_model
{
	// ParticleData (Simulation/ParticleData.h:91-100), packed xyz
	// (page-locked memory of the library where a HIP device exists: a step's transfers go straight from / into these arrays, pbdx_hostio.hip)
	pbdx::ParticleArray mass, inv_mass;
	pbdx::ParticleArray x0, x, v, a, old_x, last_x;
	std::vector<pbdx::TriMesh> tri_models;
	std::vector<pbdx::TetMesh> tet_models;
	std::vector<pbdx::HostConstraint> constraints;
	std::vector<std::vector<uint32_t>> groups;
	bool groups_initialized = false;
	const uint64_t uid = pbdx::next_model_uid();   // never reused (an address can be)
	uint64_t topology_version = 0;   // bumped by every add*/cleanup: device image invalidation
	uint64_t params_version = 0;     // bumped by set_constraint_params / set_mass
	uint64_t state_version = 0;      // bumped when the host particle state (x, v, a, oldX, lastX) is written through the API
	uint32_t dirty_arrays = 0;       // bit `which` (pbdx_model_get_array numbering) set: that host array was written since the device image last
	                                 // agreed with the host (a resident time step pulls the OTHER arrays from the device before it re-uploads)

	// Instanced models (pbdx_model_add_instances): the model holds inst_count congruent copies of a PROTOTYPE (= its first
	// inst_particles particles, the mesh models and the constraints it contained when the instances were added).  Particle
	// state is materialised for every instance; mesh topology, constraints and colour groups are stored once -- instance k's
	// constraint i is the prototype's with particle indices + k * inst_particles and rest data evaluated at ITS rest
	// positions (constraint index k * nc + i, as if the builders had been called instance after instance); colour group g
	// holds the prototype's members of g for instance 0, then for instance 1, ... (the reference's first-fit colouring of
	// K disjoint congruent instances appended in order is exactly that).
	uint32_t inst_count = 1;
	uint32_t inst_particles = 0;
	std::vector<float> inst_offset;  // 3 per instance (instance 0: zeros)

	uint32_t size() const { return (uint32_t)mass.size(); }
	uint64_t num_constraints() const { return (uint64_t)constraints.size() * inst_count; }
	uint32_t num_tri_models() const { return (uint32_t)tri_models.size() * inst_count; }
	uint32_t num_tet_models() const { return (uint32_t)tet_models.size() * inst_count; }
};

namespace pbdx {
// constraint `c` of the (possibly instanced) model by value; false if an instance's element is degenerate
bool model_constraint(const pbdx_model *m, uint64_t c, HostConstraint &out);
}

#endif
