// pbdx_internal.h -- shared internals of libpbdx (not part of the ABI).
#ifndef PBDX_INTERNAL_H
#define PBDX_INTERNAL_H

#include "../../include/pbdx.h"
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

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

// ---- host model (pbdx_model.cpp) -------------------------------------------
struct HostConstraint
{
	int type;
	uint32_t bodies[4];
	float params[24];
};

struct TriMesh
{
	uint32_t index_offset;
	uint32_t num_vertices;
	std::vector<uint32_t> faces;            // 3 per face
	struct Edge { uint32_t vert[2]; uint32_t face[2]; };
	std::vector<Edge> edges;
};

struct TetMesh
{
	uint32_t index_offset;
	uint32_t num_vertices;
	std::vector<uint32_t> tets;             // 4 per tet
	struct Edge { uint32_t vert[2]; };
	std::vector<Edge> edges;
	std::vector<uint32_t> vertex_tet_count; // |verticesTets[v]| (shape-matching cluster counts)
};

} // namespace pbdx

struct pbdx_model
{
	// ParticleData (Simulation/ParticleData.h:91-100), packed xyz
	std::vector<float> mass, inv_mass;
	std::vector<float> x0, x, v, a, old_x, last_x;
	std::vector<pbdx::TriMesh> tri_models;
	std::vector<pbdx::TetMesh> tet_models;
	std::vector<pbdx::HostConstraint> constraints;
	std::vector<std::vector<uint32_t>> groups;
	bool groups_initialized = false;
	uint64_t topology_version = 0;   // bumped by every add*/cleanup: device image invalidation
	uint64_t params_version = 0;     // bumped by set_constraint_params / set_mass
	uint64_t state_version = 0;      // bumped when the host particle state (x, v, a, oldX, lastX) is written through the API
	uint32_t dirty_arrays = 0;       // bit `which` (pbdx_model_get_array numbering) set: that host array was written since the device image last
	                                 // agreed with the host (a resident time step pulls the OTHER arrays from the device before it re-uploads)

	uint32_t size() const { return (uint32_t)mass.size(); }
};

#endif
