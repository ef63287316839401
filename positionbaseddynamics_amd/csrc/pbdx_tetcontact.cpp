// pbdx_tetcontact.cpp -- host-side evaluation of the particle-tet contact detection (pbdx_debug_tet_contacts): the SAME code the
// device runs (pbdx_tetcontact.h is host + device), so that the detection can be pinned against the reference without a GPU.
#include "pbdx_internal.h"
#include "pbdx_tetcontact.h"
#include <string.h>
#include <vector>

using namespace pbdx;

namespace pbdx {

// host mirror of one collider: owns the (recomputed) hull arrays
struct HostTetCollider
{
	TetColliderView view;
	std::vector<P4> h_points, h_tets, h_tets0;
};

bool make_host_view(const pbdx_tet_collider &c, float tolerance, HostTetCollider &out)
{
	if (!c.tets || (c.points.num_nodes && (!c.points.entities || !c.points.nodes)) || (c.tets_bvh.num_nodes && (!c.tets_bvh.entities || !c.tets_bvh.nodes)) ||
		(c.tets_rest.num_nodes && (!c.tets_rest.entities || !c.tets_rest.nodes || !c.tets_rest.hulls)))
		return false;
	TetColliderView &v = out.view;
	memset(&v, 0, sizeof(v));
	v.sdf.shape = c.shape; v.sdf.invert = c.invert;
	memcpy(v.sdf.params, c.params, sizeof(c.params));
	v.first = c.first_particle; v.num_vertices = c.num_vertices; v.num_tets = c.num_tets; v.tets = c.tets;
	memcpy(v.X0, c.initial_x, sizeof(v.X0)); memcpy(v.R0, c.initial_R, sizeof(v.R0));
	v.tolerance = tolerance; v.test_mesh = c.test_mesh; v.body_index = c.body_index;
	out.h_points.assign(c.points.num_nodes, P4{ 0, 0, 0, 0 });
	out.h_tets.assign(c.tets_bvh.num_nodes, P4{ 0, 0, 0, 0 });
	out.h_tets0.resize(c.tets_rest.num_nodes);
	for (uint32_t i = 0; i < c.tets_rest.num_nodes; i++) out.h_tets0[i] = P4{ c.tets_rest.hulls[4 * i], c.tets_rest.hulls[4 * i + 1], c.tets_rest.hulls[4 * i + 2], c.tets_rest.hulls[4 * i + 3] };
	v.points = BvhView{ c.points.entities, c.points.nodes, out.h_points.data(), c.points.num_nodes };
	v.tet_bvh = BvhView{ c.tets_bvh.entities, c.tets_bvh.nodes, out.h_tets.data(), c.tets_bvh.num_nodes };
	v.tet_bvh0 = BvhView{ c.tets_rest.entities, c.tets_rest.nodes, out.h_tets0.data(), c.tets_rest.num_nodes };
	return true;
}

void contact_to_floats(const TetContact &c, float *o)
{
	o[0] = (float)c.particle; o[1] = (float)c.solid; o[2] = (float)c.tet;
	for (int k = 0; k < 3; k++) { o[3 + k] = c.bary[k]; o[6 + k] = c.normal[k]; }
	o[9] = c.nKn_inv;
	for (int v = 0; v < 4; v++) for (int k = 0; k < 3; k++) o[10 + 3 * v + k] = c.x[v][k];
	for (int v = 0; v < 4; v++) { o[22 + v] = c.w[v]; o[26 + v] = (float)c.vert[v]; }
	for (int k = 0; k < 3; k++) o[30 + k] = c.tangent[k];
	o[33] = c.p_max;
}


// Everything the device (and the host evaluation of the same header code) trusts about a set of deformable colliders; shared by
// pbdx_solver_set_tet_colliders and pbdx_debug_tet_contacts.
int validate_tet_colliders(uint32_t n, const pbdx_tet_collider *colliders, uint32_t n_particles)
{
	for (uint32_t i = 0; i < n; i++)
	{
		const pbdx_tet_collider &c = colliders[i];
		if (c.shape < PBDX_SHAPE_BOX || c.shape > PBDX_SHAPE_HOLLOW_BOX) { set_error("set_tet_colliders: unknown shape %d", c.shape); return PBDX_ERR_UNSUPPORTED; }
		if (c.friction != 0.0f)
		{
			set_error("set_tet_colliders: friction of a deformable-deformable contact must be 0 -- the reference's friction impulse for these contacts reads an uninitialised multiplier (Constraints.h:553)");
			return PBDX_ERR_UNSUPPORTED;
		}
		if ((uint64_t)c.first_particle + c.num_vertices > n_particles || !c.num_vertices) { set_error("set_tet_colliders: collider %u exceeds the %u uploaded particles", i, n_particles); return PBDX_ERR_INVALID; }
		if (!c.tets || !c.num_tets) { set_error("set_tet_colliders: collider %u has no tets", i); return PBDX_ERR_INVALID; }
		for (const pbdx_bvh *b : { &c.points, &c.tets_bvh, &c.tets_rest })
			if (!b->num_nodes || !b->entities || !b->nodes) { set_error("set_tet_colliders: collider %u lacks a bounding-sphere hierarchy", i); return PBDX_ERR_INVALID; }
		if (!c.tets_rest.hulls) { set_error("set_tet_colliders: the rest-pose hierarchy of collider %u needs its spheres", i); return PBDX_ERR_INVALID; }
		// structure checks: children and entity ranges in range (the device code trusts them)
		const pbdx_bvh *bs[3] = { &c.points, &c.tets_bvh, &c.tets_rest };
		const uint32_t ents[3] = { c.num_vertices, c.num_tets, c.num_tets };
		for (int q = 0; q < 3; q++)
		{
			if (bs[q]->num_entities != ents[q]) { set_error("set_tet_colliders: hierarchy %d of collider %u has %u entities, expected %u", q, i, bs[q]->num_entities, ents[q]); return PBDX_ERR_INVALID; }
			for (uint32_t e = 0; e < ents[q]; e++) if (bs[q]->entities[e] >= ents[q]) { set_error("set_tet_colliders: entity out of range"); return PBDX_ERR_INVALID; }
			for (uint32_t nd = 0; nd < bs[q]->num_nodes; nd++)
			{
				const int32_t *k = bs[q]->nodes + 4 * nd;
				const bool leaf = k[0] < 0 && k[1] < 0;
				if ((!leaf && (k[0] < 0 || k[1] < 0 || (uint32_t)k[0] >= bs[q]->num_nodes || (uint32_t)k[1] >= bs[q]->num_nodes)) || k[2] < 0 || k[3] <= 0 ||
					(uint64_t)k[2] + (uint64_t)k[3] > ents[q])
				{ set_error("set_tet_colliders: node %u of hierarchy %d of collider %u is malformed", nd, q, i); return PBDX_ERR_INVALID; }
			}
		}
		// ... and each hierarchy is a TREE rooted at node 0 (every node reached at most once: no cycles, no shared subtrees) of depth <= 62:
		// the device walks it with fixed stacks (find_ref_tet_at: 64 entries = depth + 2; the pair traversal: generations) and never checks again
		for (int q = 0; q < 3; q++)
		{
			std::vector<uint8_t> seen(bs[q]->num_nodes, 0);
			std::vector<std::pair<uint32_t, uint32_t> > stack(1, std::make_pair(0u, 0u));
			while (!stack.empty())
			{
				const uint32_t nd = stack.back().first, depth = stack.back().second;
				stack.pop_back();
				if (seen[nd]) { set_error("set_tet_colliders: hierarchy %d of collider %u is not a tree (node %u is reached twice)", q, i, nd); return PBDX_ERR_INVALID; }
				seen[nd] = 1;
				if (depth > 62) { set_error("set_tet_colliders: hierarchy %d of collider %u is deeper than 62 levels", q, i); return PBDX_ERR_INVALID; }
				const int32_t *k = bs[q]->nodes + 4 * nd;
				if (k[0] >= 0 || k[1] >= 0) { stack.push_back(std::make_pair((uint32_t)k[0], depth + 1)); stack.push_back(std::make_pair((uint32_t)k[1], depth + 1)); }
			}
		}
		for (uint32_t t = 0; t < 4 * c.num_tets; t++) if (c.tets[t] >= c.num_vertices) { set_error("set_tet_colliders: tet vertex out of range"); return PBDX_ERR_INVALID; }
		// the 30-float contact records (pbdx_solver_get_tet_contacts) carry particle / tet indices as floats: exact up to 2^24
		if ((uint64_t)c.first_particle + c.num_vertices > (1u << 24) || c.num_tets > (1u << 24))
		{ set_error("set_tet_colliders: collider %u reaches particle / tet indices above 2^24 (the contact records store indices as floats)", i); return PBDX_ERR_UNSUPPORTED; }
	}
	return PBDX_OK;
}

} // namespace pbdx

extern "C" int pbdx_debug_tet_contacts(uint32_t n_particles, const float *pos4, const float *rest4, const float *vel4, uint32_t n, const pbdx_tet_collider *colliders,
	float tolerance, uint32_t capacity, uint32_t *count, float *out)
{
	if (!pos4 || !rest4 || (n && !colliders) || !count) { set_error("debug_tet_contacts: null argument"); return PBDX_ERR_INVALID; }
	{ const int rv = validate_tet_colliders(n, colliders, n_particles); if (rv) return rv; }
	const P4 *pos = reinterpret_cast<const P4 *>(pos4), *x0 = reinterpret_cast<const P4 *>(rest4);
	std::vector<HostTetCollider> cs(n);
	std::vector<float> aabb((size_t)6 * n);
	for (uint32_t i = 0; i < n; i++)
	{
		if (!make_host_view(colliders[i], tolerance, cs[i])) { set_error("debug_tet_contacts: collider %u has a null array", i); return PBDX_ERR_INVALID; }
		const TetColliderView &v = cs[i].view;
		if ((uint64_t)v.first + v.num_vertices > n_particles) { set_error("debug_tet_contacts: collider %u exceeds the particles", i); return PBDX_ERR_INVALID; }
		// KDTree::update of the point and tet hierarchies, CollisionDetection::updateAABB
		for (uint32_t nd = 0; nd < v.points.num_nodes; nd++) hull_points(v.points, nd, pos + v.first);
		for (uint32_t nd = 0; nd < v.tet_bvh.num_nodes; nd++) hull_tets(v.tet_bvh, nd, pos + v.first, v.tets, tolerance);
		float *bb = &aabb[6 * i];
		for (int k = 0; k < 3; k++) bb[k] = bb[3 + k] = (&pos[v.first].x)[k];
		for (uint32_t p = v.first + 1; p < v.first + v.num_vertices; p++)
			for (int k = 0; k < 3; k++)
			{
				const float q = (&pos[p].x)[k];
				if (bb[k] > q) bb[k] = q;
				if (bb[3 + k] < q) bb[3 + k] = q;
			}
	}
	uint32_t found = 0;
	bool ok = true;
	for (uint32_t i = 0; i < n; i++)
		for (uint32_t k = 0; k < n; k++)
		{
			if (i == k || !cs[i].view.test_mesh || !aabb_intersect(&aabb[6 * i], &aabb[6 * k])) continue;
			ok = tet_pair_contacts(cs[i].view, cs[k].view, pos, x0, reinterpret_cast<const P4 *>(vel4), [&](const TetContact &c) {
				if (found < capacity && out) contact_to_floats(c, out + (size_t)found * PBDX_TET_CONTACT_FLOATS);
				found++;
			}) && ok;
		}
	*count = found;
	if (!ok) { set_error("debug_tet_contacts: traversal stack overflow"); return PBDX_ERR_INVALID; }
	return PBDX_OK;
}

// Known-answer entry of the velocity part of a particle-tet contact (init_ParticleTetContactConstraint's tangent / pMax and
// velocitySolve_ParticleTetContactConstraint with friction 0): in: invMass0, v0[3], invMass[4], v[4][3], bary[3], normal[3] (26 floats);
// out: tangent[3], pMax, applied (1 / 0), corr_v0[3], corr_v[4][3] (20 floats; corrections of static particles and of contacts without an impulse are 0)
extern "C" int pbdx_debug_tet_velocity_kat(const float *in, float *out)
{
	if (!in || !out) return PBDX_ERR_INVALID;
	const float w0 = in[0];
	const V3 v0 = mk(in[1], in[2], in[3]);
	TetContact c;
	memset(&c, 0, sizeof(c));
	V3 v[4];
	for (int k = 0; k < 4; k++) { c.w[k] = in[4 + k]; v[k] = mk(in[8 + 3 * k], in[9 + 3 * k], in[10 + 3 * k]); }
	const V3 bary = mk(in[20], in[21], in[22]), normal = mk(in[23], in[24], in[25]);
	c.bary[0] = bary.x; c.bary[1] = bary.y; c.bary[2] = bary.z;
	const float bary0 = 1.0f - bary.x - bary.y - bary.z;
	const float JMinvJT = w0 + bary0 * bary0 * c.w[0] + bary.x * bary.x * c.w[1] + bary.y * bary.y * c.w[2] + bary.z * bary.z * c.w[3];
	V3 t; float p_max;
	tet_contact_velocity_info(v0, v, bary, normal, JMinvJT, t, p_max);
	c.tangent[0] = t.x; c.tangent[1] = t.y; c.tangent[2] = t.z; c.p_max = p_max;
	for (int k = 0; k < 20; k++) out[k] = 0.0f;
	out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = p_max;
	V3 pv;
	if (tet_contact_velocity_impulse(c, w0, pv))
	{
		out[4] = 1.0f;
		for (int r = 0; r < 5; r++)
		{
			V3 corr;
			if (tet_contact_velocity_share(c, w0, pv, r, corr)) { out[5 + 3 * r] = corr.x; out[6 + 3 * r] = corr.y; out[7 + 3 * r] = corr.z; }
		}
	}
	return PBDX_OK;
}

// solve_ParticleTetContactConstraint applied to pos4 (n x (x, y, z, invMass), updated in place) for the contacts records[order[0]],
// records[order[1]], ... (30-float records as pbdx_solver_get_tet_contacts returns them; order == NULL: 0, 1, 2, ...): the host side of the
// claim that the contact list may be solved level by level (tests/test_tetcontact.py)
extern "C" int pbdx_debug_tet_solve_host(uint32_t n_particles, float *pos4, uint32_t n_contacts, const float *records, const uint32_t *order)
{
	if ((n_particles && !pos4) || (n_contacts && !records)) return PBDX_ERR_INVALID;
	struct Access
	{
		pbdx::P4 *p;
		pbdx::P4 get(uint32_t i) const { return p[i]; }
		void add(uint32_t i, pbdx::V3 c) { p[i].x += c.x; p[i].y += c.y; p[i].z += c.z; }
	} acc = { reinterpret_cast<pbdx::P4 *>(pos4) };
	for (uint32_t q = 0; q < n_contacts; q++)
	{
		const float *o = records + (size_t)(order ? order[q] : q) * PBDX_TET_CONTACT_FLOATS;
		pbdx::TetContact c;
		c.particle = (uint32_t)o[0]; c.solid = (uint32_t)o[1]; c.tet = (uint32_t)o[2];
		for (int k = 0; k < 3; k++) { c.bary[k] = o[3 + k]; c.normal[k] = o[6 + k]; }
		c.nKn_inv = o[9];
		for (int v = 0; v < 4; v++) for (int k = 0; k < 3; k++) c.x[v][k] = o[10 + 3 * v + k];
		for (int v = 0; v < 4; v++) { c.w[v] = o[22 + v]; c.vert[v] = (uint32_t)o[26 + v]; }
		if (c.particle >= n_particles || c.vert[0] >= n_particles || c.vert[1] >= n_particles || c.vert[2] >= n_particles || c.vert[3] >= n_particles) return PBDX_ERR_INVALID;
		pbdx::tet_contact_position_solve(c, acc);
	}
	return PBDX_OK;
}

// Known-answer form of the contact of a particle with a rigid body of ANY mass on the HOST (pbdx_contact.h: compute_matrix_k, dyn_contact_init,
// dyn_contact_velocity_solve = init_ParticleRigidBodyContactConstraint + velocitySolve_ParticleRigidBodyContactConstraint +
// ParticleRigidBodyContactConstraint::solveVelocityConstraint, PositionBasedRigidBodyDynamics.cpp:11-45,2385-2539, Constraints.cpp:2148-2189).
// in (38 floats): invMass0, mass0, v0[3], invMass1, x1[3], v1[3], inertiaInverseW1[9] row-major, omega1[3], cp0[3], cp1[3], normal[3], restitution, stiffness,
// friction, sweeps.  out (20 floats): tangent[3], 1 / (n^T K n), pMax, goal velocity, then after `sweeps` velocity solves: v0[3], v1[3], omega1[3], sum of impulses, 0
extern "C" int pbdx_debug_dyn_contact_kat(const float *in, float *out)
{
	if (!in || !out) return PBDX_ERR_INVALID;
	const float w0 = in[0], m0 = in[1], w1 = in[5];
	V3 v0 = mk(in[2], in[3], in[4]);
	const V3 x1 = mk(in[6], in[7], in[8]);
	V3 v1 = mk(in[9], in[10], in[11]);
	const float *Ji = in + 12;
	V3 om = mk(in[21], in[22], in[23]);
	const V3 cp0 = mk(in[24], in[25], in[26]), cp1 = mk(in[27], in[28], in[29]), n = mk(in[30], in[31], in[32]);
	DynContactInfo ci;
	dyn_contact_init(w0, v0, w1, x1, v1, Ji, om, cp0, cp1, n, in[33], ci);
	out[0] = ci.tangent.x; out[1] = ci.tangent.y; out[2] = ci.tangent.z; out[3] = ci.nKn_inv; out[4] = ci.pMax; out[5] = ci.goal;
	float sum = 0.0f;
	for (int k = 0; k < (int)in[36]; k++) dyn_contact_velocity_solve(w0, m0, v0, w1, x1, v1, Ji, om, in[34], in[35], sum, ci);
	out[6] = v0.x; out[7] = v0.y; out[8] = v0.z; out[9] = v1.x; out[10] = v1.y; out[11] = v1.z; out[12] = om.x; out[13] = om.y; out[14] = om.z; out[15] = sum;
	out[16] = out[17] = out[18] = out[19] = 0.0f;
	return PBDX_OK;
}
