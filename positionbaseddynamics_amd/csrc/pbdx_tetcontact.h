// pbdx_tetcontact.h -- particle vs deformable solid contacts (ParticleTetContactConstraint), host + device code.
// SURVEY 8f rank 2, second half: tet models that carry an analytic distance field in their rest frame colliding with the particles
// of other tet models (DistanceFieldCollisionDetection with tet-model collision objects + initTetBVH, as
// Demos/SceneLoaderDemo/SceneLoaderDemo.cpp:740-760 sets up with Discregrid SDFs; here with the analytic shapes of pbdx_contact.h).
//
// Restates, in the reference's operation order (Real = float):
//   KDTree::update -> PointCloudBSH / TetMeshBSH::compute_hull_approx    Simulation/kdTree.inl:186-197, BoundingSphereHierarchy.cpp:34-51,72-98
//   BoundingSphere::overlaps / contains                                  Simulation/BoundingSphere.h:201-228
//   BVHTest::traverse (dual depth-first traversal; ITS VISITING ORDER IS THE CONTACT ORDER)   BoundingSphereHierarchy.cpp:124-174
//   DistanceFieldCollisionDetection::collisionDetectionSolidSolid        DistanceFieldCollisionDetection.cpp:361-483
//   DistanceFieldCollisionDetection::findRefTetAt                        :744-812   (KDTree::traverse_depth_first kdTree.inl:73-107)
//   ParticleTetContactConstraint::initConstraint / solvePositionConstraint   Constraints.cpp:2191-2277
//   PositionBasedDynamics::init_ / solve_ParticleTetContactConstraint    PositionBasedDynamics.cpp:1172-1265
// NOT restated: the kd-tree CONSTRUCTION (kdTree.inl:6-71 sorts tied coordinates with std::sort, whose order among equal keys is
// the C++ library's business).  The hierarchies are built once, by the reference itself, when a collision object is registered;
// a binding hands the engine their structure (entity order + nodes) and the engine only refreshes the bounding spheres every step
// (which is all the reference does after construction).
// The velocity solve of these contacts (velocitySolve_ParticleTetContactConstraint, PositionBasedDynamics.cpp:1274-1327) reads m_lambda
// before anything has written it (Constraints.h:553, SimulationModel.cpp:557: the contact list is rebuilt right before the velocity solve
// and the position solve that sets m_lambda runs a step later).  With a friction coefficient of zero `frictionCoeff * lambda` is zero for
// any finite garbage, and that is the case this engine implements (a non-zero friction coefficient is refused):
//     pMax = 1 / (J M^-1 J^T) * u_rel . t                    (init_ParticleTetContactConstraint :1199-1213)
//     0 > pMax   ->  pv = -pMax * t,  v0 += w0 pv,  v_k -= w_k bary_k pv        (:1305-1306)
//     otherwise  ->  pv = (-0 * garbage) * t = a signed zero: nothing moves
// pMax is NOT always >= 0: when the tangential part t = u_rel - (u_rel . n) n is short (|t|^2 <= 1e-6) it is left un-normalised and
// u_rel . t = |t|^2 + (u_rel . n)(n . t) is negative whenever the rounding residue n . t outweighs |t|^2 -- for near-normal impacts about
// every second contact.  The impulse is then tiny (it only shows in velocity components that are themselves ~0) but it is what the
// reference computes, and it is applied maxIterationsV times, interleaved with the rigid-body contact sweeps (TimeStepController.cpp:
// 342-355).  The impulse of a contact is a constant of the contact, so the interleaving decomposes into per-particle chains again.
#ifndef PBDX_TETCONTACT_H
#define PBDX_TETCONTACT_H

#include "pbdx_contact.h"
#include <stdint.h>

namespace pbdx {

struct P4 { float x, y, z, w; };       // layout of the engine's float4 position records (x, y, z, invMass)

struct BvhView
{
	const uint32_t *lst;          // entity order of the kd-tree
	const int32_t *nodes;         // 4 per node: child0, child1 (-1 = none), begin, n
	P4 *hulls;                    // per node: centre, radius
	uint32_t num_nodes;
	// device form only (pbdx_tetcontact_dev.h): the entities' vertices in list order -- 1 per point, 4 per tet -- as model-local particle
	// indices (static) and as positions (gathered every step), so that a node's sums run over a contiguous range
	const uint32_t *flat;
	P4 *gathered;
	float *soa;                   // the same positions as three component arrays of num_elements floats each (x, y, z)
	uint32_t per_entity, num_elements;
};

struct TetColliderView
{
	pbdx_collider sdf;            // shape / invert / params (the frame fields are not used: the rest frame is X0 / R0 below)
	uint32_t first;               // first particle of the tet model
	uint32_t num_vertices, num_tets;
	const uint32_t *tets;         // 4 model-local vertex indices per tet
	float X0[3], R0[9];           // TetModel::getInitialX / getInitialR (row-major)
	float tolerance;              // CollisionDetection::m_tolerance (tet hierarchy spheres are inflated by it)
	int test_mesh;
	uint32_t body_index;          // tet model index
	BvhView points, tet_bvh, tet_bvh0;
};

struct TetContact                 // one ParticleTetContactConstraint: what its position solve reads
{
	uint32_t particle;            // global particle index (m_bodies[0])
	uint32_t vert[4];             // global particle indices of the tet's vertices
	uint32_t solid, tet;          // m_solidIndex, m_tetIndex (the tet that contains the closest surface point)
	float bary[3];                // m_bary
	float normal[3];              // m_constraintInfo.col(0)
	float nKn_inv;                // m_constraintInfo(0, 2)
	float x[4][3];                // m_x: the tet's vertex positions WHEN THE CONTACT WAS DETECTED (the solve uses these, not the current ones)
	float w[4];                   // m_invMasses
	float tangent[3];             // m_constraintInfo.col(1)
	float p_max;                  // m_constraintInfo(1, 2): "maximal impulse in tangent direction"
};

// init_ParticleTetContactConstraint's velocity part (PositionBasedDynamics.cpp:1190-1213): tangent and maximal tangent impulse from the
// velocities at detection.  v0: the particle's velocity, v[4]: the tet vertices'.
PBDX_HD void tet_contact_velocity_info(V3 v0, const V3 v[4], V3 bary, V3 normal, float JMinvJT, V3 &t, float &p_max)
{
	const float bary0 = 1.0f - bary.x - bary.y - bary.z;
	const V3 v1 = ((bary0 * v[0] + bary.x * v[1]) + bary.y * v[2]) + bary.z * v[3];
	const V3 u_rel = v0 - v1;
	const float u_rel_n = dot(normal, u_rel);
	t = u_rel - u_rel_n * normal;
	const float tl2 = sqn(t);
	if ((double)tl2 > 1.0e-6)
		t = t * (float)(1.0 / sqrt((double)tl2));       // `static_cast<Real>(1.0) / sqrt(tl2)`: in this translation unit sqrt is the C library's sqrt(double)
	p_max = 1.0f / JMinvJT * dot(u_rel, t);
}
// velocitySolve_ParticleTetContactConstraint for frictionCoeff == 0 (:1296-1324): true if the contact carries a non-zero impulse pv
// (then v0 += w0 pv if the particle is dynamic, v_k += (-w_k bary_k) pv for the dynamic tet vertices, k-th barycentric weight
// bary0, bary[0], bary[1], bary[2])
// (force: developer aid PBDX_OPT_TET_FORCE_IMPULSES -- the test of the application path: contacts with pMax > 0 are treated as the ones with pMax < 0
// are, because real scenes almost never produce a negative pMax; the arithmetic of the branch itself is pinned by the known-answer test)
PBDX_HD bool tet_contact_velocity_impulse(const TetContact &c, float w0, V3 &pv, bool force = false)
{
	if ((w0 == 0.0f) && (c.w[0] == 0.0f) && (c.w[1] == 0.0f) && (c.w[2] == 0.0f))
		return false;
	if (force ? !(c.p_max > 0.0f) : !(0.0f > c.p_max))
		return false;
	pv = (-c.p_max) * mk(c.tangent[0], c.tangent[1], c.tangent[2]);
	return true;
}
// the impulse's share of particle role r (0: the contact's particle, 1..4: the tet's vertices); false if that particle is static
PBDX_HD bool tet_contact_velocity_share(const TetContact &c, float w0, V3 pv, int r, V3 &corr)
{
	if (r == 0) { if (w0 == 0.0f) return false; corr = w0 * pv; return true; }
	const float bary0 = 1.0f - c.bary[0] - c.bary[1] - c.bary[2];
	const float b = r == 1 ? bary0 : c.bary[r - 2];
	if (c.w[r - 1] == 0.0f) return false;
	corr = (-c.w[r - 1] * b) * pv;
	return true;
}

PBDX_HD V3 p3(const P4 &p) { return mk(p.x, p.y, p.z); }

// ---- bounding spheres ------------------------------------------------------------------------------------------------------
// PointCloudBSH::compute_hull_approx: centre = (sum of the points in list order) / n, radius = sqrt(max squared distance)
PBDX_HD void hull_points(const BvhView &b, uint32_t node, const P4 *pos /* of the model's first particle */)
{
	const uint32_t beg = (uint32_t)b.nodes[4 * node + 2], n = (uint32_t)b.nodes[4 * node + 3];
	V3 x = mk(0.0f, 0.0f, 0.0f);
	for (uint32_t i = beg; i < beg + n; i++) x = x + p3(pos[b.lst[i]]);
	x = x / (float)n;
	float radius2 = 0.0f;
	for (uint32_t i = beg; i < beg + n; i++)
	{
		const float d = sqn(x - p3(pos[b.lst[i]]));
		radius2 = (radius2 < d) ? d : radius2;          // std::max
	}
	P4 h; h.x = x.x; h.y = x.y; h.z = x.z; h.w = sqrtf(radius2);
	b.hulls[node] = h;
}
// TetMeshBSH::compute_hull_approx.  Its radius is `sqrt(radius2) + m_tolerance` with the C library's sqrt(double): the sum is formed in
// double and rounded to Real once (BoundingSphereHierarchy.cpp:105) -- one ulp away from the float sum for about a third of the nodes.
PBDX_HD void hull_tets(const BvhView &b, uint32_t node, const P4 *pos, const uint32_t *tets, float tolerance)
{
	const uint32_t beg = (uint32_t)b.nodes[4 * node + 2], n = (uint32_t)b.nodes[4 * node + 3];
	V3 x = mk(0.0f, 0.0f, 0.0f);
	for (uint32_t i = beg; i < beg + n; i++)
	{
		const uint32_t t = b.lst[i];
		x = x + p3(pos[tets[4 * t]]); x = x + p3(pos[tets[4 * t + 1]]); x = x + p3(pos[tets[4 * t + 2]]); x = x + p3(pos[tets[4 * t + 3]]);
	}
	x = x / (4.0f * (float)n);
	float radius2 = 0.0f;
	for (uint32_t i = beg; i < beg + n; i++)
	{
		const uint32_t t = b.lst[i];
		for (int k = 0; k < 4; k++)
		{
			const float d = sqn(x - p3(pos[tets[4 * t + k]]));
			radius2 = (radius2 < d) ? d : radius2;
		}
	}
	P4 h; h.x = x.x; h.y = x.y; h.z = x.z; h.w = (float)(sqrt((double)radius2) + (double)tolerance);
	b.hulls[node] = h;
}
// BoundingSphere::overlaps: double rr = m_r + other.m_r (Real sum, promoted); squaredNorm (Real) < rr * rr (double)
PBDX_HD bool spheres_overlap(const P4 &a, const P4 &b)
{
	const double rr = (double)(a.w + b.w);
	return (double)sqn(p3(a) - p3(b)) < rr * rr;
}
PBDX_HD bool sphere_contains(const P4 &s, V3 p) { return sqn(p3(s) - p) < s.w * s.w; }

PBDX_HD M3 cols3(V3 c0, V3 c1, V3 c2)
{
	M3 A;
	A.m[0][0] = c0.x; A.m[1][0] = c0.y; A.m[2][0] = c0.z;
	A.m[0][1] = c1.x; A.m[1][1] = c1.y; A.m[2][1] = c1.z;
	A.m[0][2] = c2.x; A.m[1][2] = c2.y; A.m[2][2] = c2.z;
	return A;
}

// ---- findRefTetAt: the tet of the REST configuration that contains X best ------------------------------------------------------
// KDTree::traverse_depth_first calls the callback on every node it reaches and descends only where the predicate (sphere contains X)
// holds -- so the tets of a leaf are examined when the PARENT's sphere contains X, whether or not the leaf's own does.
PBDX_HD bool find_ref_tet_at(const TetColliderView &c, const P4 *x0 /* rest positions of the model's first particle */, V3 X, uint32_t &tet_index, V3 &bary_out)
{
	const BvhView &b = c.tet_bvh0;
	if (!b.num_nodes || !sphere_contains(b.hulls[0], X)) return false;
	uint32_t stack[64];
	int sp = 0;
	stack[sp++] = 0;
	bool any = false;
	float min_error = 3.402823466e+38f;      // REAL_MAX
	while (sp > 0)
	{
		const uint32_t node = stack[--sp];
		const int32_t c0 = b.nodes[4 * node], c1 = b.nodes[4 * node + 1];
		const bool leaf = c0 < 0 && c1 < 0;
		if (leaf)
		{
			const uint32_t beg = (uint32_t)b.nodes[4 * node + 2], n = (uint32_t)b.nodes[4 * node + 3];
			for (uint32_t i = beg; i < beg + n; i++)
			{
				const uint32_t t = b.lst[i];
				const V3 X0 = p3(x0[c.tets[4 * t]]), X1 = p3(x0[c.tets[4 * t + 1]]), X2 = p3(x0[c.tets[4 * t + 2]]), X3 = p3(x0[c.tets[4 * t + 3]]);
				const V3 bary = mul(inverse(cols3(X1 - X0, X2 - X0, X3 - X0)), X - X0);
				// "find best set of barycentric coordinates": first minimum of the error, in visiting order
				float error = (0.0f < -bary.x) ? -bary.x : 0.0f;
				error += (0.0f < -bary.y) ? -bary.y : 0.0f;
				error += (0.0f < -bary.z) ? -bary.z : 0.0f;
				const float over = bary.x + bary.y + bary.z - 1.0f;
				error += (0.0f < over) ? over : 0.0f;
				if (error < min_error) { min_error = error; tet_index = t; bary_out = bary; }
				any = true;
			}
		}
		else if (sphere_contains(b.hulls[node], X))     // (sp + 2 <= 64 always: pbdx_solver_set_tet_colliders admits only trees of depth <= 62, and a depth-first walk holds at most depth + 2 entries)
		{
			stack[sp++] = (uint32_t)c1;          // children[0] is visited first
			stack[sp++] = (uint32_t)c0;
		}
	}
	return any;
}

// ---- one (point, tet) candidate of a leaf pair: collisionDetectionSolidSolid's inner body -----------------------------------------
// pos: current positions (engine records, global indexing), x0: rest positions (global).  co1 owns the point, co2 the tet.
// vel: the engine's velocity records (vx, vy, vz, mass), global indexing; null: all velocities zero (host evaluations without velocities)
PBDX_HD bool tet_contact_candidate(const TetColliderView &co2, const P4 *pos, const P4 *x0, const P4 *vel, uint32_t particle, uint32_t tet, TetContact &out)
{
	const uint32_t off2 = co2.first;
	const V3 x_w = p3(pos[particle]);
	const uint32_t *ti = co2.tets + 4 * tet;
	V3 xa = p3(pos[ti[0] + off2]), xb = p3(pos[ti[1] + off2]), xc = p3(pos[ti[2] + off2]), xd = p3(pos[ti[3] + off2]);
	M3 A = cols3(xb - xa, xc - xa, xd - xa);
	const V3 bary = mul(inverse(A), x_w - xa);
	if (!(((double)bary.x >= 0.0) && ((double)bary.y >= 0.0) && ((double)bary.z >= 0.0) && ((double)(bary.x + bary.y + bary.z) <= 1.0)))
		return false;
	const V3 X0 = p3(x0[ti[0] + off2]), X1 = p3(x0[ti[1] + off2]), X2 = p3(x0[ti[2] + off2]), X3 = p3(x0[ti[3] + off2]);
	const V3 X = X0 + mul(cols3(X1 - X0, X2 - X0, X3 - X0), bary);
	const V3 Xi = mk(co2.X0[0], co2.X0[1], co2.X0[2]);
	const V3 X_l = mul_Rt(co2.R0, X - Xi);
	V3 cp_l, n_l; float dist_l;
	if (!sdf_collision_test(co2.sdf, X_l, 0.0f, cp_l, n_l, dist_l))
		return false;
	const V3 cp0 = mul_R(co2.R0, cp_l) + Xi;
	uint32_t cp_tet = 0; V3 cp_bary = mk(0.0f, 0.0f, 0.0f);
	if (!find_ref_tet_at(co2, x0 + off2, cp0, cp_tet, cp_bary))
		return false;
	if (cp_tet != tet)
	{
		ti = co2.tets + 4 * cp_tet;
		xa = p3(pos[ti[0] + off2]); xb = p3(pos[ti[1] + off2]); xc = p3(pos[ti[2] + off2]); xd = p3(pos[ti[3] + off2]);
		A = cols3(xb - xa, xc - xa, xd - xa);
	}
	const V3 cp_w = xa + mul(A, cp_bary);
	V3 n_w = cp_w - x_w;
	const float dist = norm(x_w - cp_w);
	if ((double)dist > 1.0e-6) n_w = n_w / dist;
	// ParticleTetContactConstraint::initConstraint + init_ParticleTetContactConstraint (normal and 1 / (J M^-1 J^T); the tangent and the
	// maximal tangent impulse only feed the velocity solve, see the header comment)
	out.particle = particle;
	out.solid = co2.body_index;
	out.tet = cp_tet;
	out.bary[0] = cp_bary.x; out.bary[1] = cp_bary.y; out.bary[2] = cp_bary.z;
	out.normal[0] = n_w.x; out.normal[1] = n_w.y; out.normal[2] = n_w.z;
	const V3 xv[4] = { xa, xb, xc, xd };
	for (int k = 0; k < 4; k++)
	{
		out.vert[k] = ti[k] + off2;
		out.x[k][0] = xv[k].x; out.x[k][1] = xv[k].y; out.x[k][2] = xv[k].z;
		out.w[k] = pos[ti[k] + off2].w;
	}
	const float bary0 = 1.0f - cp_bary.x - cp_bary.y - cp_bary.z;
	const float JMinvJT = pos[particle].w + bary0 * bary0 * out.w[0] + cp_bary.x * cp_bary.x * out.w[1] + cp_bary.y * cp_bary.y * out.w[2] + cp_bary.z * cp_bary.z * out.w[3];
	out.nKn_inv = 1.0f / JMinvJT;
	V3 vv[4], v0 = mk(0.0f, 0.0f, 0.0f);
	for (int k = 0; k < 4; k++) vv[k] = vel ? p3(vel[ti[k] + off2]) : mk(0.0f, 0.0f, 0.0f);
	if (vel) v0 = p3(vel[particle]);
	V3 t; float p_max;
	tet_contact_velocity_info(v0, vv, cp_bary, n_w, JMinvJT, t, p_max);
	out.tangent[0] = t.x; out.tangent[1] = t.y; out.tangent[2] = t.z;
	out.p_max = p_max;
	return true;
}

// ---- detection of one ordered pair (co1's points vs co2's tets): BVHTest::traverse + the leaf callback, in the reference's order ----
// emit(contact) is called in the order the reference appends to its contact list.  Returns false if the traversal stack overflowed.
template <class Emit>
PBDX_HD bool tet_pair_contacts(const TetColliderView &co1, const TetColliderView &co2, const P4 *pos, const P4 *x0, const P4 *vel, Emit &&emit)
{
	const BvhView &b1 = co1.points, &b2 = co2.tet_bvh;
	if (!b1.num_nodes || !b2.num_nodes) return true;
	struct Pair { uint32_t a, b; };
	Pair stack[128];
	int sp = 0;
	stack[sp].a = 0; stack[sp].b = 0; sp++;
	while (sp > 0)
	{
		const Pair pr = stack[--sp];
		const P4 bs1 = b1.hulls[pr.a], bs2 = b2.hulls[pr.b];
		if (!spheres_overlap(bs1, bs2)) continue;
		const int32_t a0 = b1.nodes[4 * pr.a], a1 = b1.nodes[4 * pr.a + 1], c0 = b2.nodes[4 * pr.b], c1 = b2.nodes[4 * pr.b + 1];
		const bool leaf1 = a0 < 0 && a1 < 0, leaf2 = c0 < 0 && c1 < 0;
		if (leaf1 && leaf2)
		{
			const uint32_t beg1 = (uint32_t)b1.nodes[4 * pr.a + 2], n1 = (uint32_t)b1.nodes[4 * pr.a + 3];
			const uint32_t beg2 = (uint32_t)b2.nodes[4 * pr.b + 2], n2 = (uint32_t)b2.nodes[4 * pr.b + 3];
			for (uint32_t i = beg1; i < beg1 + n1; i++)
				for (uint32_t j = beg2; j < beg2 + n2; j++)
				{
					TetContact c;
					if (tet_contact_candidate(co2, pos, x0, vel, b1.lst[i] + co1.first, b2.lst[j], c)) emit(c);
				}
			continue;
		}
		if (sp + 2 > 128) return false;
		// descend the smaller sphere's tree first (bs1.r() < bs2.r()), unless it is a leaf; children[0] before children[1]
		const bool descend1 = (bs1.w < bs2.w) ? !leaf1 : leaf2;
		if (descend1) { stack[sp].a = (uint32_t)a1; stack[sp].b = pr.b; sp++; stack[sp].a = (uint32_t)a0; stack[sp].b = pr.b; sp++; }
		else { stack[sp].a = pr.a; stack[sp].b = (uint32_t)c1; sp++; stack[sp].a = pr.a; stack[sp].b = (uint32_t)c0; sp++; }
	}
	return true;
}

// AABB::intersection of the two objects' boxes (CollisionDetection::updateAABB: min / max over the model's particles)
PBDX_HD bool aabb_intersect(const float *a /* min[3], max[3] */, const float *b)
{
	for (int i = 0; i < 3; i++)
	{
		const float min0 = a[i], max0 = a[3 + i], min1 = b[i], max1 = b[3 + i];
		if (((max0 < min1) || (min0 > max1))) return false;
	}
	return true;
}

// ---- solve_ParticleTetContactConstraint + ParticleTetContactConstraint::solvePositionConstraint ---------------------------------
// Reads the CURRENT position of the contact's particle, writes the particle and the four tet vertices (current positions).
template <class Pos>      // Pos: get(i) -> P4 (x, y, z, invMass), add(i, V3)
PBDX_HD void tet_contact_position_solve(const TetContact &c, Pos &p)
{
	const P4 q = p.get(c.particle);
	const float w0 = q.w;
	if ((w0 == 0.0f) && (c.w[0] == 0.0f) && (c.w[1] == 0.0f) && (c.w[2] == 0.0f))
		return;
	const float bary0 = 1.0f - c.bary[0] - c.bary[1] - c.bary[2];
	const V3 x0 = mk(c.x[0][0], c.x[0][1], c.x[0][2]), x1 = mk(c.x[1][0], c.x[1][1], c.x[1][2]), x2 = mk(c.x[2][0], c.x[2][1], c.x[2][2]), x3 = mk(c.x[3][0], c.x[3][1], c.x[3][2]);
	const V3 cp1 = ((bary0 * x0 + c.bary[0] * x1) + c.bary[1] * x2) + c.bary[2] * x3;
	const V3 normal = mk(c.normal[0], c.normal[1], c.normal[2]);
	const float C = dot(normal, p3(q) - cp1);
	const float lambda = -c.nKn_inv * C;
	const V3 pp = lambda * normal;
	if (w0 != 0.0f) p.add(c.particle, w0 * pp);          // pd.getMass != 0  <=>  invMass != 0 (ParticleData::setMass)
	if (c.w[0] != 0.0f) p.add(c.vert[0], (-c.w[0] * bary0) * pp);
	if (c.w[1] != 0.0f) p.add(c.vert[1], (-c.w[1] * c.bary[0]) * pp);
	if (c.w[2] != 0.0f) p.add(c.vert[2], (-c.w[2] * c.bary[1]) * pp);
	if (c.w[3] != 0.0f) p.add(c.vert[3], (-c.w[3] * c.bary[2]) * pp);
}

// the 34-float record of pbdx_solver_get_tet_contacts / pbdx_debug_tet_contacts (pbdx_tetcontact.cpp)
void contact_to_floats(const TetContact &c, float *o);

} // namespace pbdx
#endif
