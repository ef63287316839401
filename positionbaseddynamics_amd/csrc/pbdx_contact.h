// pbdx_contact.h -- particle vs static rigid body contacts with analytic distance fields
// (the contact part of TimeStepController::step for cloth / solids colliding
// with static DistanceFieldCollisionDetection objects, as in Demos/DistanceFieldDemos/ClothCollisionDemo.cpp).
//
// Restates, in the reference's operation order (Real = float; the distance functions evaluate in
// double exactly as the reference does):
//   DistanceFieldCollisionDetection::collisionDetectionRBSolid        DistanceFieldCollisionDetection.cpp:281-358
//   DistanceFieldCollision{Box,Sphere,Torus,Cylinder,HollowSphere,HollowBox}::distance / collisionTest  :598-716
//   DistanceFieldCollisionObject::approximateNormal / collisionTest   :669-716
//   PositionBasedRigidBodyDynamics::init_ParticleRigidBodyContactConstraint   PositionBasedRigidBodyDynamics.cpp:2385-2452
//   PositionBasedRigidBodyDynamics::velocitySolve_ParticleRigidBodyContactConstraint  :2455-2539
//   ParticleRigidBodyContactConstraint::solveVelocityConstraint       Constraints.cpp:2148-2189
// With static bodies (inverse mass 0) a contact changes nothing but its own particle's velocity, so the reference's
// sequential sweep over the contact list decomposes into independent per-particle chains (contacts of one particle keep
// their order = collider order): particle_contacts().  A contact with a DYNAMIC body also changes the body's velocity and
// angular velocity, which the next contact of that body reads: those contacts (and every other contact of their particles)
// are collected into ONE list in the reference's order and solved sequentially (dyn_contact_* below, dyn_contact_solve_kernel
// in pbdx_solver.hip); the body's own time integration stays with the host (rigid-body dynamics are outside the path).
#ifndef PBDX_CONTACT_H
#define PBDX_CONTACT_H

#include "pbdx_vec.h"
#include "../../include/pbdx.h"

namespace pbdx {

struct D3 { double x, y, z; };

PBDX_HD double dmax(double a, double b) { return (a < b) ? b : a; }      // std::max
PBDX_HD double dmin(double a, double b) { return (b < a) ? b : a; }      // std::min

// signed distance in the collider's local frame, in double, minus the tolerance
PBDX_HD double sdf_distance(const pbdx_collider &c, D3 x, float tolerance)
{
	const double inv = c.invert ? -1.0 : 1.0;      // m_invertSDF (Real) promoted
	switch (c.shape)
	{
	case PBDX_SHAPE_BOX:
	{
		const double dx = fabs(x.x) - (double)c.params[0], dy = fabs(x.y) - (double)c.params[1], dz = fabs(x.z) - (double)c.params[2];
		const double mx = dmax(dx, 0.0), my = dmax(dy, 0.0), mz = dmax(dz, 0.0);
		const double nrm = sqrt(mx * mx + (my * my + mz * mz));
		return inv * (dmin(dmax(dx, dmax(dy, dz)), 0.0) + nrm) - (double)tolerance;
	}
	case PBDX_SHAPE_SPHERE:
	{
		const double dl = sqrt(x.x * x.x + (x.y * x.y + x.z * x.z));
		return inv * (dl - (double)c.params[0]) - (double)tolerance;
	}
	case PBDX_SHAPE_TORUS:
	{
		// Vector2r(x.x(), x.z()).norm(): narrowed to Real first
		const float fx = (float)x.x, fz = (float)x.z;
		const float l = sqrtf(fx * fx + fz * fz);
		const double qx = (double)l - (double)c.params[0], qy = x.y;
		return inv * (sqrt(qx * qx + qy * qy) - (double)c.params[1]) - (double)tolerance;
	}
	case PBDX_SHAPE_CYLINDER:
	{
		const double l = sqrt(x.x * x.x + x.z * x.z);
		const double dx = fabs(l) - (double)c.params[0], dy = fabs(x.y) - (double)c.params[1];
		const double mx = dmax(dx, 0.0), my = dmax(dy, 0.0);
		return inv * (dmin(dmax(dx, dy), 0.0) + sqrt(mx * mx + my * my)) - (double)tolerance;
	}
	case PBDX_SHAPE_HOLLOW_SPHERE:
	{
		const double dl = sqrt(x.x * x.x + (x.y * x.y + x.z * x.z));
		return inv * (fabs(dl - (double)c.params[0]) - (double)c.params[1]) - (double)tolerance;
	}
	case PBDX_SHAPE_HOLLOW_BOX:
	{
		const double dx = fabs(x.x) - (double)c.params[0], dy = fabs(x.y) - (double)c.params[1], dz = fabs(x.z) - (double)c.params[2];
		const double mx = dmax(dx, 0.0), my = dmax(dy, 0.0), mz = dmax(dz, 0.0);
		// d.maxCoeff(): Eigen visits coefficients in order keeping the running maximum
		double mc = dx; if (dy > mc) mc = dy; if (dz > mc) mc = dz;
		// m_thickness is Real: (fabs(...) - m_thickness) promotes it
		return inv * (fabs(dmin(mc, 0.0) + sqrt(mx * mx + (my * my + mz * mz))) - (double)c.params[3]) - (double)tolerance;
	}
	default: return 1.0e300;
	}
}

// DistanceFieldCollisionObject::approximateNormal
PBDX_HD V3 sdf_normal(const pbdx_collider &c, D3 x, float tolerance)
{
	const double eps = 1.e-6;
	double comp[3] = { x.x, x.y, x.z };
	float n[3];
	for (int j = 0; j < 3; j++)
	{
		double tmp[3] = { comp[0], comp[1], comp[2] };
		tmp[j] += eps;
		D3 xp; xp.x = tmp[0]; xp.y = tmp[1]; xp.z = tmp[2];
		const double e_p = sdf_distance(c, xp, tolerance);
		tmp[j] = comp[j] - eps;
		D3 xm; xm.x = tmp[0]; xm.y = tmp[1]; xm.z = tmp[2];
		const double e_m = sdf_distance(c, xm, tolerance);
		const double res = (e_p - e_m) * (1.0 / (2.0 * eps));
		n[j] = (float)res;
	}
	V3 nn = mk(n[0], n[1], n[2]);
	const float norm2 = sqn(nn);
	if ((double)norm2 < 1.e-6)
		return mk(0.0f, 0.0f, 0.0f);
	return nn / sqrtf(norm2);
}

// collisionTest in the collider's local frame (maxDist = 0).  cp, n in local coordinates.
PBDX_HD bool sdf_collision_test(const pbdx_collider &c, V3 x, float tolerance, V3 &cp, V3 &n, float &dist)
{
	const float inv = c.invert ? -1.0f : 1.0f;
	if (c.shape == PBDX_SHAPE_SPHERE)
	{
		const float dl = norm(x);
		dist = inv * (dl - c.params[0]) - tolerance;
		if (dist < 0.0f)
		{
			if ((double)dl < 1.e-6) n = mk(0.0f, 0.0f, 0.0f);
			else n = (inv * x) / dl;
			cp = (c.params[0] + tolerance) * n;
			return true;
		}
		return false;
	}
	if (c.shape == PBDX_SHAPE_HOLLOW_SPHERE)
	{
		const float dl = norm(x);
		dist = inv * (fabsf(dl - c.params[0]) - c.params[1]) - tolerance;
		if (dist < 0.0f)
		{
			if ((double)dl < 1.e-6) n = mk(0.0f, 0.0f, 0.0f);
			else if (dl < c.params[0]) n = ((-inv) * x) / dl;
			else n = (inv * x) / dl;
			cp = x - dist * n;
			return true;
		}
		return false;
	}
	D3 xd; xd.x = (double)x.x; xd.y = (double)x.y; xd.z = (double)x.z;
	dist = (float)sdf_distance(c, xd, tolerance);
	if (dist < 0.0f)
	{
		n = sdf_normal(c, xd, tolerance);
		cp = x - dist * n;
		return true;
	}
	return false;
}

// R (row-major 3x3) * v and R^T * v with Eigen's coefficient order  c0 + (c1 + c2)
PBDX_HD V3 mul_R(const float *R, V3 v)
{
	return mk(R[0] * v.x + (R[1] * v.y + R[2] * v.z), R[3] * v.x + (R[4] * v.y + R[5] * v.z), R[6] * v.x + (R[7] * v.y + R[8] * v.z));
}
PBDX_HD V3 mul_Rt(const float *R, V3 v)
{
	return mk(R[0] * v.x + (R[3] * v.y + R[6] * v.z), R[1] * v.x + (R[4] * v.y + R[7] * v.z), R[2] * v.x + (R[5] * v.y + R[8] * v.z));
}

struct ContactInfo     // m_constraintInfo (3x5) + m_sum_impulses + per-contact coefficients
{
	V3 cp0, cp1, normal, tangent;
	float nKn_inv, pMax, goal;
	float sum_impulses;
	float friction;
	V3 x1, v1, omega1;        // body state (static: velocities zero)
};

// init_ParticleRigidBodyContactConstraint for a static body (invMass1 == 0: computeMatrixK zeroes K)
PBDX_HD void contact_init(float invMass0, V3 v0, V3 x1, V3 v1, V3 omega1, V3 cp0, V3 cp1, V3 normal, float restitution, ContactInfo &ci)
{
	const V3 r1 = cp1 - x1;
	const V3 u1 = v1 + cross(omega1, r1);
	const V3 u_rel = v0 - u1;
	const float u_rel_n = dot(normal, u_rel);
	ci.cp0 = cp0; ci.cp1 = cp1; ci.normal = normal;
	V3 t = u_rel - u_rel_n * normal;
	const float tl2 = sqn(t);
	if ((double)tl2 > 1.0e-6)
		t = t * (1.0f / sqrtf(tl2));
	ci.tangent = t;
	// K = 0 (static body) with invMass0 added on the diagonal; the products below keep the zero terms of
	// the reference's full 3x3 arithmetic (they only matter for the sign of zero)
	float K[3][3] = { { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f } };
	if (invMass0 != 0.0f) { K[0][0] += invMass0; K[1][1] += invMass0; K[2][2] += invMass0; }
	const V3 Kn = mk(K[0][0] * normal.x + (K[0][1] * normal.y + K[0][2] * normal.z), K[1][0] * normal.x + (K[1][1] * normal.y + K[1][2] * normal.z),
		K[2][0] * normal.x + (K[2][1] * normal.y + K[2][2] * normal.z));
	ci.nKn_inv = 1.0f / dot(normal, Kn);
	const V3 Kt = mk(K[0][0] * t.x + (K[0][1] * t.y + K[0][2] * t.z), K[1][0] * t.x + (K[1][1] * t.y + K[1][2] * t.z),
		K[2][0] * t.x + (K[2][1] * t.y + K[2][2] * t.z));
	ci.pMax = 1.0f / dot(t, Kt) * dot(u_rel, t);
	ci.goal = 0.0f;
	if (u_rel_n < 0.0f)
		ci.goal = -restitution * u_rel_n;
	ci.sum_impulses = 0.0f;
	ci.x1 = x1; ci.v1 = v1; ci.omega1 = omega1;
}

// velocitySolve_ParticleRigidBodyContactConstraint; returns false when nothing is applied
PBDX_HD bool contact_velocity_solve(float invMass0, V3 v0, float stiffness, ContactInfo &ci, V3 &corr_v0)
{
	if (invMass0 == 0.0f)          // invMass1 == 0 for a static body
		return false;
	const float d = dot(ci.normal, ci.cp0 - ci.cp1);
	const V3 r1 = ci.cp1 - ci.x1;
	const V3 u1 = ci.v1 + cross(ci.omega1, r1);
	const V3 u_rel = v0 - u1;
	const float u_rel_n = dot(u_rel, ci.normal);
	const float delta_u_reln = ci.goal - u_rel_n;
	float correctionMagnitude = ci.nKn_inv * delta_u_reln;
	if (correctionMagnitude < -ci.sum_impulses)
		correctionMagnitude = -ci.sum_impulses;
	if (d < 0.0f)
		correctionMagnitude -= stiffness * ci.nKn_inv * d;
	V3 p = correctionMagnitude * ci.normal;
	ci.sum_impulses += correctionMagnitude;
	const float pn = dot(p, ci.normal);
	if (ci.friction * pn > ci.pMax)
		p = p - ci.pMax * ci.tangent;
	else if (ci.friction * pn < -ci.pMax)
		p = p + ci.pMax * ci.tangent;
	else
		p = p - (ci.friction * pn) * ci.tangent;
	corr_v0 = invMass0 * p;
	return true;
}

#define PBDX_MAX_CONTACTS_PER_PARTICLE 8

// Detection of ONE particle against the colliders in order (collisionDetectionRBSolid's leaf test, DistanceFieldCollisionDetection.cpp:334-356).
// Returns the number of contacts (or -1 on overflow).
struct RawContact { uint32_t collider; V3 cp_w, n_w; };
PBDX_HD int detect_particle_contacts(V3 x, const pbdx_collider *colliders, uint32_t num_colliders, float tolerance, RawContact *out)
{
	int nc = 0;
	for (uint32_t k = 0; k < num_colliders; k++)
	{
		const pbdx_collider &c = colliders[k];
		const V3 com = mk(c.com[0], c.com[1], c.com[2]);
		const V3 xl = mul_R(c.R, x - com) + mk(c.v1[0], c.v1[1], c.v1[2]);
		V3 cp, n; float dist;
		if (!sdf_collision_test(c, xl, tolerance, cp, n, dist))
			continue;
		if (nc >= PBDX_MAX_CONTACTS_PER_PARTICLE)
			return -1;
		out[nc].collider = k;
		out[nc].cp_w = mul_Rt(c.R, cp) + mk(c.v2[0], c.v2[1], c.v2[2]);
		out[nc].n_w = mul_Rt(c.R, n);
		nc++;
	}
	return nc;
}

// ---- contacts with dynamic bodies ---------------------------------------------------------------------------------------
// PositionBasedRigidBodyDynamics::computeMatrixK (one connector)   PositionBasedRigidBodyDynamics.cpp:11-45
PBDX_HD void compute_matrix_k(V3 connector, float invMass, V3 x, const float *Ji /* inertiaInverseW, row-major */, float K[3][3])
{
	if (invMass != 0.0f)
	{
		const V3 v = connector - x;
		const float a = v.x, b = v.y, c = v.z;
		const float j11 = Ji[0], j12 = Ji[1], j13 = Ji[2], j22 = Ji[4], j23 = Ji[5], j33 = Ji[8];
		K[0][0] = c * c * j22 - b * c * (j23 + j23) + b * b * j33 + invMass;
		K[0][1] = -(c * c * j12) + a * c * j23 + b * c * j13 - a * b * j33;
		K[0][2] = b * c * j12 - a * c * j22 - b * b * j13 + a * b * j23;
		K[1][0] = K[0][1];
		K[1][1] = c * c * j11 - a * c * (j13 + j13) + a * a * j33 + invMass;
		K[1][2] = -(b * c * j11) + a * c * j12 + a * b * j13 - a * a * j23;
		K[2][0] = K[0][2];
		K[2][1] = K[1][2];
		K[2][2] = b * b * j11 - a * b * (j12 + j12) + a * a * j22 + invMass;
	}
	else
		for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K[i][j] = 0.0f;
}
// init_ParticleRigidBodyContactConstraint for any body (:2385-2452): the constants of a contact.  For invMass1 == 0 the same values as contact_init.
struct DynContactInfo { V3 cp0, cp1, normal, tangent; float nKn_inv, pMax, goal; };
PBDX_HD void dyn_contact_init(float invMass0, V3 v0, float invMass1, V3 x1, V3 v1, const float *Ji, V3 omega1, V3 cp0, V3 cp1, V3 normal, float restitution, DynContactInfo &ci)
{
	const V3 r1 = cp1 - x1;
	const V3 u1 = v1 + cross(omega1, r1);
	const V3 u_rel = v0 - u1;
	const float u_rel_n = dot(normal, u_rel);
	ci.cp0 = cp0; ci.cp1 = cp1; ci.normal = normal;
	V3 t = u_rel - u_rel_n * normal;
	const float tl2 = sqn(t);
	if ((double)tl2 > 1.0e-6)
		t = t * (1.0f / sqrtf(tl2));
	ci.tangent = t;
	float K[3][3];
	compute_matrix_k(cp1, invMass1, x1, Ji, K);
	if (invMass0 != 0.0f) { K[0][0] += invMass0; K[1][1] += invMass0; K[2][2] += invMass0; }
	const V3 Kn = mk(K[0][0] * normal.x + (K[0][1] * normal.y + K[0][2] * normal.z), K[1][0] * normal.x + (K[1][1] * normal.y + K[1][2] * normal.z),
		K[2][0] * normal.x + (K[2][1] * normal.y + K[2][2] * normal.z));
	ci.nKn_inv = 1.0f / dot(normal, Kn);
	const V3 Kt = mk(K[0][0] * t.x + (K[0][1] * t.y + K[0][2] * t.z), K[1][0] * t.x + (K[1][1] * t.y + K[1][2] * t.z),
		K[2][0] * t.x + (K[2][1] * t.y + K[2][2] * t.z));
	ci.pMax = 1.0f / dot(t, Kt) * dot(u_rel, t);
	ci.goal = 0.0f;
	if (u_rel_n < 0.0f)
		ci.goal = -restitution * u_rel_n;
}
// velocitySolve_ParticleRigidBodyContactConstraint (:2455-2539) + ParticleRigidBodyContactConstraint::solveVelocityConstraint (Constraints.cpp:2148-2189):
// the impulse of one contact against the CURRENT velocities; v0 (if mass0 != 0) and the body's v1 / omega1 (if invMass1 != 0) are updated in place
PBDX_HD void dyn_contact_velocity_solve(float invMass0, float mass0, V3 &v0, float invMass1, V3 x1, V3 &v1, const float *Ji, V3 &omega1, float stiffness, float friction,
	float &sum_impulses, const DynContactInfo &ci)
{
	if (invMass0 == 0.0f && invMass1 == 0.0f)
		return;
	const float d = dot(ci.normal, ci.cp0 - ci.cp1);
	const V3 r1 = ci.cp1 - x1;
	const V3 u1 = v1 + cross(omega1, r1);
	const V3 u_rel = v0 - u1;
	const float u_rel_n = dot(u_rel, ci.normal);
	const float delta_u_reln = ci.goal - u_rel_n;
	float correctionMagnitude = ci.nKn_inv * delta_u_reln;
	if (correctionMagnitude < -sum_impulses)
		correctionMagnitude = -sum_impulses;
	if (d < 0.0f)
		correctionMagnitude -= stiffness * ci.nKn_inv * d;
	V3 p = correctionMagnitude * ci.normal;
	sum_impulses += correctionMagnitude;
	const float pn = dot(p, ci.normal);
	if (friction * pn > ci.pMax)
		p = p - ci.pMax * ci.tangent;
	else if (friction * pn < -ci.pMax)
		p = p + ci.pMax * ci.tangent;
	else
		p = p - (friction * pn) * ci.tangent;
	if (invMass0 != 0.0f && mass0 != 0.0f)
		v0 = v0 + invMass0 * p;
	if (invMass1 != 0.0f)
	{
		v1 = v1 + (-invMass1) * p;                    // corr_v1 = -invMass1 * p
		omega1 = omega1 + mul_R(Ji, cross(r1, -p));   // corr_omega1 = inertiaInverseW1 * (r1 x -p)
	}
}

// All contacts of ONE particle: detection against the colliders in order, contact initialisation with
// the pre-solve velocity, `iterations` velocity sweeps.  Returns the number of contacts (or -1 on overflow).
// `extra.after_sweep(v)`: what else changes this particle's velocity at the end of every iteration of velocityConstraintProjection
// (TimeStepController.cpp:342-355: after the particle-rigid-body contacts come the particle-tet contacts, whose impulses are constants).
struct NoExtraImpulses { PBDX_HD void after_sweep(V3 &) const {} };
// initialisation with the pre-solve velocity and `iterations` velocity sweeps over the contacts of ONE particle that touch static bodies only (raw: in collider order)
template <class Extra>
PBDX_HD void solve_particle_contacts(V3 x, V3 &v, float invMass, float mass, const pbdx_collider *colliders, const RawContact *raw, int nc,
	float stiffness, float model_restitution, float model_friction, uint32_t iterations, const Extra &extra)
{
	ContactInfo ci[PBDX_MAX_CONTACTS_PER_PARTICLE];
	for (int q = 0; q < nc; q++)
	{
		const pbdx_collider &c = colliders[raw[q].collider];
		contact_init(invMass, v, mk(c.com[0], c.com[1], c.com[2]), mk(c.body_v[0], c.body_v[1], c.body_v[2]), mk(c.body_omega[0], c.body_omega[1], c.body_omega[2]),
			x, raw[q].cp_w, raw[q].n_w, model_restitution * c.restitution, ci[q]);
		ci[q].friction = model_friction + c.friction;
	}
	for (uint32_t it = 0; it < iterations; it++)
	{
		for (int k = 0; k < nc; k++)
		{
			V3 corr;
			if (contact_velocity_solve(invMass, v, stiffness, ci[k], corr))
				if (mass != 0.0f)
					v = v + corr;
		}
		extra.after_sweep(v);
	}
}
template <class Extra>
PBDX_HD int particle_contacts(V3 x, V3 &v, float invMass, float mass, const pbdx_collider *colliders, uint32_t num_colliders,
	float tolerance, float stiffness, float model_restitution, float model_friction, uint32_t iterations, const Extra &extra)
{
	RawContact raw[PBDX_MAX_CONTACTS_PER_PARTICLE];
	const int nc = detect_particle_contacts(x, colliders, num_colliders, tolerance, raw);
	if (nc < 0) return -1;
	solve_particle_contacts(x, v, invMass, mass, colliders, raw, nc, stiffness, model_restitution, model_friction, iterations, extra);
	return nc;
}
PBDX_HD int particle_contacts(V3 x, V3 &v, float invMass, float mass, const pbdx_collider *colliders, uint32_t num_colliders,
	float tolerance, float stiffness, float model_restitution, float model_friction, uint32_t iterations)
{
	return particle_contacts(x, v, invMass, mass, colliders, num_colliders, tolerance, stiffness, model_restitution, model_friction, iterations, NoExtraImpulses());
}

} // namespace pbdx

#endif
