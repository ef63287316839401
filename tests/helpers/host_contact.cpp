// Test helper (never part of the product library): compiles the engine's contact arithmetic
// (positionbaseddynamics_amd/csrc/pbdx_contact.h, host+device code) for the HOST so that the CPU test suite
// can pin it against the reference's DistanceFieldCollisionDetection / ParticleRigidBodyContactConstraint
// without a GPU.  Built on the fly by tests/test_contact_math.py with g++ -ffp-contract=off.
#include "../../positionbaseddynamics_amd/csrc/pbdx_contact.h"
using namespace pbdx;
extern "C" int host_particle_contacts(const float *x, float *v, float invMass, float mass, const pbdx_collider *cols, unsigned n,
	float tol, float stiff, float mrest, float mfric, unsigned iters)
{
	V3 V = mk(v[0], v[1], v[2]);
	const int nc = particle_contacts(mk(x[0], x[1], x[2]), V, invMass, mass, cols, n, tol, stiff, mrest, mfric, iters);
	v[0] = V.x; v[1] = V.y; v[2] = V.z;
	return nc;
}
