// pbdx_pair.h -- two constraints per lane with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32).
//
// STATUS: opt-in (PBDX_OPT_PAIRS), off by default.  Measured on MI355X (1M-particle cloth): pairing the
// distance steps made their segment 10 % SLOWER (49.2 -> 53.9 us), pairing the bending steps needs 168 VGPRs
// (<= 768 threads per tile) and lost 20-30 %.  With four waves per SIMD the thread-level parallelism already
// covers the dependency chains; the second slot only adds register pressure.  Kept because it is verified
// bit-identical and documents the design point.
//
// Why it was tried: inside a fused tile a colour step holds ~1 000-2 800 slots for 1 024 threads, so many steps
// need a second (or third) chunk that only a few waves execute while the others wait at the colour
// barrier.  Executing two chunks of the SAME step jointly -- one slot of each per lane, the two
// independent instruction streams written as 2-wide vector arithmetic -- halves the multiply/add
// instruction count (CDNA's packed fp32 ALU) and gives the in-order wave two independent dependency
// chains through the IEEE divisions and square roots.
//
// Exactness: every lane of a packed operation is an ordinary IEEE fp32 operation, the association
// order is the one of pbdx_vec.h / pbdx_project.h, data-dependent branches of the scalar code are
// turned into selects between the same two values.  The paired functions therefore return exactly
// the bits of the scalar functions (checked by the GPU tests: paired and unpaired schedules are
// bit-identical), which are the bits of the reference.
#ifndef PBDX_PAIR_H
#define PBDX_PAIR_H

#include "pbdx_project.h"

namespace pbdx {

typedef float f2 __attribute__((ext_vector_type(2)));
struct B2 { bool a, b; };
struct V3P { f2 x, y, z; };        // two 3-vectors, component-wise packed

__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 splat(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 sel(B2 c, f2 t, f2 f) { return mk2(c.a ? t.x : f.x, c.b ? t.y : f.y); }
__device__ __forceinline__ B2 band(B2 p, B2 q) { B2 r = { p.a && q.a, p.b && q.b }; return r; }
__device__ __forceinline__ B2 gt(f2 v, float s) { B2 r = { v.x > s, v.y > s }; return r; }
__device__ __forceinline__ B2 ne0(f2 v) { B2 r = { v.x != 0.0f, v.y != 0.0f }; return r; }
__device__ __forceinline__ f2 fabs2(f2 v) { return mk2(fabsf(v.x), fabsf(v.y)); }
__device__ __forceinline__ f2 sqrt2(f2 v) { return mk2(sqrtf(v.x), sqrtf(v.y)); }
__device__ __forceinline__ f2 div2(f2 n, f2 d) { return mk2(n.x / d.x, n.y / d.y); }   // two correctly rounded divisions

__device__ __forceinline__ V3P mkp(V3 a, V3 b) { V3P r; r.x = mk2(a.x, b.x); r.y = mk2(a.y, b.y); r.z = mk2(a.z, b.z); return r; }
__device__ __forceinline__ V3 lane0(V3P v) { return mk(v.x.x, v.y.x, v.z.x); }
__device__ __forceinline__ V3 lane1(V3P v) { return mk(v.x.y, v.y.y, v.z.y); }
__device__ __forceinline__ V3P operator+(V3P a, V3P b) { V3P r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
__device__ __forceinline__ V3P operator-(V3P a, V3P b) { V3P r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
__device__ __forceinline__ V3P operator*(f2 s, V3P a) { V3P r; r.x = s * a.x; r.y = s * a.y; r.z = s * a.z; return r; }
__device__ __forceinline__ V3P operator*(V3P a, f2 s) { V3P r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
__device__ __forceinline__ V3P zero3p() { V3P r; r.x = splat(0.0f); r.y = r.x; r.z = r.x; return r; }
__device__ __forceinline__ V3P sel(B2 c, V3P t, V3P f) { V3P r; r.x = sel(c, t.x, f.x); r.y = sel(c, t.y, f.y); r.z = sel(c, t.z, f.z); return r; }
// same association as pbdx_vec.h: c0 + (c1 + c2)
__device__ __forceinline__ f2 dot(V3P a, V3P b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ f2 sqn(V3P a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }

// DistanceConstraint_XPBD  (scalar: solve_distance_xpbd, XPBD.cpp:14-60).  `apply` lanes always get a
// correction (possibly zero), exactly like the scalar function which returns true on its early outs.
__device__ __forceinline__ void solve_distance_xpbd2(V3P p0, f2 w0, V3P p1, f2 w1, f2 restLength, f2 stiffness,
	float dt, f2 &lambda, V3P &c0, V3P &c1)
{
	const f2 K0 = w0 + w1;
	const V3P n = p0 - p1;
	const f2 d = sqrt2(sqn(n));
	const f2 C = d - restLength;
	const B2 m1 = gt(d, 1e-6f);
	V3P nn; nn.x = div2(n.x, d); nn.y = div2(n.y, d); nn.z = div2(n.z, d);
	const B2 hasK = ne0(stiffness);
	const f2 alpha = sel(hasK, div2(splat(1.0f), stiffness * dt * dt), splat(0.0f));
	const f2 K = sel(hasK, K0 + alpha, K0);
	const B2 m2 = gt(fabs2(K), 1e-6f);
	const f2 Kinv = div2(splat(1.0f), K);
	const f2 delta_lambda = -Kinv * (C + alpha * lambda);
	const B2 ok = band(m1, m2);
	lambda = sel(ok, lambda + delta_lambda, lambda);
	const V3P pt = nn * delta_lambda;
	c0 = sel(ok, w0 * pt, zero3p());
	c1 = sel(ok, (-w1) * pt, zero3p());
}

// IsometricBendingConstraint_XPBD  (scalar: solve_isometric_bending_xpbd, XPBD.cpp:153-213).
// q[k*4+j] = {Q0(j,k), Q1(j,k)}.  Returns per lane whether corrections apply (scalar: return value).
__device__ __forceinline__ B2 solve_isometric_bending_xpbd2(const V3P p[4], const f2 wq[4], const f2 q[16], f2 stiffness,
	float dt, f2 &lambda, V3P c[4])
{
	const V3P x[4] = { p[2], p[3], p[0], p[1] };
	const f2 w[4] = { wq[2], wq[3], wq[0], wq[1] };
	f2 energy = splat(0.0f);
#pragma unroll
	for (int k = 0; k < 4; k++)
#pragma unroll
		for (int j = 0; j < 4; j++)
			energy = energy + q[k * 4 + j] * dot(x[k], x[j]);
	energy = energy * splat(0.5f);
	V3P g[4];
#pragma unroll
	for (int j = 0; j < 4; j++) g[j] = zero3p();
#pragma unroll
	for (int k = 0; k < 4; k++)
#pragma unroll
		for (int j = 0; j < 4; j++)
			g[j] = g[j] + q[k * 4 + j] * x[k];
	f2 sum = splat(0.0f);
#pragma unroll
	for (int j = 0; j < 4; j++)
		sum = sel(ne0(w[j]), sum + w[j] * sqn(g[j]), sum);
	const B2 hasK = ne0(stiffness);
	const f2 alpha = sel(hasK, div2(splat(1.0f), stiffness * dt * dt), splat(0.0f));
	sum = sel(hasK, sum + alpha, sum);
	const B2 ok = gt(fabs2(sum), PBDX_EPS);
	const f2 delta_lambda = div2(-(energy + alpha * lambda), sum);
	lambda = sel(ok, lambda + delta_lambda, lambda);
	c[0] = (delta_lambda * w[2]) * g[2];
	c[1] = (delta_lambda * w[3]) * g[3];
	c[2] = (delta_lambda * w[0]) * g[0];
	c[3] = (delta_lambda * w[1]) * g[1];
	return ok;
}

} // namespace pbdx

#endif
