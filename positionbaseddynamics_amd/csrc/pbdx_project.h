// pbdx_project.h -- per-constraint projection arithmetic of the PBD/XPBD hot
// path, written for one constraint per lane (all state in registers).
//
// Each function restates one `Constraint::solvePositionConstraint` body of the
// reference (wrapper in Simulation/Constraints.cpp + static solver in
// PositionBasedDynamics/{PositionBasedDynamics,XPBD}.cpp) and keeps the
// reference's floating-point operation order (see pbdx_vec.h), so that a
// contraction-free float build of the reference is reproduced to the last bit
// wherever libm is not involved.  `w` are inverse masses.  Every function
// returns true when the reference would apply the corrections; the caller then
// adds corr[i] to particle i only where w[i] != 0 (Constraints.cpp:1198-1204).
#ifndef PBDX_PROJECT_H
#define PBDX_PROJECT_H

#include "pbdx_vec.h"

namespace pbdx {

#define PBDX_EPS 1e-6f

// ---------------------------------------------------------------------------
// DistanceConstraint            PositionBasedDynamics.cpp:13-34
PBDX_HD bool solve_distance(V3 p0, float w0, V3 p1, float w1, float restLength, float stiffness,
	V3 &c0, V3 &c1)
{
	const float wSum = w0 + w1;
	if (wSum == 0.0f)
		return false;
	V3 n = p1 - p0;
	const float d = norm(n);
	n = normalized(n);
	const float dl = d - restLength;
	const V3 corr = mk(((stiffness * n.x) * dl) / wSum, ((stiffness * n.y) * dl) / wSum, ((stiffness * n.z) * dl) / wSum);
	c0 = w0 * corr;
	c1 = (-w1) * corr;
	return true;
}

// DistanceConstraint_XPBD       XPBD.cpp:14-60
PBDX_HD bool solve_distance_xpbd(V3 p0, float w0, V3 p1, float w1, float restLength, float stiffness,
	float dt, float &lambda, V3 &c0, V3 &c1)
{
	float K = w0 + w1;
	V3 n = p0 - p1;
	const float d = norm(n);
	const float C = d - restLength;
	// (the zero corrections of the two degenerate exits are assigned IN those exits, which are marked unlikely: initialising c0 / c1 up front, as the
	// reference does, costs the device six v_mov per projection on the path every constraint takes -- same values on every path)
	if (__builtin_expect(d > 1e-6f, 1))
		n = n / d;
	else
	{
		c0 = mk(0.0f, 0.0f, 0.0f); c1 = c0;
		return true;
	}
	float alpha = 0.0f;
	if (stiffness != 0.0f)
	{
		alpha = 1.0f / (stiffness * dt * dt);
		K += alpha;
	}
	float Kinv = 0.0f;
	if (__builtin_expect(fabsf(K) > 1e-6f, 1))
		Kinv = 1.0f / K;
	else
	{
		c0 = mk(0.0f, 0.0f, 0.0f); c1 = c0;
		return true;
	}
	const float delta_lambda = -Kinv * (C + alpha * lambda);
	lambda += delta_lambda;
	const V3 pt = n * delta_lambda;
	c0 = w0 * pt;
	c1 = (-w1) * pt;
	return true;
}

// DihedralConstraint            PositionBasedDynamics.cpp:37-102
PBDX_HD bool solve_dihedral(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	float restAngle, float stiffness, V3 &c0, V3 &c1, V3 &c2, V3 &c3)
{
	if (w0 == 0.0f && w1 == 0.0f)
		return false;
	const V3 e = p3 - p2;
	const float elen = norm(e);
	if (elen < PBDX_EPS)
		return false;
	const float invElen = 1.0f / elen;

	V3 n1 = cross(p2 - p0, p3 - p0); n1 = n1 / sqn(n1);
	V3 n2 = cross(p3 - p1, p2 - p1); n2 = n2 / sqn(n2);

	const V3 d0 = elen * n1;
	const V3 d1 = elen * n2;
	const V3 d2 = (dot(p0 - p3, e) * invElen) * n1 + (dot(p1 - p3, e) * invElen) * n2;
	const V3 d3 = (dot(p2 - p0, e) * invElen) * n1 + (dot(p2 - p1, e) * invElen) * n2;

	n1 = normalized(n1);
	n2 = normalized(n2);
	float dt = dot(n1, n2);
	if (dt < -1.0f) dt = -1.0f;
	if (dt > 1.0f) dt = 1.0f;
	// PositionBasedDynamics.cpp:73 calls ::acos(double) (no <math.h> C++ overloads in that TU) and narrows
	const float phi = (float)acos((double)dt);

	float lambda = w0 * sqn(d0) + w1 * sqn(d1) + w2 * sqn(d2) + w3 * sqn(d3);
	if (lambda == 0.0f)
		return false;
	lambda = (phi - restAngle) / lambda * stiffness;
	if (dot(cross(n1, n2), e) > 0.0f)
		lambda = -lambda;
	c0 = (-w0 * lambda) * d0;
	c1 = (-w1 * lambda) * d1;
	c2 = (-w2 * lambda) * d2;
	c3 = (-w3 * lambda) * d3;
	return true;
}

// VolumeConstraint              PositionBasedDynamics.cpp:104-142
PBDX_HD bool solve_volume(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	float restVolume, float stiffness, V3 &c0, V3 &c1, V3 &c2, V3 &c3)
{
	const float volume = (float)(1.0 / 6.0) * dot(cross(p1 - p0, p2 - p0), p3 - p0);
	c0 = mk(0.0f, 0.0f, 0.0f); c1 = c0; c2 = c0; c3 = c0;
	if (stiffness == 0.0f)
		return false;
	const V3 g0 = cross(p1 - p2, p3 - p2);
	const V3 g1 = cross(p2 - p0, p3 - p0);
	const V3 g2 = cross(p0 - p1, p3 - p1);
	const V3 g3 = cross(p1 - p0, p2 - p0);
	float lambda = w0 * sqn(g0) + w1 * sqn(g1) + w2 * sqn(g2) + w3 * sqn(g3);
	if (fabsf(lambda) < PBDX_EPS)
		return false;
	lambda = stiffness * (volume - restVolume) / lambda;
	c0 = (-lambda * w0) * g0;
	c1 = (-lambda * w1) * g1;
	c2 = (-lambda * w2) * g2;
	c3 = (-lambda * w3) * g3;
	return true;
}

// VolumeConstraint_XPBD         XPBD.cpp:63-109
PBDX_HD bool solve_volume_xpbd(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	float restVolume, float stiffness, float dt, float &lambda, V3 &c0, V3 &c1, V3 &c2, V3 &c3)
{
	const float volume = (float)(1.0 / 6.0) * dot(cross(p1 - p0, p2 - p0), p3 - p0);
	const V3 g0 = cross(p1 - p2, p3 - p2);
	const V3 g1 = cross(p2 - p0, p3 - p0);
	const V3 g2 = cross(p0 - p1, p3 - p1);
	const V3 g3 = cross(p1 - p0, p2 - p0);
	float K = w0 * sqn(g0) + w1 * sqn(g1) + w2 * sqn(g2) + w3 * sqn(g3);
	float alpha = 0.0f;
	if (stiffness != 0.0f)
	{
		alpha = 1.0f / (stiffness * dt * dt);
		K += alpha;
	}
	if (fabsf(K) < PBDX_EPS)
		return false;
	const float C = volume - restVolume;
	const float delta_lambda = -(C + alpha * lambda) / K;
	lambda += delta_lambda;
	c0 = (delta_lambda * w0) * g0;
	c1 = (delta_lambda * w1) * g1;
	c2 = (delta_lambda * w2) * g2;
	c3 = (delta_lambda * w3) * g3;
	return true;
}

// IsometricBendingConstraint    PositionBasedDynamics.cpp:186-236
// IsometricBendingConstraint_XPBD  XPBD.cpp:153-213
// Q is accessed through a functor q(j,k) so that callers can supply the full
// 4x4 matrix or the factored form K[j]*K2[k] (k<=j) the reference's init builds
// it from (PositionBasedDynamics.cpp:169-180); both give identical bits.
template <typename QF>
PBDX_HD void isometric_energy_grad(const V3 x[4], const QF &q, float &energy, V3 g[4])
{
	energy = 0.0f;
	for (int k = 0; k < 4; k++)
		for (int j = 0; j < 4; j++)
			energy += q(j, k) * dot(x[k], x[j]);
	energy *= 0.5f;
	for (int j = 0; j < 4; j++) g[j] = mk(0.0f, 0.0f, 0.0f);
	for (int k = 0; k < 4; k++)
		for (int j = 0; j < 4; j++)
			g[j] = g[j] + q(j, k) * x[k];
}

template <typename QF>
PBDX_HD bool solve_isometric_bending(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	const QF &q, float stiffness, V3 &c0, V3 &c1, V3 &c2, V3 &c3)
{
	const V3 x[4] = { p2, p3, p0, p1 };
	const float w[4] = { w2, w3, w0, w1 };
	float energy; V3 g[4];
	isometric_energy_grad(x, q, energy, g);
	float sum = 0.0f;
	for (int j = 0; j < 4; j++)
		if (w[j] != 0.0f)
			sum += w[j] * sqn(g[j]);
	if (fabsf(sum) > PBDX_EPS)
	{
		const float s = energy / sum;
		c0 = (-stiffness * (s * w[2])) * g[2];
		c1 = (-stiffness * (s * w[3])) * g[3];
		c2 = (-stiffness * (s * w[0])) * g[0];
		c3 = (-stiffness * (s * w[1])) * g[1];
		return true;
	}
	return false;
}

template <typename QF>
PBDX_HD bool solve_isometric_bending_xpbd(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	const QF &q, float stiffness, float dt, float &lambda, V3 &c0, V3 &c1, V3 &c2, V3 &c3)
{
	const V3 x[4] = { p2, p3, p0, p1 };
	const float w[4] = { w2, w3, w0, w1 };
	float energy; V3 g[4];
	isometric_energy_grad(x, q, energy, g);
	float sum = 0.0f;
	for (int j = 0; j < 4; j++)
		if (w[j] != 0.0f)
			sum += w[j] * sqn(g[j]);
	float alpha = 0.0f;
	if (stiffness != 0.0f)
	{
		alpha = 1.0f / (stiffness * dt * dt);
		sum += alpha;
	}
	if (fabsf(sum) > PBDX_EPS)
	{
		const float delta_lambda = -(energy + alpha * lambda) / sum;
		lambda += delta_lambda;
		c0 = (delta_lambda * w[2]) * g[2];
		c1 = (delta_lambda * w[3]) * g[3];
		c2 = (delta_lambda * w[0]) * g[0];
		c3 = (delta_lambda * w[1]) * g[1];
		return true;
	}
	return false;
}

// FEMTriangleConstraint         PositionBasedDynamics.cpp:844-930
// im = invRestMat (2x2) as im[r][c]
PBDX_HD bool solve_fem_triangle(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2,
	float area, const float im[2][2], float Ex, float Ey, float Es, float nuXY, float nuYX,
	V3 &c0, V3 &c1, V3 &c2)
{
	float Cm[3][3] = { { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f } };
	Cm[0][0] = Ex / (1.0f - nuXY * nuYX);
	Cm[0][1] = Ex * nuYX / (1.0f - nuXY * nuYX);
	Cm[1][1] = Ey / (1.0f - nuXY * nuYX);
	Cm[1][0] = Ey * nuXY / (1.0f - nuXY * nuYX);
	Cm[2][2] = Es;

	const V3 p13 = p0 - p2;
	const V3 p23 = p1 - p2;
	float F[3][2];
	F[0][0] = p13.x * im[0][0] + p23.x * im[1][0];
	F[0][1] = p13.x * im[0][1] + p23.x * im[1][1];
	F[1][0] = p13.y * im[0][0] + p23.y * im[1][0];
	F[1][1] = p13.y * im[0][1] + p23.y * im[1][1];
	F[2][0] = p13.z * im[0][0] + p23.z * im[1][0];
	F[2][1] = p13.z * im[0][1] + p23.z * im[1][1];

	float eps[2][2];
	eps[0][0] = 0.5f * (F[0][0] * F[0][0] + F[1][0] * F[1][0] + F[2][0] * F[2][0] - 1.0f);
	eps[1][1] = 0.5f * (F[0][1] * F[0][1] + F[1][1] * F[1][1] + F[2][1] * F[2][1] - 1.0f);
	eps[0][1] = 0.5f * (F[0][0] * F[0][1] + F[1][0] * F[1][1] + F[2][0] * F[2][1]);
	eps[1][0] = eps[0][1];

	float st[2][2];
	st[0][0] = Cm[0][0] * eps[0][0] + Cm[0][1] * eps[1][1] + Cm[0][2] * eps[0][1];
	st[1][1] = Cm[1][0] * eps[0][0] + Cm[1][1] * eps[1][1] + Cm[1][2] * eps[0][1];
	st[0][1] = Cm[2][0] * eps[0][0] + Cm[2][1] * eps[1][1] + Cm[2][2] * eps[0][1];
	st[1][0] = st[0][1];

	// piolaKirchhoffStres = F * stress  (3x2 * 2x2, depth-2 reduction a0 + a1)
	float PK[3][2];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 2; j++)
			PK[i][j] = F[i][0] * st[0][j] + F[i][1] * st[1][j];

	float psi = 0.0f;
	for (int j = 0; j < 2; j++)
		for (int k = 0; k < 2; k++)
			psi += eps[j][k] * st[j][k];
	psi = 0.5f * psi;
	const float energy = area * psi;

	// H = area * PK * invRestMat^T : (area*PK) is a scalar-times-matrix operand of the product,
	// Eigen evaluates nested (s*A)*B coefficient-wise as sum_k (s*A_ik) * B_kj
	float H[3][2];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 2; j++)
			H[i][j] = (area * PK[i][0]) * im[j][0] + (area * PK[i][1]) * im[j][1];

	const V3 g0 = mk(H[0][0], H[1][0], H[2][0]);
	const V3 g1 = mk(H[0][1], H[1][1], H[2][1]);
	const V3 g2 = (-g0) - g1;

	float sum = w0 * sqn(g0);
	sum += w1 * sqn(g1);
	sum += w2 * sqn(g2);
	if (fabsf(sum) > PBDX_EPS)
	{
		const float s = energy / sum;
		c0 = (-(s * w0)) * g0;
		c1 = (-(s * w1)) * g1;
		c2 = (-(s * w2)) * g2;
		return true;
	}
	return false;
}

// StrainTriangleConstraint      PositionBasedDynamics.cpp:584-688
PBDX_HD bool solve_strain_triangle(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2,
	const float im[2][2], float xxStiffness, float yyStiffness, float xyStiffness,
	bool normalizeStretch, bool normalizeShear, V3 &corr0, V3 &corr1, V3 &corr2)
{
	V3 c[2];
	c[0] = mk(im[0][0], im[1][0], 0.0f);
	c[1] = mk(im[0][1], im[1][1], 0.0f);
	V3 r[3];
	corr0 = mk(0.0f, 0.0f, 0.0f); corr1 = corr0; corr2 = corr0;

	for (int i = 0; i < 2; i++)
	{
		for (int j = 0; j <= i; j++)
		{
			r[0] = mk((p1.x + corr1.x) - (p0.x + corr0.x), (p2.x + corr2.x) - (p0.x + corr0.x), 0.0f);
			r[1] = mk((p1.y + corr1.y) - (p0.y + corr0.y), (p2.y + corr2.y) - (p0.y + corr0.y), 0.0f);
			r[2] = mk((p1.z + corr1.z) - (p0.z + corr0.z), (p2.z + corr2.z) - (p0.z + corr0.z), 0.0f);

			float Sij = 0.0f;
			for (int k = 0; k < 3; k++)
				Sij += dot(r[k], c[i]) * dot(r[k], c[j]);

			V3 d[3];
			d[0] = mk(0.0f, 0.0f, 0.0f);
			for (int k = 0; k < 2; k++)
			{
				d[k + 1] = mk(dot(r[0], c[j]), dot(r[1], c[j]), dot(r[2], c[j])) * im[k][i];
				d[k + 1] = d[k + 1] + mk(dot(r[0], c[i]), dot(r[1], c[i]), dot(r[2], c[i])) * im[k][j];
				d[0] = d[0] - d[k + 1];
			}

			if (i != j && normalizeShear)
			{
				float fi2 = 0.0f, fj2 = 0.0f;
				for (int k = 0; k < 3; k++)
				{
					fi2 += dot(r[k], c[i]) * dot(r[k], c[i]);
					fj2 += dot(r[k], c[j]) * dot(r[k], c[j]);
				}
				const float fi = sqrtf(fi2);
				const float fj = sqrtf(fj2);
				d[0] = mk(0.0f, 0.0f, 0.0f);
				const float s = Sij / (fi2 * fi * fj2 * fj);
				for (int k = 0; k < 2; k++)
				{
					d[k + 1] = d[k + 1] / (fi * fj);
					d[k + 1] = d[k + 1] - ((fj * fj) * mk(dot(r[0], c[i]), dot(r[1], c[i]), dot(r[2], c[i]))) * im[k][i] * s;
					d[k + 1] = d[k + 1] - ((fi * fi) * mk(dot(r[0], c[j]), dot(r[1], c[j]), dot(r[2], c[j]))) * im[k][j] * s;
					d[0] = d[0] - d[k + 1];
				}
				Sij = Sij / (fi * fj);
			}

			float lambda = w0 * sqn(d[0]) + w1 * sqn(d[1]) + w2 * sqn(d[2]);
			if (lambda == 0.0f)
				continue;

			if (i == 0 && j == 0)
			{
				if (normalizeStretch) { const float s = sqrtf(Sij); lambda = 2.0f * s * (s - 1.0f) / lambda * xxStiffness; }
				else lambda = (Sij - 1.0f) / lambda * xxStiffness;
			}
			else if (i == 1 && j == 1)
			{
				if (normalizeStretch) { const float s = sqrtf(Sij); lambda = 2.0f * s * (s - 1.0f) / lambda * yyStiffness; }
				else lambda = (Sij - 1.0f) / lambda * yyStiffness;
			}
			else
				lambda = Sij / lambda * xyStiffness;

			corr0 = corr0 - (lambda * w0) * d[0];
			corr1 = corr1 - (lambda * w1) * d[1];
			corr2 = corr2 - (lambda * w2) * d[2];
		}
	}
	return true;
}

// ---------------------------------------------------------------------------
// 3x3 Jacobi eigen decomposition + SVD with inversion handling
// MathFunctions.cpp:11-75, 261-388 (only reached by crushed / inverted tets)
// The two double expressions of a Jacobi rotation, narrowed to float (see jacobi_rotate).  On the device the correctly rounded double square root and
// division are what a crushed tet's colour step WAITS for: ten sequential rotations, each a chain of two sqrt and two divisions of ~15 and ~11 dependent
// double instructions (the 100 k-tet bar's late state: 2.07 ms per substep against 0.58, profiles/HISTORY.md [9], [10]).  Only the FLOAT the expression
// is narrowed to matters, so the device takes a short way -- v_rsq_f64 / v_rcp_f64 and two Newton steps, good to a few units of the 52nd bit -- and
// PROVES that the narrowing cannot tell: if the value shrunk and the value grown by 2^-44 round to the same float, every double in between does
// (rounding is monotonic), in particular the one the reference computes with its three correctly rounded operations; otherwise (one case in a million,
// and for non-finite inputs) the expression is evaluated as written.  Bit-identical by construction; the host always evaluates as written.
#ifndef PBDX_FAST_NARROW
#define PBDX_FAST_NARROW 1
#endif
#ifndef PBDX_UNIFORM_JACOBI
#define PBDX_UNIFORM_JACOBI 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && PBDX_FAST_NARROW
__device__ __forceinline__ double pbdx_rsqrt_refined(double x)
{
	double r = __builtin_amdgcn_rsq(x);
	const double h = 0.5 * x;
#pragma unroll
	for (int k = 0; k < 2; k++) { const double e = __builtin_fma(-(h * r), r, 0.5); r = __builtin_fma(r, e, r); }
	return r;
}
__device__ __forceinline__ bool pbdx_narrow_is_certain(double v, float &out)
{
	const float lo = (float)(v * (1.0 - 0x1p-44)), hi = (float)(v * (1.0 + 0x1p-44));
	out = lo;
	return lo == hi;      // (false for NaN)
}
// (float)(1.0 / (fabs((double)d) + sqrt((double)x)))
__device__ __forceinline__ float narrowed_recip_abs_plus_sqrt(float d, float x)
{
	const double X = (double)x;
	const double sq = X * pbdx_rsqrt_refined(X);
	const double den = fabs((double)d) + sq;
	double r = __builtin_amdgcn_rcp(den);
#pragma unroll
	for (int k = 0; k < 2; k++) { const double e = __builtin_fma(-den, r, 1.0); r = __builtin_fma(r, e, r); }
	float out;
	if (__builtin_expect(pbdx_narrow_is_certain(r, out), 1)) return out;
	return (float)(1.0 / (fabs((double)d) + sqrt(X)));
}
// (float)(1.0 / sqrt((double)y))
__device__ __forceinline__ float narrowed_rsqrt(float y)
{
	const double Y = (double)y;
	float out;
	if (__builtin_expect(pbdx_narrow_is_certain(pbdx_rsqrt_refined(Y), out), 1)) return out;
	return (float)(1.0 / sqrt(Y));
}
#else
PBDX_HD float narrowed_recip_abs_plus_sqrt(float d, float x) { return (float)(1.0 / (fabs((double)d) + sqrt((double)x))); }
PBDX_HD float narrowed_rsqrt(float y) { return (float)(1.0 / sqrt((double)y)); }
#endif

PBDX_HD void jacobi_rotate(M3 &A, M3 &R, int p, int q)
{
	if (A.m[p][q] == 0.0f)
		return;
	const float d = (A.m[p][p] - A.m[q][q]) / (2.0f * A.m[p][q]);
	// ::fabs / ::sqrt are the double functions in MathFunctions.cpp (:18,:20): the sum and the quotient
	// are double expressions narrowed to Real on assignment
	float t = narrowed_recip_abs_plus_sqrt(d, d * d + 1.0f);
	if (d < 0.0f) t = -t;
	const float c = narrowed_rsqrt(t * t + 1.0f);
	const float s = t * c;
	A.m[p][p] += t * A.m[p][q];
	A.m[q][q] -= t * A.m[p][q];
	A.m[p][q] = A.m[q][p] = 0.0f;
	for (int k = 0; k < 3; k++)
	{
		if (k != p && k != q)
		{
			const float Akp = c * A.m[k][p] + s * A.m[k][q];
			const float Akq = -s * A.m[k][p] + c * A.m[k][q];
			A.m[k][p] = A.m[p][k] = Akp;
			A.m[k][q] = A.m[q][k] = Akq;
		}
	}
	for (int k = 0; k < 3; k++)
	{
		const float Rkp = c * R.m[k][p] + s * R.m[k][q];
		const float Rkq = -s * R.m[k][p] + c * R.m[k][q];
		R.m[k][p] = Rkp;
		R.m[k][q] = Rkq;
	}
}

// Device form of eigen_decomposition: ONE rotation body for all three pivots.  The lanes of a wave choose their pivots independently, so with the three
// statically dispatched bodies below a wave in which the pivots differ executes up to three rotations per iteration, one after the other -- and a
// rotation is a long dependent chain (a float division, two narrowed double expressions).  Here the pivot only SELECTS which of the six entries of the
// symmetric matrix and which two columns of the vectors take part (v_cndmask); the arithmetic is jacobi_rotate's, operation for operation
// (MathFunctions.cpp:11-43: the symmetric twin of every entry is written with the same value there, so six entries carry the matrix).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void eigen_decomposition_uniform(const M3 &A, M3 &vecs, float vals[3])
{
	const float epsilon = 1e-15f;
	float d0 = A.m[0][0], d1 = A.m[1][1], d2 = A.m[2][2], o01 = A.m[0][1], o02 = A.m[0][2], o12 = A.m[1][2];
	// (the reference reads D(0,1), D(0,2), D(1,2) for the pivot search and A(k,p) / A(k,q) with k the remaining index in the rotation: for a matrix whose
	// twins are equal -- A^T A as mul() builds it is not bitwise symmetric in general, so the LOWER entries the rotation reads are carried as well)
	float l10 = A.m[1][0], l20 = A.m[2][0], l21 = A.m[2][1];
	float r00 = 1.0f, r01 = 0.0f, r02 = 0.0f, r10 = 0.0f, r11 = 1.0f, r12 = 0.0f, r20 = 0.0f, r21 = 0.0f, r22 = 1.0f;
	for (int iter = 0; iter < 10; iter++)
	{
		int pv = 0;                                   // 0: (p, q) = (0, 1), k = 2;  1: (0, 2), k = 1;  2: (1, 2), k = 0
		float mx = fabsf(o01);
		float a = fabsf(o02);
		if (a > mx) { pv = 1; mx = a; }
		a = fabsf(o12);
		if (a > mx) { pv = 2; mx = a; }
		if (mx < epsilon) break;
		const float Apq = pv == 0 ? o01 : pv == 1 ? o02 : o12;
		if (Apq == 0.0f) continue;                    // (jacobi_rotate returns; the iteration still counts)
		const float App = pv == 2 ? d1 : d0, Aqq = pv == 0 ? d1 : d2;
		// A(k, p), A(k, q): row k of the full matrix
		const float Akp = pv == 0 ? l20 : pv == 1 ? l10 : o01;
		const float Akq = pv == 0 ? l21 : pv == 1 ? o12 : o02;
		const float d = (App - Aqq) / (2.0f * Apq);
		float t = narrowed_recip_abs_plus_sqrt(d, d * d + 1.0f);
		if (d < 0.0f) t = -t;
		const float c = narrowed_rsqrt(t * t + 1.0f);
		const float s = t * c;
		const float nApp = App + t * Apq, nAqq = Aqq - t * Apq;
		const float nAkp = c * Akp + s * Akq;
		const float nAkq = -s * Akp + c * Akq;
		// write back: A(p,p), A(q,q), A(p,q) = A(q,p) = 0, A(k,p) = A(p,k), A(k,q) = A(q,k)
		if (pv == 0) { d0 = nApp; d1 = nAqq; o01 = 0.0f; l10 = 0.0f; l20 = nAkp; o02 = nAkp; l21 = nAkq; o12 = nAkq; }
		else if (pv == 1) { d0 = nApp; d2 = nAqq; o02 = 0.0f; l20 = 0.0f; l10 = nAkp; o01 = nAkp; o12 = nAkq; l21 = nAkq; }
		else { d1 = nApp; d2 = nAqq; o12 = 0.0f; l21 = 0.0f; o01 = nAkp; l10 = nAkp; o02 = nAkq; l20 = nAkq; }
		// columns p and q of the vectors
		const float p0 = pv == 2 ? r01 : r00, p1 = pv == 2 ? r11 : r10, p2 = pv == 2 ? r21 : r20;
		const float q0 = pv == 0 ? r01 : r02, q1 = pv == 0 ? r11 : r12, q2 = pv == 0 ? r21 : r22;
		const float np0 = c * p0 + s * q0, nq0 = -s * p0 + c * q0;
		const float np1 = c * p1 + s * q1, nq1 = -s * p1 + c * q1;
		const float np2 = c * p2 + s * q2, nq2 = -s * p2 + c * q2;
		if (pv == 2) { r01 = np0; r11 = np1; r21 = np2; } else { r00 = np0; r10 = np1; r20 = np2; }
		if (pv == 0) { r01 = nq0; r11 = nq1; r21 = nq2; } else { r02 = nq0; r12 = nq1; r22 = nq2; }
	}
	vecs.m[0][0] = r00; vecs.m[0][1] = r01; vecs.m[0][2] = r02; vecs.m[1][0] = r10; vecs.m[1][1] = r11; vecs.m[1][2] = r12; vecs.m[2][0] = r20; vecs.m[2][1] = r21; vecs.m[2][2] = r22;
	vals[0] = d0; vals[1] = d1; vals[2] = d2;
}
#endif

PBDX_HD void eigen_decomposition(const M3 &A, M3 &vecs, float vals[3])
{
#if defined(__HIP_DEVICE_COMPILE__) && PBDX_UNIFORM_JACOBI
	eigen_decomposition_uniform(A, vecs, vals);
	return;
#endif
	const float epsilon = 1e-15f;
	M3 D = A;
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) vecs.m[i][j] = (i == j) ? 1.0f : 0.0f;
	int iter = 0;
	while (iter < 10)
	{
		int p = 0, q = 1;
		float mx = fabsf(D.m[0][1]);
		float a = fabsf(D.m[0][2]);
		if (a > mx) { p = 0; q = 2; mx = a; }
		a = fabsf(D.m[1][2]);
		if (a > mx) { p = 1; q = 2; mx = a; }
		if (mx < epsilon) break;
		// static dispatch keeps the matrices in registers on the GPU
		if (p == 0 && q == 1) jacobi_rotate(D, vecs, 0, 1);
		else if (p == 0 && q == 2) jacobi_rotate(D, vecs, 0, 2);
		else jacobi_rotate(D, vecs, 1, 2);
		iter++;
	}
	vals[0] = D.m[0][0]; vals[1] = D.m[1][1]; vals[2] = D.m[2][2];
}

PBDX_HD void set_col(M3 &A, int c, V3 v) { A.m[0][c] = v.x; A.m[1][c] = v.y; A.m[2][c] = v.z; }
PBDX_HD V3 get_col(const M3 &A, int c) { return mk(A.m[0][c], A.m[1][c], A.m[2][c]); }

PBDX_HD void svd_with_inversion_handling(const M3 &A, float sigma[3], M3 &U, M3 &VT)
{
	M3 V;
	const M3 AT_A = mul(transpose(A), A);
	float S[3];
	eigen_decomposition(AT_A, V, S);

	const float detV = det(V);
	if (detV < 0.0f)
	{
		float minLambda = 3.402823466e+38f;
		int pos = 0;
		for (int l = 0; l < 3; l++)
			if (S[l] < minLambda) { pos = l; minLambda = S[l]; }
		for (int l = 0; l < 3; l++)
			if (l == pos) { V.m[0][l] = -V.m[0][l]; V.m[1][l] = -V.m[1][l]; V.m[2][l] = -V.m[2][l]; }
	}
	if (S[0] < 0.0f) S[0] = 0.0f;
	if (S[1] < 0.0f) S[1] = 0.0f;
	if (S[2] < 0.0f) S[2] = 0.0f;
	sigma[0] = sqrtf(S[0]); sigma[1] = sqrtf(S[1]); sigma[2] = sqrtf(S[2]);
	VT = transpose(V);

	int chk = 0, pos = 0;
	for (int l = 0; l < 3; l++)
		if ((double)fabsf(sigma[l]) < 1.0e-4) { pos = l; chk++; }

	if (chk > 0)
	{
		if (chk > 1)
		{
			for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U.m[i][j] = (i == j) ? 1.0f : 0.0f;
		}
		else
		{
			U = mul(A, V);
			for (int l = 0; l < 3; l++)
				if (l != pos)
					for (int m = 0; m < 3; m++)
						U.m[m][l] *= 1.0f / sigma[l];
			V3 v[2];
			int index = 0;
			for (int l = 0; l < 3; l++)
				if (l != pos)
				{
					const V3 col = get_col(U, l);
					if (index == 0) v[0] = col; else v[1] = col;
					index++;
				}
			const V3 vec = normalized(cross(v[0], v[1]));
			for (int l = 0; l < 3; l++)
				if (l == pos) set_col(U, l, vec);
		}
	}
	else
	{
		const float sInv[3] = { 1.0f / sigma[0], 1.0f / sigma[1], 1.0f / sigma[2] };
		U = mul(A, V);
		for (int l = 0; l < 3; l++)
			for (int m = 0; m < 3; m++)
				U.m[m][l] *= sInv[l];
	}

	const float detU = det(U);
	if (detU < 0.0f)
	{
		float minLambda = 3.402823466e+38f;
		int p2 = 0;
		for (int l = 0; l < 3; l++)
			if (sigma[l] < minLambda) { p2 = l; minLambda = sigma[l]; }
		for (int l = 0; l < 3; l++)
			if (l == p2)
			{
				sigma[l] = -sigma[l];
				U.m[0][l] = -U.m[0][l]; U.m[1][l] = -U.m[1][l]; U.m[2][l] = -U.m[2][l];
			}
	}
}

// F = Ds * Dm^-1 written out element-wise  PositionBasedDynamics.cpp:965-979
PBDX_HD M3 deformation_gradient(V3 x1, V3 x2, V3 x3, V3 x4, const M3 &im)
{
	const V3 p14 = x1 - x4, p24 = x2 - x4, p34 = x3 - x4;
	M3 F;
	for (int c = 0; c < 3; c++)
	{
		F.m[0][c] = p14.x * im.m[0][c] + p24.x * im.m[1][c] + p34.x * im.m[2][c];
		F.m[1][c] = p14.y * im.m[0][c] + p24.y * im.m[1][c] + p34.y * im.m[2][c];
		F.m[2][c] = p14.z * im.m[0][c] + p24.z * im.m[1][c] + p34.z * im.m[2][c];
	}
	return F;
}

// computeGreenStrainAndPiolaStress   PositionBasedDynamics.cpp:958-1008
PBDX_HD void green_strain_piola_stress(V3 x1, V3 x2, V3 x3, V3 x4, const M3 &im, float restVolume,
	float mu, float lambda, M3 &sigma, float &energy)
{
	const M3 F = deformation_gradient(x1, x2, x3, x4, im);
	M3 e;
	e.m[0][0] = 0.5f * (F.m[0][0] * F.m[0][0] + F.m[1][0] * F.m[1][0] + F.m[2][0] * F.m[2][0] - 1.0f);
	e.m[1][1] = 0.5f * (F.m[0][1] * F.m[0][1] + F.m[1][1] * F.m[1][1] + F.m[2][1] * F.m[2][1] - 1.0f);
	e.m[2][2] = 0.5f * (F.m[0][2] * F.m[0][2] + F.m[1][2] * F.m[1][2] + F.m[2][2] * F.m[2][2] - 1.0f);
	e.m[0][1] = 0.5f * (F.m[0][0] * F.m[0][1] + F.m[1][0] * F.m[1][1] + F.m[2][0] * F.m[2][1]);
	e.m[0][2] = 0.5f * (F.m[0][0] * F.m[0][2] + F.m[1][0] * F.m[1][2] + F.m[2][0] * F.m[2][2]);
	e.m[1][2] = 0.5f * (F.m[0][1] * F.m[0][2] + F.m[1][1] * F.m[1][2] + F.m[2][1] * F.m[2][2]);
	e.m[1][0] = e.m[0][1]; e.m[2][0] = e.m[0][2]; e.m[2][1] = e.m[1][2];

	const float trace = e.m[0][0] + e.m[1][1] + e.m[2][2];
	const float ltrace = lambda * trace;
	M3 s;
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) s.m[i][j] = e.m[i][j] * 2.0f * mu;
	s.m[0][0] += ltrace; s.m[1][1] += ltrace; s.m[2][2] += ltrace;
	sigma = mul(F, s);

	float psi = 0.0f;
	for (int j = 0; j < 3; j++)
		for (int k = 0; k < 3; k++)
			psi += e.m[j][k] * e.m[j][k];
	psi = mu * psi + 0.5f * lambda * trace * trace;
	energy = restVolume * psi;
}

// computeGreenStrainAndPiolaStressInversion   PositionBasedDynamics.cpp:1034-1104
PBDX_HD void green_strain_piola_stress_inversion(V3 x1, V3 x2, V3 x3, V3 x4, const M3 &im, float restVolume,
	float mu, float lambda, M3 &sigma, float &energy)
{
	const M3 F = deformation_gradient(x1, x2, x3, x4, im);
	M3 U, VT;
	float hatF[3];
	svd_with_inversion_handling(F, hatF, U, VT);
	const float minXVal = 0.577f;
	for (int j = 0; j < 3; j++)
		if (hatF[j] < minXVal) hatF[j] = minXVal;
	const float eH[3] = { 0.5f * (hatF[0] * hatF[0] - 1.0f), 0.5f * (hatF[1] * hatF[1] - 1.0f), 0.5f * (hatF[2] * hatF[2] - 1.0f) };
	// Vector3r trace: no Eigen reduction here, plain left-to-right sum
	const float trace = eH[0] + eH[1] + eH[2];
	const float ltrace = lambda * trace;
	float sv[3] = { eH[0] * 2.0f * mu, eH[1] * 2.0f * mu, eH[2] * 2.0f * mu };
	sv[0] += ltrace; sv[1] += ltrace; sv[2] += ltrace;
	sv[0] = hatF[0] * sv[0]; sv[1] = hatF[1] * sv[1]; sv[2] = hatF[2] * sv[2];

	M3 sD, eD;
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { sD.m[i][j] = (i == j) ? sv[i] : 0.0f; eD.m[i][j] = (i == j) ? eH[i] : 0.0f; }
	const M3 epsilon = mul(mul(U, eD), VT);
	sigma = mul(mul(U, sD), VT);

	float psi = 0.0f;
	for (int j = 0; j < 3; j++)
		for (int k = 0; k < 3; k++)
			psi += epsilon.m[j][k] * epsilon.m[j][k];
	psi = mu * psi + 0.5f * lambda * trace * trace;
	energy = restVolume * psi;
}

// computeGradCGreen             PositionBasedDynamics.cpp:1011-1031
PBDX_HD void grad_c_green(float restVolume, const M3 &im, const M3 &sigma, V3 J[4])
{
	const M3 T = transpose(im);
	M3 H = mul(sigma, T);
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) H.m[i][j] = H.m[i][j] * restVolume;
	J[0] = mk(H.m[0][0], H.m[1][0], H.m[2][0]);
	J[1] = mk(H.m[0][1], H.m[1][1], H.m[2][1]);
	J[2] = mk(H.m[0][2], H.m[1][2], H.m[2][2]);
	J[3] = (-J[0]) - J[1] - J[2];
}

// wrapper part shared by FEMTetConstraint / XPBD_FEMTetConstraint  Constraints.cpp:1795-1798
PBDX_HD bool fem_tet_handle_inversion(V3 x1, V3 x2, V3 x3, V3 x4, float restVolume)
{
	const float currentVolume = -(float)(1.0 / 6.0) * dot(x4 - x1, cross(x3 - x1, x2 - x1));
	return (double)(currentVolume / restVolume) < 0.2;
}

// FEMTetConstraint              PositionBasedDynamics.cpp:1109-1169
PBDX_HD bool solve_fem_tet(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	float restVolume, const M3 &im, float youngsModulus, float poissonRatio, bool handleInversion,
	V3 &c0, V3 &c1, V3 &c2, V3 &c3)
{
	c0 = mk(0.0f, 0.0f, 0.0f); c1 = c0; c2 = c0; c3 = c0;
	if (youngsModulus <= 0.0f)
		return true;
	if (poissonRatio < 0.0f || (double)poissonRatio > 0.49)
		return false;
	float C = 0.0f;
	V3 g[4];
	M3 sigma;
	const float volume = dot(cross(p1 - p0, p2 - p0), p3 - p0) / 6.0f;
	const float mu = youngsModulus / 2.0f / (1.0f + poissonRatio);
	const float lambda = youngsModulus * poissonRatio / (1.0f + poissonRatio) / (1.0f - 2.0f * poissonRatio);
	if (!handleInversion || volume > 0.0f)
		green_strain_piola_stress(p0, p1, p2, p3, im, restVolume, mu, lambda, sigma, C);
	else
		green_strain_piola_stress_inversion(p0, p1, p2, p3, im, restVolume, mu, lambda, sigma, C);
	grad_c_green(restVolume, im, sigma, g);
	const float sum = w0 * sqn(g[0]) + w1 * sqn(g[1]) + w2 * sqn(g[2]) + w3 * sqn(g[3]);
	if (sum < PBDX_EPS)
		return false;
	const float s = C / sum;
	c0 = (-s * w0) * g[0];
	c1 = (-s * w1) * g[1];
	c2 = (-s * w2) * g[2];
	c3 = (-s * w3) * g[3];
	return true;
}

// XPBD_FEMTetConstraint         XPBD.cpp:217-294
PBDX_HD bool solve_fem_tet_xpbd(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	float restVolume, const M3 &im, float youngsModulus, float poissonRatio, bool handleInversion,
	float dt, float &multiplier, V3 &c0, V3 &c1, V3 &c2, V3 &c3)
{
	c0 = mk(0.0f, 0.0f, 0.0f); c1 = c0; c2 = c0; c3 = c0;
	if (youngsModulus <= 0.0f)
		return true;
	if (poissonRatio < 0.0f || (double)poissonRatio > 0.49)
		return false;
	V3 g[4];
	M3 sigma;
	const float volume = dot(cross(p1 - p0, p2 - p0), p3 - p0) / 6.0f;
	// `1.0 / Real(2)` etc. are double expressions in the reference (1.0 is a double literal),
	// rounded to Real on assignment
	const float mu_ = (float)(1.0 / (double)2.0f / (double)(1.0f + poissonRatio));
	const float lambda_ = (float)(1.0 * (double)poissonRatio / (double)(1.0f + poissonRatio) / (double)(1.0f - 2.0f * poissonRatio));
	float U_ = 0.0f;
	if (!handleInversion || volume > 0.0f)
		green_strain_piola_stress(p0, p1, p2, p3, im, restVolume, mu_, lambda_, sigma, U_);
	else
		green_strain_piola_stress_inversion(p0, p1, p2, p3, im, restVolume, mu_, lambda_, sigma, U_);
	grad_c_green(restVolume, im, sigma, g);
	// sqrt(2.0 * U_) is evaluated in double and rounded to Real
	const float C = (float)sqrt(2.0 * (double)U_);
	float sum = w0 * sqn(g[0]) + w1 * sqn(g[1]) + w2 * sqn(g[2]) + w3 * sqn(g[3]);
	const float alpha = 1.0f / (youngsModulus * dt * dt);
	sum += C * C * alpha;
	if (sum < PBDX_EPS)
		return false;
	const float lambda = -C * (C + alpha * multiplier) / sum;
	multiplier += lambda;
	c0 = (lambda * w0) * g[0];
	c1 = (lambda * w1) * g[1];
	c2 = (lambda * w2) * g[2];
	c3 = (lambda * w3) * g[3];
	return true;
}

// StrainTetConstraint           PositionBasedDynamics.cpp:713-805
// (wrapper passes stretch/shear stiffness replicated to all three components)
PBDX_HD bool solve_strain_tet(V3 p0, float w0, V3 p1, float w1, V3 p2, float w2, V3 p3, float w3,
	const M3 &im, float stretchStiffness, float shearStiffness, bool normalizeStretch, bool normalizeShear,
	V3 &corr0, V3 &corr1, V3 &corr2, V3 &corr3)
{
	corr0 = mk(0.0f, 0.0f, 0.0f); corr1 = corr0; corr2 = corr0; corr3 = corr0;
	V3 c[3];
	c[0] = get_col(im, 0); c[1] = get_col(im, 1); c[2] = get_col(im, 2);
	// m_stretchStiffness * Vector3r::Ones(): 1.0f * k == k exactly
	for (int i = 0; i < 3; i++)
	{
		for (int j = 0; j <= i; j++)
		{
			M3 P;
			set_col(P, 0, (p1 + corr1) - (p0 + corr0));
			set_col(P, 1, (p2 + corr2) - (p0 + corr0));
			set_col(P, 2, (p3 + corr3) - (p0 + corr0));
			const V3 fi = mul(P, c[i]);
			const V3 fj = mul(P, c[j]);
			float Sij = dot(fi, fj);
			float wi = 0.0f, wj = 0.0f, s1 = 0.0f, s3 = 0.0f;
			const bool ns = normalizeShear && i != j;
			if (ns)
			{
				wi = norm(fi);
				wj = norm(fj);
				s1 = 1.0f / (wi * wj);
				s3 = s1 * s1 * s1;
			}
			V3 d[4];
			d[0] = mk(0.0f, 0.0f, 0.0f);
			for (int k = 0; k < 3; k++)
			{
				d[k + 1] = fj * im.m[k][i] + fi * im.m[k][j];
				if (ns)
					d[k + 1] = s1 * d[k + 1] - (Sij * s3) * (((wj * wj) * fi) * im.m[k][i] + ((wi * wi) * fj) * im.m[k][j]);
				d[0] = d[0] - d[k + 1];
			}
			if (ns)
				Sij *= s1;
			float lambda = w0 * sqn(d[0]) + w1 * sqn(d[1]) + w2 * sqn(d[2]) + w3 * sqn(d[3]);
			if (fabsf(lambda) < PBDX_EPS)
				continue;
			if (i == j)
			{
				if (normalizeStretch) { const float s = sqrtf(Sij); lambda = 2.0f * s * (s - 1.0f) / lambda * stretchStiffness; }
				else lambda = (Sij - 1.0f) / lambda * stretchStiffness;
			}
			else
				lambda = Sij / lambda * shearStiffness;
			corr0 = corr0 - (lambda * w0) * d[0];
			corr1 = corr1 - (lambda * w1) * d[1];
			corr2 = corr2 - (lambda * w2) * d[2];
			corr3 = corr3 - (lambda * w3) * d[3];
		}
	}
	return true;
}

// ---------------------------------------------------------------------------
// ShapeMatchingConstraint (4-particle clusters)
// PositionBasedDynamics.cpp:501-558, MathFunctions.cpp:147-254
PBDX_HD float one_norm(const M3 &A)
{
	// ::fabs(double): every sum is accumulated in double and narrowed (MathFunctions.cpp:149-151)
	const float s1 = (float)(fabs((double)A.m[0][0]) + fabs((double)A.m[1][0]) + fabs((double)A.m[2][0]));
	const float s2 = (float)(fabs((double)A.m[0][1]) + fabs((double)A.m[1][1]) + fabs((double)A.m[2][1]));
	const float s3 = (float)(fabs((double)A.m[0][2]) + fabs((double)A.m[1][2]) + fabs((double)A.m[2][2]));
	float mx = s1;
	if (s2 > mx) mx = s2;
	if (s3 > mx) mx = s3;
	return mx;
}
PBDX_HD float inf_norm(const M3 &A)
{
	const float s1 = (float)(fabs((double)A.m[0][0]) + fabs((double)A.m[0][1]) + fabs((double)A.m[0][2]));
	const float s2 = (float)(fabs((double)A.m[1][0]) + fabs((double)A.m[1][1]) + fabs((double)A.m[1][2]));
	const float s3 = (float)(fabs((double)A.m[2][0]) + fabs((double)A.m[2][1]) + fabs((double)A.m[2][2]));
	float mx = s1;
	if (s2 > mx) mx = s2;
	if (s3 > mx) mx = s3;
	return mx;
}
PBDX_HD V3 get_row(const M3 &A, int r) { return mk(A.m[r][0], A.m[r][1], A.m[r][2]); }
PBDX_HD void set_row(M3 &A, int r, V3 v) { A.m[r][0] = v.x; A.m[r][1] = v.y; A.m[r][2] = v.z; }

PBDX_HD void polar_decomposition_stable(const M3 &M, float tolerance, M3 &R)
{
	M3 Mt = transpose(M);
	float Mone = one_norm(M);
	float Minf = inf_norm(M);
	float Eone;
	M3 MadjTt, Et;
	int guard = 0;
	do
	{
		set_row(MadjTt, 0, cross(get_row(Mt, 1), get_row(Mt, 2)));
		set_row(MadjTt, 1, cross(get_row(Mt, 2), get_row(Mt, 0)));
		set_row(MadjTt, 2, cross(get_row(Mt, 0), get_row(Mt, 1)));
		float dt = Mt.m[0][0] * MadjTt.m[0][0] + Mt.m[0][1] * MadjTt.m[0][1] + Mt.m[0][2] * MadjTt.m[0][2];
		if ((double)fabsf(dt) < 1.0e-12)
		{
			int index = -1;
			for (int i = 0; i < 3; i++)
			{
				const float len = sqn(get_row(MadjTt, i));
				if ((double)len > 1.0e-12) { index = i; break; }
			}
			if (index < 0)
			{
				for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = (i == j) ? 1.0f : 0.0f;
				return;
			}
			const int i1 = (index + 1) % 3, i2 = (index + 2) % 3;
			set_row(Mt, index, cross(get_row(Mt, i1), get_row(Mt, i2)));
			set_row(MadjTt, i1, cross(get_row(Mt, i2), get_row(Mt, index)));
			set_row(MadjTt, i2, cross(get_row(Mt, index), get_row(Mt, i1)));
			const M3 M2 = transpose(Mt);
			Mone = one_norm(M2);
			Minf = inf_norm(M2);
			dt = Mt.m[0][0] * MadjTt.m[0][0] + Mt.m[0][1] * MadjTt.m[0][1] + Mt.m[0][2] * MadjTt.m[0][2];
		}
		const float MadjTone = one_norm(MadjTt);
		const float MadjTinf = inf_norm(MadjTt);
		const float gamma = (float)sqrt(sqrt((double)((MadjTone * MadjTinf) / (Mone * Minf))) / fabs((double)dt));   // double ::sqrt / ::fabs, MathFunctions.cpp:235
		const float g1 = gamma * 0.5f;
		const float g2 = 0.5f / (gamma * dt);
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++)
			{
				Et.m[i][j] = Mt.m[i][j];
				Mt.m[i][j] = g1 * Mt.m[i][j] + g2 * MadjTt.m[i][j];
				Et.m[i][j] -= Mt.m[i][j];
			}
		Eone = one_norm(Et);
		Mone = one_norm(Mt);
		Minf = inf_norm(Mt);
	} while (Eone > Mone * tolerance && ++guard < 100);   // guard: a NaN-free input converges in < 10 sweeps
	R = transpose(Mt);
}

// x0, w (inverse masses captured at init), numClusters: per-constraint constants.
PBDX_HD bool solve_shape_matching4(const V3 x0[4], const V3 x[4], const float w[4], V3 restCm, float stiffness, V3 corr[4])
{
	for (int i = 0; i < 4; i++) corr[i] = mk(0.0f, 0.0f, 0.0f);
	V3 cm = mk(0.0f, 0.0f, 0.0f);
	float wsum = 0.0f;
	for (int i = 0; i < 4; i++)
	{
		const float wi = 1.0f / (w[i] + PBDX_EPS);
		cm = cm + x[i] * wi;
		wsum += wi;
	}
	if (wsum == 0.0f)
		return false;
	cm = cm / wsum;
	M3 mat;
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mat.m[i][j] = 0.0f;
	for (int i = 0; i < 4; i++)
	{
		const V3 q = x0[i] - restCm;
		V3 p = x[i] - cm;
		const float wi = 1.0f / (w[i] + PBDX_EPS);
		p = p * wi;
		mat.m[0][0] += p.x * q.x; mat.m[0][1] += p.x * q.y; mat.m[0][2] += p.x * q.z;
		mat.m[1][0] += p.y * q.x; mat.m[1][1] += p.y * q.y; mat.m[1][2] += p.y * q.z;
		mat.m[2][0] += p.z * q.x; mat.m[2][1] += p.z * q.y; mat.m[2][2] += p.z * q.z;
	}
	M3 R;
	polar_decomposition_stable(mat, PBDX_EPS, R);
	for (int i = 0; i < 4; i++)
	{
		const V3 goal = cm + mul(R, x0[i] - restCm);
		corr[i] = (goal - x[i]) * stiffness;
	}
	return true;
}

} // namespace pbdx
#endif
