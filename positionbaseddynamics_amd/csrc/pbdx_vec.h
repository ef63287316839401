// pbdx_vec.h -- tiny fp32 3-vector / 3x3 helpers shared by the host model
// (constraint initialisation) and the HIP kernels.
//
// Every helper fixes the association order of its floating-point operations to
// the order the reference's Eigen 3.4 expressions evaluate in (fixed-size,
// DontAlign, non-vectorised): a 3-term reduction is  c0 + (c1 + c2)
// (Eigen/src/Core/Redux.h redux_novec_unroller), cross products use the
// textbook component formula, x.normalize() divides by sqrt(squaredNorm()).
// The library is compiled with -ffp-contract=off so no multiply-add is fused and
// results can be compared bit-for-bit with a contraction-free float build of the
// reference.
#ifndef PBDX_VEC_H
#define PBDX_VEC_H

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PBDX_HD __host__ __device__ __forceinline__
#else
#define PBDX_HD inline
#endif

namespace pbdx {

struct V3 { float x, y, z; };

PBDX_HD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PBDX_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PBDX_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PBDX_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PBDX_HD V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
PBDX_HD V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PBDX_HD V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
PBDX_HD float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
PBDX_HD float sqn(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
PBDX_HD float norm(V3 a) { return sqrtf(sqn(a)); }
PBDX_HD V3 cross(V3 a, V3 b)
{
	return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Eigen MatrixBase::normalize(): z = squaredNorm(); if (z > 0) *this /= sqrt(z)
PBDX_HD V3 normalized(V3 a)
{
	const float z = sqn(a);
	if (z > 0.0f) { const float s = sqrtf(z); return mk(a.x / s, a.y / s, a.z / s); }
	return a;
}

// 3x3 matrix, element (r,c) at m[r][c]
struct M3 { float m[3][3]; };

// lazy coefficient product: (A*B)(i,j) = a_i0 b_0j + (a_i1 b_1j + a_i2 b_2j)
PBDX_HD M3 mul(const M3 &A, const M3 &B)
{
	M3 R;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++)
			R.m[i][j] = A.m[i][0] * B.m[0][j] + (A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j]);
	return R;
}
PBDX_HD M3 transpose(const M3 &A)
{
	M3 R;
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = A.m[j][i];
	return R;
}
PBDX_HD V3 mul(const M3 &A, V3 v)
{
	return mk(A.m[0][0] * v.x + (A.m[0][1] * v.y + A.m[0][2] * v.z),
	          A.m[1][0] * v.x + (A.m[1][1] * v.y + A.m[1][2] * v.z),
	          A.m[2][0] * v.x + (A.m[2][1] * v.y + A.m[2][2] * v.z));
}
// Eigen determinant_impl<3>: bruteforce_det3_helper(0,1,2) - (1,0,2) + (2,0,1)
PBDX_HD float det3_helper(const M3 &A, int a, int b, int c)
{
	return A.m[0][a] * (A.m[1][b] * A.m[2][c] - A.m[1][c] * A.m[2][b]);
}
PBDX_HD float det(const M3 &A)
{
	return det3_helper(A, 0, 1, 2) - det3_helper(A, 1, 0, 2) + det3_helper(A, 2, 0, 1);
}
// Eigen compute_inverse<3>: cofactor expansion (LU/InverseImpl.h:125-175)
PBDX_HD float cofactor3(const M3 &A, int i, int j)
{
	const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
	return A.m[i1][j1] * A.m[i2][j2] - A.m[i1][j2] * A.m[i2][j1];
}
PBDX_HD M3 inverse(const M3 &A)
{
	M3 R;
	const float c0 = cofactor3(A, 0, 0), c1 = cofactor3(A, 1, 0), c2 = cofactor3(A, 2, 0);
	const float d = c0 * A.m[0][0] + (c1 * A.m[1][0] + c2 * A.m[2][0]);
	const float invdet = 1.0f / d;
	R.m[1][2] = cofactor3(A, 2, 1) * invdet;
	R.m[2][1] = cofactor3(A, 1, 2) * invdet;
	R.m[2][2] = cofactor3(A, 2, 2) * invdet;
	R.m[1][0] = cofactor3(A, 0, 1) * invdet;
	R.m[1][1] = cofactor3(A, 1, 1) * invdet;
	R.m[2][0] = cofactor3(A, 0, 2) * invdet;
	R.m[0][0] = c0 * invdet; R.m[0][1] = c1 * invdet; R.m[0][2] = c2 * invdet;
	return R;
}

} // namespace pbdx
#endif
