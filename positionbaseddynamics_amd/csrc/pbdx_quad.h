// pbdx_quad.h -- ONE projection spread over the four lanes of a quad (FEMTetConstraint, XPBD_FEMTetConstraint).
//
// A FEM tet projection is a ~390-instruction dependent chain of 3x3 algebra (pbdx_project.h: solve_fem_tet).  On small scenes a colour
// step of a tile holds fewer slots than the workgroup has lanes, so most SIMDs idle while a few lone wavefronts walk that chain at one
// instruction per ~4.5 cycles.  Here the four lanes of a quad share one slot: lane c (0..2) owns COLUMN c of the matrices -- F = Ds Dm^-1,
// the Green strain, the Piola stress, the gradient matrix H = sigma Dm^-T V0 -- and particle c; lane 3 owns particle 3 (whose gradient is
// -J0 - J1 - J2).  Between the stages the columns are exchanged with DPP quad_perm broadcasts (v_mov_b32_dpp, no LDS, no latency beyond a
// VALU instruction); reductions over the columns (trace, the energy's Frobenius sum, the denominator sum w |grad|^2) are replayed on every
// lane in the reference's association order from broadcast terms.  Every floating-point operation is one the scalar code performs, on the
// same operands, in the same order: the result is bit-identical (cross-checked on the GPU against the one-lane-per-constraint kernels of the
// per-colour schedule, which still run the scalar code, and against the reference itself).
// The chain per lane drops to ~250 instructions and a step of n slots occupies 4n lanes: idle SIMDs get wavefronts, two wavefronts per SIMD
// issue every ~2.3 cycles.
//
// Reference arithmetic: PositionBasedDynamics.cpp:958-1031 (computeGreenStrainAndPiolaStress, computeGradCGreen), :1109-1169
// (solve_FEMTetraConstraint), XPBD.cpp:217-294; wrappers Constraints.cpp:1776-1825, 1852-1906.
#ifndef PBDX_QUAD_H
#define PBDX_QUAD_H

#include "pbdx_access.h"

namespace pbdx {

// value of x on lane K of this lane's quad
template <int K> __device__ __forceinline__ float qb(float x)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, false));
}
template <int K> __device__ __forceinline__ V3 qb3(V3 v) { return mk(qb<K>(v.x), qb<K>(v.y), qb<K>(v.z)); }
__device__ __forceinline__ float sel3(uint32_t c, float a0, float a1, float a2) { return c == 0u ? a0 : (c == 1u ? a1 : a2); }

// what a lane of the quad prefetches for its slot: packed indices, multiplier, rest volume, COLUMN c and ROW c of Dm^-1 (c = its column;
// lane 3 doubles lane 0), and -- full layout only -- Young's modulus and Poisson ratio
template <int TYPE, bool COMPACT> struct RecQ
{
	// FEM tets: indices (2), multiplier, rest volume, column and row of Dm^-1 (3 + 3) [+ Young's modulus, Poisson ratio]
	// strain tets: indices (2), unused, Dm^-1 (9) [+ two stiffnesses, two flags]
	uint32_t w[TYPE == PBDX_STRAIN_TET ? (COMPACT ? 12 : 16) : (COMPACT ? 10 : 12)];
};
struct QuadLane
{
	uint32_t q, c;            // lane of the quad, its column
};
__device__ __forceinline__ QuadLane quad_lane()
{
	QuadLane l;
	l.q = threadIdx.x & 3u;
	l.c = l.q == 3u ? 0u : l.q;
	return l;
}

template <int TYPE, bool COMPACT, class A>
__device__ __forceinline__ void load_rec_quad(const A &a, const QuadLane &l, uint32_t slot, RecQ<TYPE, COMPACT> &r)
{
	const uint2 v = a.idx_raw2(slot);
	r.w[0] = v.x; r.w[1] = v.y;
	r.w[2] = 0u;
	if constexpr (TYPE == PBDX_STRAIN_TET)
	{
		// every lane needs all of Dm^-1 (the six sub-projections use every column); parameter k = plane k in both layouts (k < 9)
		a.par_planes(r.w + 3);
		return;
	}
	if constexpr (kHasLambda[TYPE]) r.w[2] = __builtin_bit_cast(uint32_t, a.lam_load(slot));
	// Dm^-1(r, c) is parameter 1 + 3 c + r = plane 1 + 3 c + r of both layouts (the rest volume is plane 0)
	r.w[3] = a.par_raw(a.plane_voff_c(0u));
#pragma unroll
	for (uint32_t k = 0; k < 3; k++) r.w[4 + k] = a.par_raw(a.plane_voff(1u + 3u * l.c + k));
#pragma unroll
	for (uint32_t k = 0; k < 3; k++) r.w[7 + k] = a.par_raw(a.plane_voff(1u + l.c + 3u * k));
	if constexpr (!COMPACT) { r.w[10] = a.par_raw(a.plane_voff_c(10u)); r.w[11] = a.par_raw(a.plane_voff_c(11u)); }
}

// Column c of the stress sigma and the energy, lane-parallel (computeGreenStrainAndPiolaStress); p14 etc. as in deformation_gradient
__device__ __forceinline__ void quad_green_strain_piola_stress(uint32_t c, V3 x1, V3 x2, V3 x3, V3 x4, V3 imc, float restVolume, float mu, float lambda,
	V3 &sigma_c, float &energy)
{
	const V3 p14 = x1 - x4, p24 = x2 - x4, p34 = x3 - x4;
	// column c of F
	V3 Fc;
	Fc.x = p14.x * imc.x + p24.x * imc.y + p34.x * imc.z;
	Fc.y = p14.y * imc.x + p24.y * imc.y + p34.y * imc.z;
	Fc.z = p14.z * imc.x + p24.z * imc.y + p34.z * imc.z;
	const V3 F0 = qb3<0>(Fc), F1 = qb3<1>(Fc), F2 = qb3<2>(Fc);
	// column c of the strain: e(i, c) = 0.5 (F_i . F_c - delta_ic); the products of e(i, c) and e(c, i) are the same numbers in commuted form
	// and `s - 0.0f` is s, so one expression serves the diagonal and the mirrored off-diagonal entries
	const float d0 = c == 0u ? 1.0f : 0.0f, d1 = c == 1u ? 1.0f : 0.0f, d2 = c == 2u ? 1.0f : 0.0f;
	V3 e;
	e.x = 0.5f * (F0.x * Fc.x + F0.y * Fc.y + F0.z * Fc.z - d0);
	e.y = 0.5f * (F1.x * Fc.x + F1.y * Fc.y + F1.z * Fc.z - d1);
	e.z = 0.5f * (F2.x * Fc.x + F2.y * Fc.y + F2.z * Fc.z - d2);
	// trace from the three diagonal entries, each on its own lane
	const float diag = sel3(c, e.x, e.y, e.z);
	const float trace = qb<0>(diag) + qb<1>(diag) + qb<2>(diag);
	const float ltrace = lambda * trace;
	// column c of s = 2 mu e + lambda trace I
	V3 s = mk(e.x * 2.0f * mu, e.y * 2.0f * mu, e.z * 2.0f * mu);
	const V3 sd = mk(s.x + ltrace, s.y + ltrace, s.z + ltrace);
	s.x = c == 0u ? sd.x : s.x; s.y = c == 1u ? sd.y : s.y; s.z = c == 2u ? sd.z : s.z;
	// column c of sigma = F s:  sigma(i, c) = F(i,0) s(0,c) + (F(i,1) s(1,c) + F(i,2) s(2,c))
	sigma_c.x = F0.x * s.x + (F1.x * s.y + F2.x * s.z);
	sigma_c.y = F0.y * s.x + (F1.y * s.y + F2.y * s.z);
	sigma_c.z = F0.z * s.x + (F1.z * s.y + F2.z * s.z);
	// psi = sum over (j, k), j outer, of e(j,k)^2, starting from 0: the squares of column k sit on lane k
	const V3 sq = mk(e.x * e.x, e.y * e.y, e.z * e.z);
	float psi = 0.0f;
	psi += qb<0>(sq.x); psi += qb<1>(sq.x); psi += qb<2>(sq.x);
	psi += qb<0>(sq.y); psi += qb<1>(sq.y); psi += qb<2>(sq.y);
	psi += qb<0>(sq.z); psi += qb<1>(sq.z); psi += qb<2>(sq.z);
	psi = mu * psi + 0.5f * lambda * trace * trace;
	energy = restVolume * psi;
}

// the lane's own gradient (computeGradCGreen): J_c = column c of sigma Dm^-T V0 on lanes 0..2, -J0 - J1 - J2 on lane 3
__device__ __forceinline__ V3 quad_grad_c_green(uint32_t q, float restVolume, V3 imr /* row c of Dm^-1 */, V3 sigma_c)
{
	const V3 S0 = qb3<0>(sigma_c), S1 = qb3<1>(sigma_c), S2 = qb3<2>(sigma_c);
	V3 J;
	J.x = (S0.x * imr.x + (S1.x * imr.y + S2.x * imr.z)) * restVolume;
	J.y = (S0.y * imr.x + (S1.y * imr.y + S2.y * imr.z)) * restVolume;
	J.z = (S0.z * imr.x + (S1.z * imr.y + S2.z * imr.z)) * restVolume;
	const V3 J0 = qb3<0>(J), J1 = qb3<1>(J), J2 = qb3<2>(J);
	const V3 J3 = (-J0) - J1 - J2;
	return q == 3u ? J3 : J;
}

// the stress of the inversion branch is computed by every lane with the scalar code (rare: a tet crushed below 20 % of its volume) and the
// lane keeps its column
__device__ __forceinline__ void quad_inversion_branch(uint32_t c, V3 x1, V3 x2, V3 x3, V3 x4, V3 imc, float restVolume, float mu, float lambda, V3 &sigma_c, float &energy)
{
	M3 im;
	const V3 c0 = qb3<0>(imc), c1 = qb3<1>(imc), c2 = qb3<2>(imc);
	im.m[0][0] = c0.x; im.m[1][0] = c0.y; im.m[2][0] = c0.z;
	im.m[0][1] = c1.x; im.m[1][1] = c1.y; im.m[2][1] = c1.z;
	im.m[0][2] = c2.x; im.m[1][2] = c2.y; im.m[2][2] = c2.z;
	M3 sigma;
	green_strain_piola_stress_inversion(x1, x2, x3, x4, im, restVolume, mu, lambda, sigma, energy);
	sigma_c = mk(sel3(c, sigma.m[0][0], sigma.m[0][1], sigma.m[0][2]), sel3(c, sigma.m[1][0], sigma.m[1][1], sigma.m[1][2]), sel3(c, sigma.m[2][0], sigma.m[2][1], sigma.m[2][2]));
}

// ---- StrainTetConstraint (PositionBasedDynamics.cpp:713-805): lane q of the quad owns particle q -------------------------------------------
// Each of the six sub-projections (i, j), j <= i, is: P = [x1 - x0 | x2 - x0 | x3 - x0] at the CURRENT corrections, f_i = P c_i, f_j = P c_j,
// S = f_i . f_j, the gradients d_1..3 (one per lane 1..3), d_0 = 0 - d_1 - d_2 - d_3 (lane 0), lambda from sum_q w_q |d_q|^2, corr_q -= lambda w_q d_q.
// Per lane: its particle's position + correction, its gradient, its correction; exchanged: x0 + corr0 (from lane 0), the three columns of P
// (from lanes 1..3), the gradients for d_0, the four terms of the denominator.  f_i, f_j, S and the division are repeated on every lane.
template <bool COMPACT, class A>
__device__ __forceinline__ void exec_rec_quad_strain(const A &a, const QuadLane &l, const RecQ<PBDX_STRAIN_TET, COMPACT> &r, uint32_t slot)
{
	const uint32_t q = l.q;
	const uint32_t idw = q < 2u ? r.w[0] : r.w[1];
	const uint32_t my_id = (q & 1u) ? (idw >> 16) : (idw & 0xffffu);
	V3 my_p; float my_w;
	ldp(a, my_id, my_p, my_w);
	M3 im;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int rr = 0; rr < 3; rr++) im.m[rr][c] = __builtin_bit_cast(float, r.w[3 + c * 3 + rr]);
	float stretchStiffness, shearStiffness; bool normalizeStretch, normalizeShear;
	if constexpr (COMPACT) { stretchStiffness = a.view.u[9]; shearStiffness = a.view.u[10]; normalizeStretch = a.view.u[11] != 0.0f; normalizeShear = a.view.u[12] != 0.0f; }
	else
	{
		stretchStiffness = __builtin_bit_cast(float, r.w[12]); shearStiffness = __builtin_bit_cast(float, r.w[13]);
		normalizeStretch = __builtin_bit_cast(float, r.w[14]) != 0.0f; normalizeShear = __builtin_bit_cast(float, r.w[15]) != 0.0f;
	}
	const uint32_t k = q == 0u ? 0u : q - 1u;          // lanes 1..3 own gradient d_{k+1} and column k of P (lane 0's copies are not used)
	V3 corr = mk(0.0f, 0.0f, 0.0f);
	V3 cc[3];
	cc[0] = get_col(im, 0); cc[1] = get_col(im, 1); cc[2] = get_col(im, 2);
#pragma unroll
	for (int i = 0; i < 3; i++)
	{
#pragma unroll
		for (int j = 0; j <= i; j++)
		{
			const V3 xq = my_p + corr;                       // (p_q + corr_q)
			const V3 x0 = qb3<0>(xq);
			const V3 col = xq - x0;                          // lanes 1..3: column q-1 of P
			const V3 P0 = qb3<1>(col), P1 = qb3<2>(col), P2 = qb3<3>(col);
			// mul(P, c): row r = P(r,0) c.x + (P(r,1) c.y + P(r,2) c.z)
			const V3 fi = mk(P0.x * cc[i].x + (P1.x * cc[i].y + P2.x * cc[i].z), P0.y * cc[i].x + (P1.y * cc[i].y + P2.y * cc[i].z), P0.z * cc[i].x + (P1.z * cc[i].y + P2.z * cc[i].z));
			const V3 fj = mk(P0.x * cc[j].x + (P1.x * cc[j].y + P2.x * cc[j].z), P0.y * cc[j].x + (P1.y * cc[j].y + P2.y * cc[j].z), P0.z * cc[j].x + (P1.z * cc[j].y + P2.z * cc[j].z));
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
			// this lane's gradient d_{k+1} = fj im(k,i) + fi im(k,j)
			const float imki = sel3(k, im.m[0][i], im.m[1][i], im.m[2][i]), imkj = sel3(k, im.m[0][j], im.m[1][j], im.m[2][j]);
			V3 d = fj * imki + fi * imkj;
			if (ns)
				d = s1 * d - (Sij * s3) * (((wj * wj) * fi) * imki + ((wi * wi) * fj) * imkj);
			const V3 d1 = qb3<1>(d), d2 = qb3<2>(d), d3 = qb3<3>(d);
			V3 d0 = mk(0.0f, 0.0f, 0.0f);
			d0 = d0 - d1; d0 = d0 - d2; d0 = d0 - d3;
			if (q == 0u) d = d0;
			if (ns)
				Sij *= s1;
			const float t = my_w * sqn(d);
			float lambda = qb<0>(t) + qb<1>(t) + qb<2>(t) + qb<3>(t);
			if (fabsf(lambda) < PBDX_EPS)
				continue;
			if (i == j)
			{
				if (normalizeStretch) { const float s = sqrtf(Sij); lambda = 2.0f * s * (s - 1.0f) / lambda * stretchStiffness; }
				else lambda = (Sij - 1.0f) / lambda * stretchStiffness;
			}
			else
				lambda = Sij / lambda * shearStiffness;
			corr = corr - (lambda * my_w) * d;
		}
	}
	if (my_w != 0.0f)
		a.st(my_id, make_float4(my_p.x + corr.x, my_p.y + corr.y, my_p.z + corr.z, my_w));
}

template <int TYPE, bool COMPACT, class A>
__device__ __forceinline__ void exec_rec_quad(const A &a, const QuadLane &l, const RecQ<TYPE, COMPACT> &r, uint32_t slot, float dt, int first_iter)
{
	if constexpr (TYPE == PBDX_STRAIN_TET) { exec_rec_quad_strain<COMPACT>(a, l, r, slot); return; }
	else {
	static_assert(TYPE == PBDX_FEM_TET || TYPE == PBDX_FEM_TET_XPBD || TYPE == PBDX_STRAIN_TET, "quad-lane projection: FEM tets, strain tets");
	const uint32_t id[4] = { r.w[0] & 0xffffu, r.w[0] >> 16, r.w[1] & 0xffffu, r.w[1] >> 16 };
	V3 p0, p1, p2, p3; float w0, w1, w2, w3;
	ldp(a, id[0], p0, w0); ldp(a, id[1], p1, w1); ldp(a, id[2], p2, w2); ldp(a, id[3], p3, w3);
	const float restVolume = __builtin_bit_cast(float, r.w[3]);
	const V3 imc = mk(__builtin_bit_cast(float, r.w[4]), __builtin_bit_cast(float, r.w[5]), __builtin_bit_cast(float, r.w[6]));
	const V3 imr = mk(__builtin_bit_cast(float, r.w[7]), __builtin_bit_cast(float, r.w[8]), __builtin_bit_cast(float, r.w[9]));
	float youngsModulus, poissonRatio;
	if constexpr (COMPACT) { youngsModulus = a.view.u[10]; poissonRatio = a.view.u[11]; }
	else { youngsModulus = __builtin_bit_cast(float, r.w[10]); poissonRatio = __builtin_bit_cast(float, r.w[11]); }
	float multiplier = first_iter ? 0.0f : __builtin_bit_cast(float, r.w[2]);

	// the lane's own particle
	const uint32_t q = l.q;
	const uint32_t my_id = q == 0u ? id[0] : (q == 1u ? id[1] : (q == 2u ? id[2] : id[3]));
	const V3 my_p = q == 0u ? p0 : (q == 1u ? p1 : (q == 2u ? p2 : p3));
	const float my_w = q == 0u ? w0 : (q == 1u ? w1 : (q == 2u ? w2 : w3));

	bool ok = true;
	V3 corr = mk(0.0f, 0.0f, 0.0f);
	if (!(youngsModulus <= 0.0f))
	{
		if (poissonRatio < 0.0f || (double)poissonRatio > 0.49)
			ok = false;
		else
		{
			const bool handleInversion = fem_tet_handle_inversion(p0, p1, p2, p3, restVolume);
			const float volume = dot(cross(p1 - p0, p2 - p0), p3 - p0) / 6.0f;
			float mu, lambda;
			if constexpr (TYPE == PBDX_FEM_TET)
			{
				mu = youngsModulus / 2.0f / (1.0f + poissonRatio);
				lambda = youngsModulus * poissonRatio / (1.0f + poissonRatio) / (1.0f - 2.0f * poissonRatio);
			}
			else
			{
				mu = (float)(1.0 / (double)2.0f / (double)(1.0f + poissonRatio));
				lambda = (float)(1.0 * (double)poissonRatio / (double)(1.0f + poissonRatio) / (double)(1.0f - 2.0f * poissonRatio));
			}
			V3 sigma_c; float U;
			if (!handleInversion || volume > 0.0f)
				quad_green_strain_piola_stress(l.c, p0, p1, p2, p3, imc, restVolume, mu, lambda, sigma_c, U);
			else
				quad_inversion_branch(l.c, p0, p1, p2, p3, imc, restVolume, mu, lambda, sigma_c, U);
			const V3 g = quad_grad_c_green(q, restVolume, imr, sigma_c);
			const float a_q = my_w * sqn(g);
			float sum = qb<0>(a_q) + qb<1>(a_q) + qb<2>(a_q) + qb<3>(a_q);
			if constexpr (TYPE == PBDX_FEM_TET)
			{
				if (sum < PBDX_EPS)
					ok = false;
				else
				{
					const float s = U / sum;
					corr = (-s * my_w) * g;
				}
			}
			else
			{
				const float C = (float)sqrt(2.0 * (double)U);
				const float alpha = 1.0f / (youngsModulus * dt * dt);
				sum += C * C * alpha;
				if (sum < PBDX_EPS)
					ok = false;
				else
				{
					const float lam = -C * (C + alpha * multiplier) / sum;
					multiplier += lam;
					corr = (lam * my_w) * g;
				}
			}
		}
	}
	if (ok && my_w != 0.0f)
		a.st(my_id, make_float4(my_p.x + corr.x, my_p.y + corr.y, my_p.z + corr.z, my_w));
	if constexpr (kHasLambda[TYPE])
		if (q == 0u) a.lam_store(slot, multiplier);
	}
}

} // namespace pbdx

#endif
