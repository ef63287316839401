// pbdx_chainsum.h -- the sequential float sum  s <- RN(s + x_i), i = 0 .. n-1  (round to nearest even, s_0 = +0), evaluated EXACTLY --
// every intermediate rounding reproduced -- without doing the additions one after the other.  Host + device code.
//
// Why: the centre of a bounding sphere is such a sum over the node's vertices in list order (BoundingSphereHierarchy.cpp:34-51,72-98);
// the root of a hierarchy sums everything, and a dependent v_add_f32 costs 1.75 ns on this GPU whatever else the GPU does
// (pbdx_tetcontact_dev.h): for two bars of 610 k tets the root's chain is 5.4 ms of an 8.9 ms step.
//
// How: while the running sum stays in one binade [2^e, 2^(e+1)) it lives on the grid u = 2^(e-23): s = S u with an integer
// 2^23 <= S < 2^24, and adding x moves it by an integer: with x / u = X + f (X = floor, 0 <= f < 1)
//     S' = S + X + r,   r = 0 if f < 1/2,  1 if f > 1/2,  (S + X) mod 2 if f = 1/2   (ties to even)
// as long as 2^23 <= S + X <= 2^24 - 1 (the exact sum is still inside the binade).  An element is therefore a function of S that depends
// on S only through its PARITY, and a run of elements is again such a function: "add a_0 if S was even, a_1 if it was odd".  These
// functions compose associatively, so the state after every element of a run follows from a parallel prefix scan over the elements'
// functions, and a second pass checks the in-binade condition at every element.  The first element that violates it ends the run: the
// state before it is exact, ONE real float addition is made there, and the next run starts on the new binade.  A sum that changes
// binade all the time (around zero) degenerates into those single additions -- still exact, no longer parallel.
// Negative sums are handled by their mirror image (RN is odd: RN(-y) = -RN(y)).
#ifndef PBDX_CHAINSUM_H
#define PBDX_CHAINSUM_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PBDX_CS_HD __host__ __device__ inline
#else
#define PBDX_CS_HD inline
#endif

namespace pbdx {

PBDX_CS_HD uint32_t cs_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PBDX_CS_HD float cs_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

enum { kCsDown = 0, kCsUp = 1, kCsTie = 2, kCsOut = 3 };
struct CsElem { uint32_t X; uint32_t cls; };           // x / u = X + f on the run's grid; cls: where f lies; kCsOut: certainly leaves the binade

// the run's frame: sign of the running sum and its binade
struct CsFrame { bool negate; int e; };
// s finite, normal, non-zero -> its frame and its integer S (2^23 <= S < 2^24)
PBDX_CS_HD bool cs_frame_of(float s, CsFrame &fr, uint32_t &S)
{
	const uint32_t b = cs_bits(s);
	const uint32_t ex = (b >> 23) & 255u;
	if (ex == 0u || ex == 255u) return false;           // zero, denormal, infinite, NaN: no run
	fr.negate = (b >> 31) != 0u;
	fr.e = (int)ex - 127;
	S = (b & 0x7fffffu) | 0x800000u;
	return true;
}
// S (2^23 <= S <= 2^24) in the frame -> the float
PBDX_CS_HD float cs_value(const CsFrame &fr, uint32_t S)
{
	// S = 2^24 is 2^(e+1) exactly; otherwise mantissa bits of S with exponent e
	const uint32_t sign = fr.negate ? 0x80000000u : 0u;
	if (S == 0x1000000u) return cs_float(sign | ((uint32_t)(fr.e + 1 + 127) << 23));
	return cs_float(sign | ((uint32_t)(fr.e + 127) << 23) | (S & 0x7fffffu));
}

PBDX_CS_HD CsElem cs_classify(float x, const CsFrame &fr)
{
	CsElem q; q.X = 0u; q.cls = kCsDown;
	const uint32_t b = cs_bits(x);
	const bool neg = ((b >> 31) != 0u) != fr.negate;
	const uint32_t ex = (b >> 23) & 255u, frac = b & 0x7fffffu;
	if (ex == 255u) { q.X = 0x1000000u; q.cls = kCsOut; return q; }
	const uint32_t m = ex ? (frac | 0x800000u) : frac;  // |x| = m 2^(ee - 23)
	if (m == 0u) return q;
	const int ee = ex ? (int)ex - 127 : -126;
	const int d = ee - fr.e;                            // |x| / u = m 2^d
	if (d > 0 || (fr.e + 1 + 127) >= 255) { q.X = 0x1000000u; q.cls = kCsOut; return q; }
	if (d == 0) { q.X = neg ? (0u - m) : m; return q; }
	const int k = -d;
	uint32_t quo, rem_is_zero, cmp;                     // cmp: 0 rem < half, 1 rem == half, 2 rem > half
	if (k >= 25) { quo = 0u; rem_is_zero = 0u; cmp = 0u; }
	else
	{
		quo = m >> k;
		const uint32_t rem = m & ((1u << k) - 1u), half = 1u << (k - 1);
		rem_is_zero = rem == 0u ? 1u : 0u;
		cmp = rem < half ? 0u : rem == half ? 1u : 2u;
	}
	if (rem_is_zero) { q.X = neg ? (0u - quo) : quo; return q; }
	if (!neg) { q.X = quo; q.cls = cmp == 0u ? kCsDown : cmp == 1u ? kCsTie : kCsUp; }
	else { q.X = 0u - quo - 1u; q.cls = cmp == 2u ? kCsDown : cmp == 1u ? kCsTie : kCsUp; }      // f = 1 - rem / 2^k
	return q;
}

// one element applied to the actual state; in_binade: the exact sum stays in [2^e, 2^(e+1)) (otherwise S' is meaningless)
PBDX_CS_HD uint32_t cs_apply(uint32_t S, const CsElem &q, bool &in_binade)
{
	const uint32_t T = S + q.X;
	in_binade = q.cls != kCsOut && (T - 0x800000u) <= 0x7fffffu;
	const uint32_t r = q.cls == kCsUp ? 1u : q.cls == kCsTie ? (T & 1u) : 0u;
	return T + r;
}

// a run of elements as a function of the state's parity: S' = S + a[S & 1]   (arithmetic mod 2^32: garbage past a violation stays garbage)
struct CsFun { uint32_t a[2]; };
PBDX_CS_HD CsFun cs_identity() { CsFun f; f.a[0] = 0u; f.a[1] = 0u; return f; }
PBDX_CS_HD void cs_push(CsFun &f, const CsElem &q)      // f <- (q after f)
{
	for (int p = 0; p < 2; p++)
	{
		const uint32_t par = ((uint32_t)p + f.a[p]) & 1u;                       // parity of the state before q, if it started with parity p
		const uint32_t t = q.X + (q.cls == kCsUp ? 1u : q.cls == kCsTie ? ((par + q.X) & 1u) : 0u);
		f.a[p] += t;
	}
}
PBDX_CS_HD CsFun cs_then(const CsFun &first, const CsFun &second)
{
	CsFun h;
	for (int p = 0; p < 2; p++) h.a[p] = first.a[p] + second.a[((uint32_t)p + first.a[p]) & 1u];
	return h;
}

// ---- host reference of the whole procedure, organised as the device kernel is (blocks of `per_thread` elements, `threads` blocks per
// window): what tests/test_chainsum.py compares with the plain loop ---------------------------------------------------------------------
inline float cs_sum_blocked_host(const float *x, uint64_t n, uint32_t threads, uint32_t per_thread, uint64_t *single_additions)
{
	float s = 0.0f;
	uint64_t pos = 0, singles = 0;
	while (pos < n)
	{
		CsFrame fr; uint32_t S0;
		if (!cs_frame_of(s, fr, S0)) { s = s + x[pos++]; singles++; continue; }
		const uint64_t window = (n - pos < (uint64_t)threads * per_thread) ? n - pos : (uint64_t)threads * per_thread;
		// pass 1: every block's function; scan
		const uint32_t blocks = (uint32_t)((window + per_thread - 1) / per_thread);
		CsFun prefix = cs_identity();                 // functions of the blocks before the current one, composed
		uint64_t stop = window;                       // first element (window-relative) at which the run ends
		uint32_t S_stop = 0; bool stop_inside = false;
		for (uint32_t b = 0; b < blocks && stop == window; b++)
		{
			uint32_t S = S0 + prefix.a[S0 & 1u];
			CsFun f = cs_identity();
			for (uint32_t k = 0; k < per_thread; k++)
			{
				const uint64_t i = (uint64_t)b * per_thread + k;
				if (i >= window) break;
				const CsElem q = cs_classify(x[pos + i], fr);
				bool ok;
				const uint32_t S1 = cs_apply(S, q, ok);
				if (!ok) { stop = i; S_stop = S; stop_inside = false; break; }
				if (S1 == 0x1000000u) { stop = i; S_stop = S1; stop_inside = true; break; }      // rounded up into the next binade: valid, run ends after it
				cs_push(f, q);
				S = S1;
			}
			if (stop == window) { prefix = cs_then(prefix, f); if (b + 1 == blocks) S_stop = S; }
		}
		if (stop == window) { s = cs_value(fr, S_stop); pos += window; continue; }
		if (stop_inside) { s = cs_value(fr, S_stop); pos += stop + 1; continue; }
		s = cs_value(fr, S_stop);
		s = s + x[pos + stop];                        // the one real addition
		singles++;
		pos += stop + 1;
	}
	if (single_additions) *single_additions = singles;
	return s;
}

// the device kernel's POLICY (windows, fall-back bursts) replayed on the host, with counts: stats[0] window attempts, [1] values gained by
// them, [2] bursts, [3] values summed the plain way
inline float cs_sum_policy_host(const float *x, uint64_t n, uint32_t window_max, uint32_t poor_below, uint32_t burst0, uint64_t stats[4])
{
	float s = 0.0f;
	uint64_t pos = 0; uint32_t poor = 0; bool plain_next = true;
	stats[0] = stats[1] = stats[2] = stats[3] = 0;
	while (pos < n)
	{
		CsFrame fr; uint32_t S = 0;
		const bool framed = cs_frame_of(s, fr, S);
		if (plain_next || !framed)
		{
			const uint64_t want = (uint64_t)burst0 << (poor < 4u ? poor : 4u);
			const uint64_t len = n - pos < want ? n - pos : want;
			for (uint64_t i = 0; i < len; i++) { volatile float t = s + x[pos + i]; s = t; }
			pos += len; plain_next = false; stats[2]++; stats[3] += len;
			continue;
		}
		const uint64_t window = n - pos < window_max ? n - pos : window_max;
		uint64_t i = 0; bool inside = false, stopped = false;
		for (; i < window; i++)
		{
			bool ok;
			const uint32_t S1 = cs_apply(S, cs_classify(x[pos + i], fr), ok);
			if (!ok) { stopped = true; break; }
			S = S1;
			if (S1 == 0x1000000u) { stopped = true; inside = true; break; }
		}
		stats[0]++;
		s = cs_value(fr, S);
		if (!stopped) { pos += window; stats[1] += window; poor = 0; continue; }
		if (!inside) { volatile float t = s + x[pos + i]; s = t; }
		pos += i + 1; stats[1] += i + 1;
		if (i + 1 < poor_below) { plain_next = true; poor++; } else poor = 0;
	}
	return s;
}

} // namespace pbdx
#endif
