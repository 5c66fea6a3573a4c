// qp_lane.hpp -- what the in-register QP solvers share: the per-lane problem record, the lane-group
// reductions (DPP butterflies), and the register LDL' of an nv x nv matrix.
//
// The solver headers built on this file (gi_small.hpp) also compile for the host with one lane per QP
// (G = 1, every group operation is the identity) so that tests/host_gi_driver.cpp can run the very same
// arithmetic on the CPU against the oracle.  ASIF_HD is host+device under hipcc and plain inline under g++;
// that is a host/device split of one code base, not a second implementation.
#pragma once
#include <type_traits>
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ASIF_HD __host__ __device__ __forceinline__
#else
#define ASIF_HD inline
#endif

namespace asif {

// One QP as the filter classes hand it to QPWrapperAbstract (include/qpwrapper_abstract.h:30-43),
// this lane's share of the rows only.
template <int NV, int RPL>
struct QpLaneData {
	double Hd[NV], c[NV], lb[NV], ub[NV];
	double A[RPL][NV], b[RPL];
	bool eq[RPL];
};

// Cross-lane exchange inside a group of G <= 16 consecutive lanes by DPP (data-parallel primitives:
// the permutation rides on the VALU operand fetch, no LDS crossbar round trip like ds_bpermute).
// Butterfly stage m: 1 -> quad_perm[1,0,3,2], 2 -> quad_perm[2,3,0,1]; after those two all four lanes
// of a quad agree, so stage 4 may use row_half_mirror (i <-> 7-i) and stage 8 row_mirror (i <-> 15-i).
// (The host body is the identity: host code only ever runs one lane per QP.)
template <int M>
ASIF_HD int dpp_xchg(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	constexpr int ctrl = M == 1 ? 0xB1 : (M == 2 ? 0x4E : (M == 4 ? 0x141 : 0x140));
	return __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);
#else
	return v;
#endif
}
template <int M>
ASIF_HD double dpp_xchg(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const int lo = dpp_xchg<M>(__double2loint(v)), hi = dpp_xchg<M>(__double2hiint(v));
	return __hiloint2double(hi, lo);
#else
	return v;
#endif
}

template <int G>
ASIF_HD double gsum(double v)
{
	static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "lanes per QP");
	if (G >= 2) v += dpp_xchg<1>(v);
	if (G >= 4) v += dpp_xchg<2>(v);
	if (G >= 8) v += dpp_xchg<4>(v);
	if (G >= 16) v += dpp_xchg<8>(v);
	return v;
}
template <int G>
ASIF_HD double gmax(double v)
{
	if (G >= 2) v = fmax(v, dpp_xchg<1>(v));
	if (G >= 4) v = fmax(v, dpp_xchg<2>(v));
	if (G >= 8) v = fmax(v, dpp_xchg<4>(v));
	if (G >= 16) v = fmax(v, dpp_xchg<8>(v));
	return v;
}
template <int G>
ASIF_HD double gmin(double v)
{
	if (G >= 2) v = fmin(v, dpp_xchg<1>(v));
	if (G >= 4) v = fmin(v, dpp_xchg<2>(v));
	if (G >= 8) v = fmin(v, dpp_xchg<4>(v));
	if (G >= 16) v = fmin(v, dpp_xchg<8>(v));
	return v;
}
template <int G>
ASIF_HD int gand(int p)
{
	if (G >= 2) p &= dpp_xchg<1>(p);
	if (G >= 4) p &= dpp_xchg<2>(p);
	if (G >= 8) p &= dpp_xchg<4>(p);
	if (G >= 16) p &= dpp_xchg<8>(p);
	return p;
}
template <int G>
ASIF_HD int gor(int p)
{
	if (G >= 2) p |= dpp_xchg<1>(p);
	if (G >= 4) p |= dpp_xchg<2>(p);
	if (G >= 8) p |= dpp_xchg<4>(p);
	if (G >= 16) p |= dpp_xchg<8>(p);
	return p;
}
template <int G>
ASIF_HD int gmin_int(int p)
{
	if (G >= 2) { const int o = dpp_xchg<1>(p); p = o < p ? o : p; }
	if (G >= 4) { const int o = dpp_xchg<2>(p); p = o < p ? o : p; }
	if (G >= 8) { const int o = dpp_xchg<4>(p); p = o < p ? o : p; }
	if (G >= 16) { const int o = dpp_xchg<8>(p); p = o < p ? o : p; }
	return p;
}
// Problem data outside the solvers' domain, one verdict per lane group: NaN, infinite, or so large (beyond ~1e148)
// that a product of two entries overflows.  A NaN or infinite state reaches the rows through h, Lfh, Lgh; OSQP takes
// such data as it is and never converges on it (every residual comparison is false on NaN): it runs to max_iter and
// QPWrapperOsqp::solve returns that raw status (src/qpwrapper_osqp.cpp:225-238).  The comparisons of the solvers here
// would read a NaN row as "not violated" (and inf > inf as false), so they ask first.  One fused multiply-add per
// entry: v * 1e160 overflows for |v| > 1.8e148 and the running sum is then inf or NaN; one test at the end.
// Bounds may be infinite (one-sided), not NaN.
template <int NV, int RPL, int G>
ASIF_HD bool qp_data_nonfinite(const double (&Hd)[NV], const double (&c)[NV], const double (&lb)[NV], const double (&ub)[NV],
                               const double (&A)[RPL][NV], const double (&b)[RPL])
{
	constexpr double kBig = 1e160;
	double s = 0.0;
	int bad = 0;
#pragma unroll
	for (int j = 0; j < NV; j++) {
		s = fma(Hd[j], kBig, s);
		s = fma(c[j], kBig, s);
		bad |= ((lb[j] != lb[j]) | (ub[j] != ub[j])) ? 1 : 0;
	}
#pragma unroll
	for (int k = 0; k < RPL; k++) {
#pragma unroll
		for (int j = 0; j < NV; j++) s = fma(A[k][j], kBig, s);
		s = fma(b[k], kBig, s);
	}
	bad |= !(fabs(s) < __builtin_huge_val()) ? 1 : 0;
	return gor<G>(bad) != 0;
}
// compile-time loop with early exit: f(integral_constant<int, I>) for I = 0 .. N-1 until one returns true
template <int N, int I = 0, class F>
ASIF_HD bool unrolled_until(F &&f)
{
	if constexpr (I < N) {
		if (f(std::integral_constant<int, I>())) return true;
		return unrolled_until<N, I + 1>(f);
	}
	return false;
}
// every lane of the wavefront agrees (host: the one "lane")
ASIF_HD bool wave_all(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __all(p);
#else
	return p;
#endif
}
ASIF_HD bool wave_any(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __any(p);
#else
	return p;
#endif
}
// ranking-grade 1/sqrt (hardware seed on the device; only ever used to order candidates)
ASIF_HD double rank_rsqrt(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_rsq(v);
#else
	return 1.0 / sqrt(v);
#endif
}
// reciprocal of a positive number far from the denormal range: hardware seed + two Newton steps instead of
// the full IEEE division sequence (scale / fixup handling not needed)
ASIF_HD double pos_rcp(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r = __builtin_amdgcn_rcp(d);
	r = fma(fma(-d, r, 1.0), r, r);
	r = fma(fma(-d, r, 1.0), r, r);
	return r;
#else
	return 1.0 / d;
#endif
}

// LDL' of a symmetric NV x NV matrix given by its lower triangle M[a][b], a >= b, in place:
// on return M holds L (strictly lower) and Dinv = 1/d.  False if a pivot is not positive.
template <int NV>
ASIF_HD bool ldl_factor(double (&M)[NV][NV], double (&Dinv)[NV])
{
	bool ok = true;
	double d[NV];
#pragma unroll
	for (int j = 0; j < NV; j++) {
		double dj = M[j][j];
#pragma unroll
		for (int k = 0; k < j; k++) dj -= M[j][k] * M[j][k] * d[k];
		ok = ok && (dj > 0.0);
		d[j] = dj;
		const double di = pos_rcp(dj);
		Dinv[j] = di;
#pragma unroll
		for (int i = j + 1; i < NV; i++) {
			double s = M[i][j];
#pragma unroll
			for (int k = 0; k < j; k++) s -= M[i][k] * M[j][k] * d[k];
			M[i][j] = s * di;
		}
	}
	return ok;
}
template <int NV>
ASIF_HD void ldl_solve(const double (&L)[NV][NV], const double (&Dinv)[NV], double (&v)[NV])
{
#pragma unroll
	for (int i = 1; i < NV; i++)
#pragma unroll
		for (int k = 0; k < i; k++) v[i] -= L[i][k] * v[k];
#pragma unroll
	for (int i = 0; i < NV; i++) v[i] *= Dinv[i];
#pragma unroll
	for (int i = NV - 2; i >= 0; i--)
#pragma unroll
		for (int k = i + 1; k < NV; k++) v[i] -= L[k][i] * v[k];
}

} // namespace asif
