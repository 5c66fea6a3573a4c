// k_robust_data.hip -- ASIFrobust::filter (src/asif_robust.cpp:218-367) on the model and data the reference
// ships and builds by default: examples/DoubleIntegrator_Robust.cpp (interval mass, gain, friction) with the
// half-planes of include/KernelData_*.h as safety set, npSSmax of npSS rows kept (5 of 100).  nx == 2, nu == 1.
//
// robust_data_point_kernel (once per handle): the state-independent pieces of the point-state interval
//   dynamics -- the affine forms of inv(m) and K/m -- evaluated generically with affine_dev.hpp.
// Fused filter kernel = qp_policy_kernel<2,16,G> with RobustDataPolicy::load:
//   1. scan of the half-planes, shared by the G lanes of the QP's group (lane g takes rows g, g+G, ...):
//      h_i = 1 - a_i.x, each lane's smallest kept sorted in registers, then the group's npSSmax smallest
//      extracted head by head with DPP minima (:296-315; std::sort's order of equal keys is unspecified,
//      lowest index first here);
//   2. interval Lie derivatives of the kept rows at the point state.  As in k_robust.hip the only non-zero
//      noise coefficients come from the interval parameters; written out with libaffa's roundings:
//        f1 = (-F*x1)/m   centre cu*ci, coefficients {cu*am, ci*uF, cu*de, radu*ri}    (cu = -Fc*x1, uF = x1*(-Fr))
//        g1 = K/m         centre and four coefficients, state-independent
//        Lfh_s = (0 + d0*x1) + d1*f1c  -+  sum |d1*coef|        Lgh_s = d1*g1c  -+  sum |d1*gcoef|
//   3. the multipliers eliminated exactly (nu == 1, k_robust.hip): rows [lo(Lgh), h] and [hi(Lgh), h] >= -lo(Lfh);
//   4. the 2-variable QP on the in-register ADMM, clamp, relax, rc 1 / -1.
// asif_hip_assemble_batch (robust_data_rows_kernel) writes the full 3M x (2+4M) rows instead.
#include "affine_dev.hpp"
#include "qp_kernel.hpp"

namespace asif {

constexpr int kRbMaxRows = 8; // npSSmax <= 8: 16 reduced rows = the <2,16,G> QP kernels of the robust filter

__global__ void robust_data_point_kernel(RbDev z, double *pc)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	AfCtx cx = {0u, false};
	Af m, K, F, im, nF, g1;
	af_interval(cx, m, z.mMin, z.mMax);
	af_interval(cx, K, z.Klo, z.Khi);
	af_interval(cx, F, z.Flo, z.Fhi);
	af_inv(cx, m, im); // {eps_m, new}
	af_neg(F, nF);
	pc[0] = im.c;
	pc[1] = im.v[0];
	pc[2] = im.v[1];
	pc[3] = af_rad(im);
	pc[4] = nF.c;
	pc[5] = nF.v[0];
	af_div(cx, K, m, g1); // K*inv(m): {eps_m, eps_K, new(inv), new(product)}
	pc[6] = g1.c;
	for (int k = 0; k < 4; k++) pc[7 + k] = k < g1.n ? g1.v[k] : 0.0;
	pc[11] = cx.overflow ? 1.0 : 0.0;
}

struct RbRows {
	double h[kRbMaxRows], fl[kRbMaxRows], fh[kRbMaxRows], gl[kRbMaxRows], gh[kRbMaxRows];
	int idx[kRbMaxRows];
};

template <int G>
__device__ __forceinline__ int gmini(int v)
{
	if (G >= 2) v = min(v, dpp_xchg<1>(v));
	if (G >= 4) v = min(v, dpp_xchg<2>(v));
	if (G >= 8) v = min(v, dpp_xchg<4>(v));
	if (G >= 16) v = min(v, dpp_xchg<8>(v));
	return v;
}

// steps 1 and 2 for one instance, by the G lanes of its group (g = this lane's place in it)
template <int G>
__device__ __forceinline__ void robust_data_rows(const RbDev &z, double x0, double x1, int g, RbRows &R)
{
#pragma clang fp contract(off)
	// 1a. every lane keeps the smallest of its share of the half-planes (g, g+G, ...), sorted by (h, index)
	double lh[kRbMaxRows];
	int li[kRbMaxRows];
#pragma unroll
	for (int q = 0; q < kRbMaxRows; q++) {
		lh[q] = __builtin_huge_val();
		li[q] = 0x7fffffff;
	}
	for (int i = g; i < z.N; i += G) {
		const double a0 = z.hp[2 * i], a1 = z.hp[2 * i + 1];
		double hv = 1. - a0 * x0 - a1 * x1; // examples/DoubleIntegrator_Robust.cpp:45
		int hi_ = i;
#pragma unroll
		for (int q = 0; q < kRbMaxRows; q++) { // sorted insert, strict < (indexes arrive in increasing order)
			const bool lt = hv < lh[q];
			const double tv = lh[q];
			const int ti = li[q];
			lh[q] = lt ? hv : tv;
			li[q] = lt ? hi_ : ti;
			hv = lt ? tv : hv;
			hi_ = lt ? ti : hi_;
		}
	}
	// 1b. the group's npSSmax smallest, extracted head by head: smallest h over the lanes' heads, lowest index on
	//     ties (std::sort leaves equal keys unspecified), the owner pops.  All lanes end with the same list.
#pragma unroll
	for (int k = 0; k < kRbMaxRows; k++) {
		R.h[k] = __builtin_huge_val();
		R.idx[k] = 0;
		if (k < z.npSSmax) { // wave-uniform
			const double hm = gmin<G>(lh[0]);
			const int im = gmini<G>(lh[0] == hm ? li[0] : 0x7fffffff);
			R.h[k] = hm;
			R.idx[k] = im;
			const bool mine = (lh[0] == hm) && (li[0] == im);
#pragma unroll
			for (int q = 0; q + 1 < kRbMaxRows; q++) {
				lh[q] = mine ? lh[q + 1] : lh[q];
				li[q] = mine ? li[q + 1] : li[q];
			}
			lh[kRbMaxRows - 1] = mine ? __builtin_huge_val() : lh[kRbMaxRows - 1];
			li[kRbMaxRows - 1] = mine ? 0x7fffffff : li[kRbMaxRows - 1];
		}
	}
	if (z.npSSmax == z.N) { // no selection (:316-320): rows in data order
#pragma unroll
		for (int q = 0; q < kRbMaxRows; q++)
			if (q < z.N) {
				R.idx[q] = q;
				R.h[q] = 1. - z.hp[2 * q] * x0 - z.hp[2 * q + 1] * x1;
			}
	}
	// point-state interval dynamics (see the header of this file and k_realizable.hip::pointDynamicsMid)
	const double *pc = z.pointC;
	const double ci = pc[0], am = pc[1], de = pc[2], ri = pc[3], nFc = pc[4], nFr = pc[5], g1c = pc[6];
	const double f0 = (x1 + x1) / 2;
	const double cu = nFc * f0;
	const double uF = f0 * nFr, ux = nFc * 0.0, un = fabs(nFr) * 0.0;
	const double radu = ((0.0 + fabs(uF)) + fabs(ux)) + fabs(un);
	const double f1c = cu * ci;
	const double v0 = cu * am, v1 = ci * uF, v2 = ci * ux, v3 = ci * un, v4 = cu * de, v5 = radu * ri;
#pragma unroll
	for (int q = 0; q < kRbMaxRows; q++) {
		const double a0 = z.hp[2 * R.idx[q]], a1 = z.hp[2 * R.idx[q] + 1]; // per-lane gather, cache resident
		const double d0 = (-a0 + -a0) / 2, d1 = (-a1 + -a1) / 2;           // centres of AAF(interval(Dh))
		const double cf = (0.0 + d0 * f0) + d1 * f1c;
		double rf = 0.0;
		rf += fabs(d1 * v0);
		rf += fabs(d1 * v1);
		rf += fabs(d1 * v2);
		rf += fabs(d1 * v3);
		rf += fabs(d1 * v4);
		rf += fabs(d1 * v5);
		R.fl[q] = cf - rf;
		R.fh[q] = cf + rf;
		const double cg = (0.0 + d0 * 0.0) + d1 * g1c;
		double rg = 0.0;
		rg += fabs(d1 * pc[7]);
		rg += fabs(d1 * pc[8]);
		rg += fabs(d1 * pc[9]);
		rg += fabs(d1 * pc[10]);
		R.gl[q] = cg - rg;
		R.gh[q] = cg + rg;
	}
}

// asif_hip_assemble_batch: the full rows of src/asif_robust.cpp:103-133,340-358
__global__ __launch_bounds__(64) void robust_data_rows_kernel(RbDev z, FilterArgs a)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.B) return;
	const int64_t ld = a.ld;
	const int M = z.npSSmax, nv = 2 + 4 * M, nc = 3 * M;
	RbRows R;
	robust_data_rows<1>(z, a.x[i], a.x[ld + i], 0, R);
	for (int e = 0; e < nc * nv; e++) a.A[(int64_t)e * ld + i] = 0.0;
	for (int r = 0; r < nc; r++) a.b[(int64_t)r * ld + i] = 0.0;
#pragma unroll
	for (int s = 0; s < kRbMaxRows; s++)
		if (s < M) {
			const int iRow = 3 * s, iCol = 2 + 4 * s;
			a.A[(int64_t)(iRow + 1 * nc) * ld + i] = R.h[s];
			a.A[(int64_t)(iRow + (iCol + 0) * nc) * ld + i] = R.gl[s];
			a.A[(int64_t)(iRow + (iCol + 2) * nc) * ld + i] = -R.gh[s];
			a.A[(int64_t)(iRow + (iCol + 1) * nc) * ld + i] = R.fl[s];
			a.A[(int64_t)(iRow + (iCol + 3) * nc) * ld + i] = -R.fh[s];
			a.A[(int64_t)((iRow + 1) + 0 * nc) * ld + i] = -1.0;
			a.A[(int64_t)((iRow + 1) + (iCol + 0) * nc) * ld + i] = 1.0;
			a.A[(int64_t)((iRow + 1) + (iCol + 2) * nc) * ld + i] = -1.0;
			a.A[(int64_t)((iRow + 2) + (iCol + 1) * nc) * ld + i] = 1.0;
			a.A[(int64_t)((iRow + 2) + (iCol + 3) * nc) * ld + i] = -1.0;
			a.b[(int64_t)(iRow + 2) * ld + i] = 1.0;
			if (a.diag) a.diag[(int64_t)s * ld + i] = (double)R.idx[s];
		}
	a.code[i] = 1;
}

struct RobustDataPolicy {
	int64_t B;
	RbDev z;
	FilterArgs a;

	// reduced row r = 2s + p is [p ? hi(Lgh_s) : lo(Lgh_s), h_s] >= -lo(Lfh_s); lane g of the group owns rows
	// g, g+G, ...: s = (G/2) k + (g >> 1), p = g & 1 (same dealing as RobustPolicy in k_robust.hip)
	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
		static_assert(NV == 2 && NC == 2 * kRbMaxRows && (G == 2 || G == 4 || G == 8), "reduced robust QP");
		constexpr int RPL = (NC + G - 1) / G, H = G / 2;
		RbRows R;
		robust_data_rows<G>(z, a.x[i], a.x[a.ld + i], g, R);
		const int M = z.npSSmax;
		const bool hiRow = (g & 1) != 0;
		const int sub = g >> 1;
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			double lg = 0.0, hs = 0.0, rhs = -1e20;
#pragma unroll
			for (int q = 0; q < H; q++) {
				const int s = H * k + q; // compile-time
				if (s < kRbMaxRows) {
					const bool pick = (q == sub) && (s < M);
					lg = pick ? (hiRow ? R.gh[s] : R.gl[s]) : lg;
					hs = pick ? R.h[s] : hs;
					rhs = pick ? -R.fl[s] : rhs;
				}
			}
			qp.A[k][0] = lg;
			qp.A[k][1] = hs;
			qp.b[k] = rhs;
			qp.eq[k] = false;
		}
		// src/asif_robust.cpp:89-101,140-142 restricted to (u, delta)
		qp.Hd[0] = 1.0;
		qp.Hd[1] = z.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * z.relaxCost * z.relaxLb;
		qp.lb[0] = z.lb;
		qp.lb[1] = z.relaxLb;
		qp.ub[0] = z.ub;
		qp.ub[1] = z.inf;
		if (a.diag && g == 0) {
#pragma unroll
			for (int s = 0; s < kRbMaxRows; s++)
				if (s < M) a.diag[(int64_t)s * a.ld + i] = (double)R.idx[s];
		}
	}
	template <int NV>
	__device__ __forceinline__ void store(int64_t i, const double (&sol)[NV], int st, int it) const
	{
		if (st == kStatusSolved) {
			a.uact[i] = fmin(fmax(sol[0], z.lb), z.ub);
			a.relax[i] = sol[1];
			a.rc[i] = ASIF_HIP_RC_OK;
		} else {
			a.rc[i] = ASIF_HIP_RC_QP_FAILED; // uAct, relax untouched, src/asif_robust.cpp:250-251
		}
		if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * a.ld + i] = (double)it;
	}
};

int launch_robust_data_point(const RbDev &z, double *pointC, hipStream_t stream)
{
	hipLaunchKernelGGL(robust_data_point_kernel, dim3(1), dim3(1), 0, stream, z, pointC);
	return (int)hipGetLastError();
}

int launch_robust_data(const RbDev &z, const asif_hip_solver &S0, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream)
{
	if (a.B <= 0) return 0;
	if (z.npSSmax < 1 || z.npSSmax > kRbMaxRows) return ASIF_HIP_EUNSUPPORTED;
	if (assemble_only) {
		hipLaunchKernelGGL(robust_data_rows_kernel, dim3(grid_for(a.B, 1, 64)), dim3(64), 0, stream, z, a);
		return (int)hipGetLastError();
	}
	const RobustDataPolicy p = {a.B, z, a};
	// (Near the boundary h is ~1e-4 against |Lgh| ~ 3e-2: the kept rows are nearly parallel in (u, delta).  This is
	// the workload that made the working-set solve of admm_small.hpp escalate its penalty.)
	const asif_hip_solver &S = S0;
	int G = S.lanes_per_qp;
	if (G == 0) G = a.B >= 32768 ? 2 : (a.B >= 16384 ? 4 : 8);
	switch (G) {
	case 2: return launch_policy<2, 2 * kRbMaxRows, 2>(S, p, stream);
	case 4: return launch_policy<2, 2 * kRbMaxRows, 4>(S, p, stream);
	case 8: return launch_policy<2, 2 * kRbMaxRows, 8>(S, p, stream);
	default: return ASIF_HIP_EINVAL;
	}
}

} // namespace asif
