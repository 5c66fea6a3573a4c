// k_realizable.hip -- realizable filter on a polytopic sampled-data kernel: ASIFrealizable
// (src/asif_realizable.cpp), device model = the sampled double integrator with interval parameters
// (examples/DoubleIntegrator_RealizableSampled.cpp:16-62).  nx == 2 (facets are segments), nu == 1.
//
// realizable_table_kernel (once per handle, at create / updateOptions): one thread per (facet, active
//   constraint) pair.  Restates what initialize() and updateConstraints() evaluate in affine arithmetic
//   and that does not depend on the state: xFaceInt = lam*v0 + (1-lam)*v1, lam = [0,1] (:137-157), the
//   bounding box (:160-175), dynamics_(xFaceInt) and  Lfh = sum f_k*DhInt_k,  Lgh = sum g_k*DhInt_k  with
//   Dh = -normal of the active constraint (:465-506), converted to intervals.  Symbols are created in the
//   reference's order (globals m, K, F; facet; per-call), so the table is bit-identical to libaffa's.
//
// realizable_filter_kernel (per batch, fused, one instance per lane):
//   1. facet scan (:386-442), facet data read through wave-uniform (scalar) loads: h_i = 1 - n_i.x, the
//      npSSmax smallest h (:562-571), "any h < 0" (:602), bounding-box prefilter and the per-facet
//      feasibility problem of facetSolver_ (:381-384,411-427).  That QP only answers "does the facet touch
//      the uncertainty box around x"; for a segment this is a closed-form clip, evaluated exactly.
//   2. rows of the critical facets (:509-527) gathered from the table.  The multipliers of each row group
//      are eliminated exactly (nu == 1, same argument as k_robust.hip): group s is satisfiable for a given
//      u iff  lo(Lgh) u + lo(Lfh) >= 0  and  hi(Lgh) u + lo(Lfh) >= 0,  so the 3*npSS rows and 4*npSS
//      multipliers collapse into an interval [uLo, uHi] for u.
//   3. barrier rows (:530-600): point-state dynamics in affine arithmetic, midpoints, Lfh/Lgh, right-hand side.
//   4. QP in (u, delta):  min (u-uDes)^2 + relaxCost delta^2  s.t.  Lgh_i u + delta >= b_i,
//      max(lb,uLo) <= u <= min(ub,uHi), 0 <= delta <= inf;  in-register ADMM + active-set finish.
//   5. inputSaturate, relax = {l+_0 of group 0, delta}, rc 1 / -1 / -2 (:324-351).
//   asif_hip_assemble_batch writes the full nc x nv rows the reference hands to updateA/updateb instead.
#include "admm_small.hpp"
#include "affine_dev.hpp"
#include "launchers.hpp"

namespace asif {

// examples/DoubleIntegrator_RealizableSampled.cpp:47-54:  f = (x1, -F*x1/m),  g = (0, K/m)
struct DoubleIntegratorSampled {
	__device__ static void dynamicsAffine(AfCtx &cx, const Af &m, const Af &K, const Af &F, const Af (&x)[2], Af (&f)[2],
	                                      Af (&g)[2])
	{
		Af t, u;
		f[0] = x[1];
		af_neg(F, t);
		af_mul(cx, t, x[1], u);
		af_div(cx, u, m, f[1]);
		af_const(g[0], 0.);
		af_div(cx, K, m, g[1]);
	}
};

__global__ __launch_bounds__(64) void realizable_table_kernel(RzDev z, const double *vertices, const int32_t *fverts,
                                                              const double *normals, const int32_t *factive,
                                                              double *facetRec, double *table, int32_t *overflow)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= z.nF * z.nA) return;
	const int i = t / z.nA, j = t % z.nA;
	AfCtx cx = {0u, false};
	Af m, K, F;
	af_interval(cx, m, z.mMin, z.mMax);
	af_interval(cx, K, z.Klo, z.Khi);
	af_interval(cx, F, z.Flo, z.Fhi);
	const double *v0 = &vertices[fverts[i * 2 + 0] * 2], *v1 = &vertices[fverts[i * 2 + 1] * 2];
	Af xf[2], lam, one, oml, a, b2;
	af_const(xf[0], v0[0]);
	af_const(xf[1], v0[1]);
	af_interval(cx, lam, 0., 1.);
	for (int k = 0; k < 2; k++) {
		af_mul(cx, lam, xf[k], a);
		af_const(one, 1.);
		af_sub(cx, one, lam, oml);
		af_scale(oml, v1[k], b2);
		af_add(cx, a, b2, xf[k]);
	}
	const int src = factive[i * z.nA + j];
	Af Dh[2], f[2], g[2], Lfh, Lgh, tt;
	for (int k = 0; k < 2; k++) {
		const double d = -normals[src * 2 + k];
		af_interval(cx, Dh[k], d, d);
	}
	DoubleIntegratorSampled::dynamicsAffine(cx, m, K, F, xf, f, g);
	af_const(Lfh, 0.);
	for (int k = 0; k < 2; k++) {
		af_mul(cx, f[k], Dh[k], tt);
		af_add(cx, Lfh, tt, Lfh);
	}
	af_const(Lgh, 0.);
	for (int k = 0; k < 2; k++) {
		af_mul(cx, g[k], Dh[k], tt);
		af_add(cx, Lgh, tt, Lgh);
	}
	double *o = &table[(size_t)t * 4];
	af_convert(Lgh, o[0], o[1]);
	af_convert(Lfh, o[2], o[3]);
	if (cx.overflow) atomicAdd(overflow, 1);
	if (j == 0) {
		double *r = &facetRec[(size_t)i * kRzRec];
		r[0] = v0[0];
		r[1] = v0[1];
		r[2] = v1[0];
		r[3] = v1[1];
		r[4] = fmin(v0[0], v1[0]);
		r[5] = fmax(v0[0], v1[0]);
		r[6] = fmin(v0[1], v1[1]);
		r[7] = fmax(v0[1], v1[1]);
		r[8] = normals[i * 2 + 0];
		r[9] = normals[i * 2 + 1];
	}
}

constexpr int kRzMaxCrit = 8;  // critical facets kept per instance (kernel_t::maxCriticalFacets <= 8)
constexpr int kRzMaxBarrier = 4; // barrier rows (npSSmax <= 4) = rows of the in-register QP

// one row  a*u + c >= 0  folded into [lo, hi]
__device__ __forceinline__ void fold_row(double a, double c, double &lo, double &hi, bool &feasible)
{
	if (a > 0.0) lo = fmax(lo, -c / a);
	else if (a < 0.0) hi = fmin(hi, -c / a);
	else if (c < 0.0) feasible = false;
}

__global__ __launch_bounds__(64) void realizable_filter_kernel(RzDev z, asif_hip_solver S, FilterArgs a,
                                                               bool assemble_only)
{
	constexpr int NV = 2, RPL = kRzMaxBarrier;
	const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t i = tid;
	const bool live = i < a.B;
	if (!live) i = a.B - 1;
	const int64_t ld = a.ld;
	const double x0 = a.x[i], x1 = a.x[ld + i];

	// ---- 1. facet scan
	double hmin[kRzMaxBarrier];
	int hidx[kRzMaxBarrier], crit[kRzMaxCrit];
	int nCrit = 0;
	bool anyNeg = false;
#pragma unroll
	for (int q = 0; q < kRzMaxBarrier; q++) {
		hmin[q] = __builtin_huge_val();
		hidx[q] = -1;
	}
#pragma unroll
	for (int q = 0; q < kRzMaxCrit; q++) crit[q] = -1;
	{
#pragma clang fp contract(off)
		for (int fi = 0; fi < z.nF; fi++) {
			const double *r = z.facetRec + (size_t)fi * kRzRec; // wave-uniform address
			double h = 1.;
			h -= r[8] * x0;
			h -= r[9] * x1;
			anyNeg = anyNeg || (h < 0.);
			double hv = h;
			int hi_ = fi;
#pragma unroll
			for (int q = 0; q < kRzMaxBarrier; q++) // sorted insert; strict < keeps the lower facet index on ties
				if (hv < hmin[q]) {
					const double tv = hmin[q];
					const int ti = hidx[q];
					hmin[q] = hv;
					hidx[q] = hi_;
					hv = tv;
					hi_ = ti;
				}
			if (nCrit < z.maxCrit) {
				const bool potential = !(x0 < r[4] - z.unc[0] || x0 > r[5] + z.unc[0] || x1 < r[6] - z.unc[1] ||
				                         x1 > r[7] + z.unc[1]);
				if (potential) {
					// exists t in [0,1]: | t v0 + (1-t) v1 - x | <= unc  (componentwise)
					double tlo = 0.0, thi = 1.0;
					bool ok = true;
#pragma unroll
					for (int k = 0; k < 2; k++) {
						const double xk = k ? x1 : x0;
						const double d = r[k] - r[2 + k];
						const double lo = xk - z.unc[k] - r[2 + k], hi = xk + z.unc[k] - r[2 + k];
						if (d > 0.0) {
							tlo = fmax(tlo, lo / d);
							thi = fmin(thi, hi / d);
						} else if (d < 0.0) {
							tlo = fmax(tlo, hi / d);
							thi = fmin(thi, lo / d);
						} else if (lo > 0.0 || hi < 0.0) ok = false;
					}
					if (ok && tlo <= thi) {
#pragma unroll
						for (int q = 0; q < kRzMaxCrit; q++)
							if (q == nCrit) crit[q] = fi;
						nCrit++;
					}
				}
			}
		}
	}
	const int code = (nCrit == 0 && anyNeg) ? -1 : 1; // :602-605

	// ---- 3. barrier rows: point-state dynamics in affine arithmetic, midpoints (:533-553)
	double Lgh[kRzMaxBarrier], bb[kRzMaxBarrier];
	{
#pragma clang fp contract(off)
		AfCtx cx = {0u, false};
		Af m, K, F, xI[2], fI[2], gI[2];
		af_interval(cx, m, z.mMin, z.mMax);
		af_interval(cx, K, z.Klo, z.Khi);
		af_interval(cx, F, z.Flo, z.Fhi);
		af_interval(cx, xI[0], x0, x0);
		af_interval(cx, xI[1], x1, x1);
		DoubleIntegratorSampled::dynamicsAffine(cx, m, K, F, xI, fI, gI);
		double f[2], g[2], lo, hi;
#pragma unroll
		for (int k = 0; k < 2; k++) {
			af_convert(fI[k], lo, hi);
			f[k] = lo * 0.5 + hi * 0.5;
			af_convert(gI[k], lo, hi);
			g[k] = lo * 0.5 + hi * 0.5;
		}
#pragma unroll
		for (int q = 0; q < kRzMaxBarrier; q++) {
			const int fi = (q < z.npSSmax) ? hidx[q] : 0;
			const double *r = z.facetRec + (size_t)fi * kRzRec;
			const double hq = hmin[q];
			double lf = 0.0, lg = 0.0;
			lf += -r[8] * f[0];
			lf += -r[9] * f[1];
			lg += -r[8] * g[0];
			lg += -r[9] * g[1];
			Lgh[q] = lg;
			bb[q] = -lf - z.relaxDes * (hq - z.relaxOffset);
		}
	}

	if (assemble_only) {
		if (!live) return;
		const int nc = z.nc, nv = z.nv;
		for (int e = 0; e < nc * nv; e++) a.A[(int64_t)e * ld + i] = 0.0;
		for (int r = 0; r < nc; r++) a.b[(int64_t)r * ld + i] = 0.0;
		for (int s = 0; s < z.npSS; s++) {
			const int iRow = 3 * s, col = 1 + 4 * s;
			const int c = s / z.nA, j = s % z.nA;
			double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
			if (c < nCrit) {
				int fc = 0;
#pragma unroll
				for (int q = 0; q < kRzMaxCrit; q++)
					if (q == c) fc = crit[q];
				const double *t = z.table + ((size_t)fc * z.nA + j) * 4;
				t0 = t[0];
				t1 = t[1];
				t2 = t[2];
				t3 = t[3];
			}
			a.A[(int64_t)(iRow + (col + 0) * nc) * ld + i] = t0;
			a.A[(int64_t)(iRow + (col + 2) * nc) * ld + i] = -t1;
			a.A[(int64_t)(iRow + (col + 1) * nc) * ld + i] = t2;
			a.A[(int64_t)(iRow + (col + 3) * nc) * ld + i] = -t3;
			a.A[(int64_t)((iRow + 1) + 0 * nc) * ld + i] = -1.0;
			a.A[(int64_t)((iRow + 1) + (col + 0) * nc) * ld + i] = 1.0;
			a.A[(int64_t)((iRow + 1) + (col + 2) * nc) * ld + i] = -1.0;
			a.A[(int64_t)((iRow + 2) + (col + 1) * nc) * ld + i] = 1.0;
			a.A[(int64_t)((iRow + 2) + (col + 3) * nc) * ld + i] = -1.0;
			a.b[(int64_t)(iRow + 2) * ld + i] = 1.0;
		}
#pragma unroll
		for (int q = 0; q < kRzMaxBarrier; q++)
			if (q < z.npSSmax) {
				const int row = 3 * z.npSS + q;
				a.A[(int64_t)(row + 0 * nc) * ld + i] = Lgh[q];
				a.A[(int64_t)(row + (nv - 1) * nc) * ld + i] = 1.0;
				a.b[(int64_t)row * ld + i] = bb[q];
			}
		a.code[i] = code;
	} else {
		// ---- 2. critical rows folded into an interval for u
		double ulo = z.lb, uhi = z.ub;
		bool feasible = true;
#pragma unroll
		for (int c = 0; c < kRzMaxCrit; c++)
			if (c < nCrit) {
				for (int j = 0; j < z.nA; j++) {
					const double *t = z.table + ((size_t)crit[c] * z.nA + j) * 4;
					const double lo_g = t[0], hi_g = t[1], lo_f = t[2];
					fold_row(lo_g, lo_f, ulo, uhi, feasible);
					fold_row(hi_g, lo_f, ulo, uhi, feasible);
				}
			}
		feasible = feasible && (ulo <= uhi);
		const bool want = (code == 1) && feasible;
		const double uDes = a.udes[i];

		// ---- 4. QP in (u, delta); lanes without a problem solve a benign one (the solver is wave-uniform)
		QpLaneData<NV, RPL> qp;
		qp.Hd[0] = 1.0;
		qp.Hd[1] = z.relaxCost;
		qp.c[0] = -2.0 * uDes;
		qp.c[1] = 0.0;
		qp.lb[0] = want ? ulo : z.lb;
		qp.ub[0] = want ? uhi : z.ub;
		qp.lb[1] = 0.0;
		qp.ub[1] = z.inf;
#pragma unroll
		for (int q = 0; q < RPL; q++) {
			const bool row = want && q < z.npSSmax;
			qp.A[q][0] = row ? Lgh[q] : 0.0;
			qp.A[q][1] = row ? 1.0 : 0.0;
			qp.b[q] = row ? bb[q] : -1e20;
			qp.eq[q] = false;
		}
		double sol[NV];
		int status = kStatusSolved, iters = 0;
		if (z.npSSmax > 0) { // wave-uniform
			AdmmSmall<NV, RPL, 1> admm;
			admm.solve(qp, S, sol, status, iters);
		} else { // no barrier rows, no delta: the QP is a clip
			sol[0] = fmin(fmax(uDes, qp.lb[0]), qp.ub[0]);
			sol[1] = 0.0;
		}
		if (!live) return;
		// ---- 5. outputs (:340-351)
		int rc;
		if (code < 0) rc = ASIF_HIP_RC_OUTSIDE_KERNEL;
		else if (!feasible || status != kStatusSolved) rc = ASIF_HIP_RC_QP_FAILED;
		else {
			rc = ASIF_HIP_RC_OK;
			a.uact[i] = fmin(fmax(sol[0], z.lb), z.ub);
			a.relax[i] = fmax(sol[0], 0.0); // smallest feasible l+_0 of group 0 (not determined by the QP: H is 0 on it)
			a.relax[ld + i] = sol[1];
		}
		a.rc[i] = rc;
		if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * ld + i] = (double)iters;
	}
	if (a.diag) {
		a.diag[i] = (double)nCrit;
#pragma unroll
		for (int q = 0; q < kRzMaxCrit; q++)
			if (q < z.maxCrit) a.diag[(int64_t)(1 + q) * ld + i] = (double)crit[q];
#pragma unroll
		for (int q = 0; q < kRzMaxBarrier; q++)
			if (q < z.npSSmax) a.diag[(int64_t)(1 + z.maxCrit + q) * ld + i] = (double)hidx[q];
	}
}

int launch_realizable_tables(const RzDev &z, const double *vertices, const int32_t *fverts, const double *normals,
                             const int32_t *factive, double *facetRec, double *table, int32_t *overflow,
                             hipStream_t stream)
{
	const int n = z.nF * z.nA;
	hipLaunchKernelGGL(realizable_table_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, z, vertices, fverts, normals,
	                   factive, facetRec, table, overflow);
	return (int)hipGetLastError();
}

int launch_realizable(const RzDev &z, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                      hipStream_t stream)
{
	if (a.B <= 0) return 0;
	if (z.maxCrit > kRzMaxCrit || z.npSSmax > kRzMaxBarrier) return ASIF_HIP_EUNSUPPORTED;
	hipLaunchKernelGGL(realizable_filter_kernel, dim3(grid_for(a.B, 1, 64)), dim3(64), 0, stream, z, S, a,
	                   assemble_only);
	return (int)hipGetLastError();
}

} // namespace asif
