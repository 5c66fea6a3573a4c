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
//   Thread 0 also evaluates the state-independent pieces of the point-state dynamics (inv(m), K/m).
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
//   3. barrier rows (:530-600): midpoints of the point-state interval dynamics, Lfh/Lgh, right-hand side.
//   4. QP in (u, delta):  min (u-uDes)^2 + relaxCost delta^2  s.t.  Lgh_i u + delta >= b_i,
//      max(lb,uLo) <= u <= min(ub,uHi), 0 <= delta <= inf;  in-register ADMM + active-set finish.
//   5. inputSaturate, relax = {l+_0 of group 0, delta}, rc 1 / -1 / -2 (:324-351).
//   asif_hip_assemble_batch writes the full nc x nv rows the reference hands to updateA/updateb instead.
// The assembly arithmetic is compiled without FMA contraction: rows are bit-identical to the oracle's.
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

	// State-independent pieces of dynamicsAffine at a POINT state x = ([x0,x0], [x1,x1]), written to
	// pc[kRzPoint] by one thread at table-build time:  inv(m) = ci + am*eps_m + de*eps_new,  -F = nFc + nFr*eps_F,
	// mid(K/m).
	__device__ static void pointConstants(AfCtx &cx, const Af &m, const Af &K, const Af &F, double *pc)
	{
		Af im, nF, g1;
		af_inv(cx, m, im); // {eps_m, new}
		af_neg(F, nF);
		af_div(cx, K, m, g1);
		double lo, hi;
		af_convert(g1, lo, hi);
		pc[0] = im.c;
		pc[1] = im.v[0];
		pc[2] = im.v[1];
		pc[3] = af_rad(im);
		pc[4] = nF.c;
		pc[5] = nF.v[0];
		pc[6] = lo * 0.5 + hi * 0.5;
		pc[7] = 0.0;
	}

	// Midpoints (interval::mid() = lo*0.5 + hi*0.5) of dynamicsAffine at a point state, :533-553.  This is
	// the same sequence of roundings the affine forms go through (symbols in index order m, F, x1, new(-F*x1),
	// new(inv m), new(product); coefficient rules of aa_aafapprox.cpp:34-101), with the zero terms kept.
	__device__ __forceinline__ static void pointDynamicsMid(const double *pc, double x1, double (&f)[2], double (&g)[2])
	{
#pragma clang fp contract(off)
		const double ci = pc[0], am = pc[1], de = pc[2], ri = pc[3], nFc = pc[4], nFr = pc[5];
		const double cu = nFc * x1;  // centre of -F*x1
		const double uF = x1 * nFr;  // its coefficient on eps_F; eps_x1 gets nFc*0, the new symbol rad*0
		const double ux = nFc * 0.0;
		const double radu = ((0.0 + fabs(uF)) + fabs(ux)) + fabs(fabs(nFr) * 0.0);
		const double c = cu * ci;
		double r = 0.0;
		r += fabs(cu * am);
		r += fabs(ci * uF);
		r += fabs(ci * ux);
		r += fabs(ci * (fabs(nFr) * 0.0));
		r += fabs(cu * de);
		r += fabs(radu * ri);
		const double lo = c - r, hi = c + r;
		f[0] = x1 * 0.5 + x1 * 0.5;
		f[1] = lo * 0.5 + hi * 0.5;
		g[0] = 0.0;
		g[1] = pc[6];
	}
};

__global__ __launch_bounds__(64) void realizable_table_kernel(RzDev z, const double *vertices, const int32_t *fverts,
                                                              const double *normals, const int32_t *factive,
                                                              double *facetRec, double *table, double *pointC,
                                                              int32_t *overflow)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= z.nF * z.nA) return;
	const int i = t / z.nA, j = t % z.nA;
	AfCtx cx = {0u, false};
	Af m, K, F;
	af_interval(cx, m, z.mMin, z.mMax);
	af_interval(cx, K, z.Klo, z.Khi);
	af_interval(cx, F, z.Flo, z.Fhi);
	if (t == 0) {
		AfCtx c2 = cx;
		DoubleIntegratorSampled::pointConstants(c2, m, K, F, pointC);
	}
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
		// segment stored from its end with the smaller first coordinate: p(t) = va + t*(a0, s1*a1), t in [0,1]
		double *r = &facetRec[(size_t)i * kRzRec];
		const bool swap = v1[0] < v0[0];
		const double *va = swap ? v1 : v0, *vb = swap ? v0 : v1;
		r[0] = va[0];
		r[1] = va[1];
		r[2] = vb[0] - va[0];
		r[3] = fabs(vb[1] - va[1]);
		r[4] = (vb[1] >= va[1]) ? 1.0 : -1.0;
		r[5] = normals[i * 2 + 0];
		r[6] = normals[i * 2 + 1];
		r[7] = 0.0;
		r[8] = fmin(v0[0], v1[0]); // bounding box, :160-175
		r[9] = fmax(v0[0], v1[0]);
		r[10] = fmin(v0[1], v1[1]);
		r[11] = fmax(v0[1], v1[1]);
	}
}

constexpr int kRzMaxCrit = 8; // critical facets kept per instance (kernel_t::maxCriticalFacets <= 8)

// one row  a*u + c >= 0  folded into [lo, hi]
__device__ __forceinline__ void fold_row(double a, double c, double &lo, double &hi, bool &feasible)
{
	// the three-way chain  a > 0: lo = max(lo, -c/a);  a < 0: hi = min(hi, -c/a);  otherwise infeasible if c < 0  as
	// selects on ONE quotient: the chain compiled to seven branches per table row, ~45 cycles each for a lone wave
	const double q = -c / a;
	const bool pos = a > 0.0, neg = a < 0.0;
	lo = pos ? fmax(lo, q) : lo;
	hi = neg ? fmin(hi, q) : hi;
	feasible = feasible && !(!pos && !neg && c < 0.0);
}

// KB = barrier rows carried in registers (>= npSSmax) = rows of the in-register QP
// LDSREC: the seven scan fields of every facet record are staged in LDS (64 B per facet) by the block's wave and
// read back with wave-uniform ds_reads, which return in order and so can be kept several records deep in flight;
// the scalar-load form (used when the kernel polytope is too large for LDS) has to drain its whole queue at
// every wait (SMEM returns out of order) and spent ~0.2 us of load latency per facet per launch.
template <int KB, bool LDSREC>
__global__ __launch_bounds__(256) void realizable_filter_kernel(RzDev z, asif_hip_solver S, FilterArgs a,
                                                               bool assemble_only)
{
	constexpr int NV = 2, RPL = KB;
	extern __shared__ double srec[]; // [nF][8] when LDSREC
	if (LDSREC) {
		for (int k = threadIdx.x; k < z.nF * 8; k += blockDim.x) // (one to four waves per workgroup share the table)
			srec[k] = (k & 7) < 7 ? z.facetRec[(size_t)(k >> 3) * kRzRec + (k & 7)] : 0.0;
		__syncthreads();
	}
	const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t i = tid;
	const bool live = i < a.B;
	if (!live) i = a.B - 1;
	const int64_t ld = a.ld;
	const double x0 = a.x[i], x1 = a.x[ld + i];

	// ---- 1. facet scan
	double hmin[KB];
	int hidx[KB];
	unsigned long long critLo = 0ull, critHi = 0ull; // critical facet indexes, 16 bits each, in facet order
	int nCrit = 0;
	bool anyNeg = false;
#pragma unroll
	for (int q = 0; q < KB; q++) {
		hmin[q] = __builtin_huge_val();
		hidx[q] = -1;
	}
	{
#pragma clang fp contract(off)
		const double xm0 = x0 - z.unc[0], xp0 = x0 + z.unc[0], xm1 = x1 - z.unc[1], xp1 = x1 + z.unc[1];
		// One facet, branch-free.  `valid` is wave-uniform (false only for the padding slot of an odd facet count).
		auto facet = [&](const double (&r)[7], const int fiu, const bool valid) {
			// The facet index is wave-uniform (an SGPR).  Selecting it into per-lane registers as `c ? fi : old` makes
			// the compiler emit an exec-mask sequence per select (v_cmp -> s_and_saveexec -> masked v_mov from the
			// SGPR): three of them per facet, ~0.15 us per facet per launch in this one-wave-per-SIMD loop.  A copy
			// in a VGPR turns them into plain v_cndmask.
			int fi;
			asm("v_mov_b32 %0, %1" : "=v"(fi) : "s"(fiu));
			double h = 1.;
			h -= r[5] * x0;
			h -= r[6] * x1;
			h = valid ? h : __builtin_huge_val();
			anyNeg = anyNeg | (h < 0.);
			if constexpr (KB == 2) {
				// two smallest, strict < keeps the lower facet index on ties: one compare-exchange against each slot,
				// written as independent selects (a nested ?: here compiles to exec-mask branches)
				const bool c1 = h < hmin[0];
				const int hiI = c1 ? hidx[0] : fi;     // index travelling on to slot 1
				const double hiH = fmax(hmin[0], h);     // its margin
				hidx[0] = c1 ? fi : hidx[0];
				hmin[0] = fmin(hmin[0], h);
				const bool c2 = hiH < hmin[1];
				hidx[1] = c2 ? hiI : hidx[1];
				hmin[1] = fmin(hmin[1], hiH);
			} else {
				double hv = h;
				int hi_ = fi;
#pragma unroll
				for (int q = 0; q < KB; q++) { // sorted insert
					const bool lt = hv < hmin[q];
					const double tv = hmin[q];
					const int ti = hidx[q];
					hmin[q] = lt ? hv : tv;
					hidx[q] = lt ? hi_ : ti;
					hv = lt ? tv : hv;
					hi_ = lt ? ti : hi_;
				}
			}
			// Does the facet touch the box [x-unc, x+unc]?  With p(t) = va + t*(a0, s1*a1): exists t in [0,1] with
			// A0 <= t*a0 <= B0 and L1 <= t*a1 <= H1.  Lower bounds {0, A0/a0, L1/a1} against upper bounds
			// {1, B0/a0, H1/a1}, cross-multiplied (a0, a1 >= 0; a zero extent degenerates correctly): no divide,
			// no branch.  The first four comparisons are the reference's bounding-box prefilter (:400-409).
			const double A0 = xm0 - r[0], B0 = xp0 - r[0], A1 = xm1 - r[1], B1 = xp1 - r[1];
			const bool up = r[4] > 0.0; // wave-uniform
			const double L1 = up ? A1 : -B1, H1 = up ? B1 : -A1;
			const bool touch = (A0 <= r[2]) & (B0 >= 0.0) & (L1 <= r[3]) & (H1 >= 0.0) & (A0 * r[3] <= H1 * r[2]) &
			                   (L1 * r[2] <= B0 * r[3]);
			// first maxCriticalFacets in facet order (:436-438), appended as 16-bit fields
			const bool app = touch & (nCrit < z.maxCrit) & valid;
			const unsigned long long v = app ? ((unsigned long long)fi << (16 * (nCrit & 3))) : 0ull;
			critLo |= (nCrit < 4) ? v : 0ull;
			critHi |= (nCrit < 4) ? 0ull : v;
			nCrit += app ? 1 : 0;
		};
		// Records come through scalar loads (wave-uniform addresses).  Scalar loads return out of order, so a
		// wait covers everything outstanding; four records are requested per wait to amortise the latency.
		const int nF4 = z.nF & ~3;
		for (int fi = 0; fi < nF4; fi += 4) {
			double r[4][7];
#pragma unroll
			for (int u = 0; u < 4; u++)
#pragma unroll
				for (int k = 0; k < 7; k++)
					r[u][k] = LDSREC ? srec[(fi + u) * 8 + k] : z.facetRec[(size_t)(fi + u) * kRzRec + k];
#pragma unroll
			for (int u = 0; u < 4; u++) facet(r[u], fi + u, true);
		}
		for (int fi = nF4; fi < z.nF; fi++) {
			double r[7];
#pragma unroll
			for (int k = 0; k < 7; k++) r[k] = LDSREC ? srec[fi * 8 + k] : z.facetRec[(size_t)fi * kRzRec + k];
			facet(r, fi, true);
		}
	}
	int crit[kRzMaxCrit];
#pragma unroll
	for (int q = 0; q < kRzMaxCrit; q++)
		crit[q] = (q < nCrit) ? (int)(((q < 4 ? critLo : critHi) >> (16 * (q & 3))) & 0xFFFFull) : -1;
	const int code = (nCrit == 0 && anyNeg) ? -1 : 1; // :602-605

	// ---- 3. barrier rows (:533-599)
	double Lgh[KB], bb[KB];
	{
#pragma clang fp contract(off)
		double f[2], g[2];
		DoubleIntegratorSampled::pointDynamicsMid(z.pointC, x1, f, g);
#pragma unroll
		for (int q = 0; q < KB; q++) {
			const int fi = (q < z.npSSmax) ? hidx[q] : 0;
			const double *r = z.facetRec + (size_t)fi * kRzRec;
			const double n0 = r[5], n1 = r[6];
			double lf = 0.0, lg = 0.0;
			lf += -n0 * f[0];
			lf += -n1 * f[1];
			lg += -n0 * g[0];
			lg += -n1 * g[1];
			Lgh[q] = lg;
			bb[q] = -lf - z.relaxDes * (hmin[q] - z.relaxOffset);
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
		for (int q = 0; q < KB; q++)
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
			admm.solve(qp, S, sol, status, iters, false, S.polish != 1);
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
		for (int q = 0; q < KB; q++)
			if (q < z.npSSmax) a.diag[(int64_t)(1 + z.maxCrit + q) * ld + i] = (double)hidx[q];
	}
}

int launch_realizable_tables(const RzDev &z, const double *vertices, const int32_t *fverts, const double *normals,
                             const int32_t *factive, double *facetRec, double *table, double *pointC,
                             int32_t *overflow, hipStream_t stream)
{
	const int n = z.nF * z.nA;
	hipLaunchKernelGGL(realizable_table_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, z, vertices, fverts, normals,
	                   factive, facetRec, table, pointC, overflow);
	return (int)hipGetLastError();
}

int launch_realizable(const RzDev &z, const asif_hip_solver &S0, const FilterArgs &a, bool assemble_only,
                      hipStream_t stream)
{
	if (a.B <= 0) return 0;
	const asif_hip_solver S = resolve_scaling(S0, 2);
	if (z.maxCrit > kRzMaxCrit || z.npSSmax > 4) return ASIF_HIP_EUNSUPPORTED;
	const int nw = waves_per_workgroup(grid_for(a.B, 1, 64)); // independent waves (launchers.hpp)
	const dim3 grid(grid_for(a.B, 1, 64 * nw)), block(64 * nw);
	const size_t recBytes = (size_t)z.nF * 8 * sizeof(double);
	const bool lds = recBytes <= 48 * 1024;
	if (z.npSSmax <= 2) {
		if (lds) hipLaunchKernelGGL((realizable_filter_kernel<2, true>), grid, block, recBytes, stream, z, S, a, assemble_only);
		else hipLaunchKernelGGL((realizable_filter_kernel<2, false>), grid, block, 0, stream, z, S, a, assemble_only);
	} else {
		if (lds) hipLaunchKernelGGL((realizable_filter_kernel<4, true>), grid, block, recBytes, stream, z, S, a, assemble_only);
		else hipLaunchKernelGGL((realizable_filter_kernel<4, false>), grid, block, 0, stream, z, S, a, assemble_only);
	}
	return (int)hipGetLastError();
}

} // namespace asif
