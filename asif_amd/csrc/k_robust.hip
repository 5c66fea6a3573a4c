// k_robust.hip -- robust explicit filter: ASIFrobust::filter, src/asif_robust.cpp:218-252.
//
// Stage 1  robust_rows_kernel, one instance per lane: ASIFrobust::updateConstraints (:275-367).
//   The reference lifts state and safety-set gradient to affine forms (:282-284,322-324), runs the model
//   dynamics in affine arithmetic and forms Lfh = Dh f, Lgh = Dh g with its matmul (include/asif_utils.h:22-62
//   instantiated on AAF); pendulum_point_lie() below is that computation at a point state with the zero terms
//   dropped.  The intervals are scattered into the fixed sparsity pattern initialize() lays down (:103-133):
//   per safety row s
//       ineq  h_s d + lo(Lgh) l+_0 - hi(Lgh) l-_0 + lo(Lfh) l+_1 - hi(Lfh) l-_1 >= 0
//       eq    -u + l+_0 - l-_0 = 0
//       eq    l+_1 - l-_1 = 1 ,   l >= 0.
//   asif_hip_assemble_batch hands out exactly these nc x nv = 3N x (2+4N) rows.
// Stage 2  The multipliers are eliminated exactly before the solve (nu == 1): for fixed (u, d) the best
//   l of row group s gives   h_s d + min(lo(Lgh) u, hi(Lgh) u) + lo(Lfh) >= 0,   i.e. the two plain rows
//   [lo(Lgh), h_s] and [hi(Lgh), h_s] with right-hand side -lo(Lfh).  (u*, d*) of the (2+4N)-variable QP
//   the reference hands to OSQP is the optimum of this 2-variable, 2N-row QP -- H is zero on every
//   multiplier (:89-90) and (u, d) is all filter() reads back (:243-248).  The reduced rows are what
//   stage 1 stages in filter mode; qp_policy_kernel<2,16,2> solves them (rows beyond 2N are inert).
#include "qp_kernel.hpp"

namespace asif {

// Interval Lie derivatives of the robust pendulum at a POINT state, src/asif_robust.cpp:282-337 on
// examples/InvertedPendulum_Robust.cpp:62-69 (f = (x1, sin x0), g = (0, [pMin,pMax])).
//
// The reference lifts x and every entry of Dh to affine forms with ZERO radius, so along the whole
// computation the only noise symbol with a non-zero coefficient is the one of g[1]; every other
// coefficient libaffa carries is an exact zero (a.c*0, 0*rad, rad*0) that contributes |0| to the radius.
// Written out with the same roundings (affine_dev.hpp evaluates the identical expressions generically and
// is what the realizable filter's table kernel still uses):
//   f[0] = AAF(interval(x1))             centre (x1+x1)/2 = x1
//   f[1] = sin(AAF(interval(x0)))        width 0 < 1e-10 -> AAF(interval(t,t)), t = sin(x0*0.5 + x0*0.5)
//   Lfh_s = (0 + Dh_s0*f0) + Dh_s1*f1    radius 0                      -> lo = hi
//   Lgh_s = (0 + Dh_s0*0) + Dh_s1*gc     radius |Dh_s1*gr|, gc = (pMax+pMin)/2, gr = (pMax-pMin)/2
// The affine forms of the previous implementation lived in per-lane scratch memory (5x the algorithmic HBM
// traffic by the PMC counters); this needs a dozen flops and no memory.
__device__ __forceinline__ static void pendulum_point_lie(const DevOptions &o, const double (&x)[2],
                                                          const double (&Dh)[ASIF_HIP_MAX_HALFPLANES * 2], int N,
                                                          double (&fl)[ASIF_HIP_MAX_HALFPLANES],
                                                          double (&fh)[ASIF_HIP_MAX_HALFPLANES],
                                                          double (&gl)[ASIF_HIP_MAX_HALFPLANES],
                                                          double (&gh)[ASIF_HIP_MAX_HALFPLANES])
{
#pragma clang fp contract(off)
	const double f0 = (x[1] + x[1]) / 2;
	const double xs = x[0] * 0.5 + x[0] * 0.5;
	const double t = sin(xs);
	const double f1 = (t + t) / 2;
	const double gc = (o.pMax + o.pMin) / 2, gr = (o.pMax - o.pMin) / 2;
	for (int s = 0; s < N; s++) {
		const double d0 = (Dh[s] + Dh[s]) / 2, d1 = (Dh[s + N] + Dh[s + N]) / 2; // centres of AAF(interval(Dh))
		const double cf = (0.0 + d0 * f0) + d1 * f1;
		fl[s] = cf - 0.0;
		fh[s] = cf + 0.0;
		const double cg = (0.0 + d0 * 0.0) + d1 * gc;
		const double r = fabs(d1 * gr);
		gl[s] = cg - r;
		gh[s] = cg + r;
	}
}

// asif_hip_assemble_batch: the full rows the reference hands to updateA/updateb
__global__ __launch_bounds__(64) void robust_rows_kernel(DevOptions o, FilterArgs a)
{
	using M = InvertedPendulumRobust;
	constexpr int NX = M::NX, NU = M::NU, MAXNP = M::MAXNP;
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.B) return;
	const int64_t ld = a.ld;
	const int N = o.nHalfPlanes;
	const int nv = NU + 1 + N * 2 * (NU + 1), nc = N * (NU + 2);

	double x[NX], h[MAXNP], Dh[MAXNP * NX];
#pragma unroll
	for (int k = 0; k < NX; k++) x[k] = a.x[k * ld + i];
	M::safetySet(o, x, h, Dh);
	double fl[MAXNP], fh[MAXNP], gl[MAXNP], gh[MAXNP];
	pendulum_point_lie(o, x, Dh, N, fl, fh, gl, gh);
	// fixed structure (src/asif_robust.cpp:103-133) + interval entries (:340-358)
	for (int e = 0; e < nc * nv; e++) a.A[(int64_t)e * ld + i] = 0.0;
	for (int r = 0; r < nc; r++) a.b[(int64_t)r * ld + i] = 0.0;
	int iCol = NU + 1;
	for (int s = 0; s < N; s++) {
		const int iRow = s * (NU + 2);
		a.A[(int64_t)(iRow + NU * nc) * ld + i] = h[s];
		a.A[(int64_t)(iRow + (iCol + 0) * nc) * ld + i] = gl[s];
		a.A[(int64_t)(iRow + (iCol + (NU + 1) + 0) * nc) * ld + i] = -gh[s];
		a.A[(int64_t)(iRow + (iCol + NU) * nc) * ld + i] = fl[s];
		a.A[(int64_t)(iRow + (iCol + (NU + 1) + NU) * nc) * ld + i] = -fh[s];
		a.A[(int64_t)((iRow + 1) + 0 * nc) * ld + i] = -1.0; // -u
		for (int q = 0; q < NU + 1; q++) {
			a.A[(int64_t)((iRow + 1 + q) + (iCol + q) * nc) * ld + i] = 1.0;
			a.A[(int64_t)((iRow + 1 + q) + (iCol + NU + 1 + q) * nc) * ld + i] = -1.0;
		}
		a.b[(int64_t)(iRow + NU + 1) * ld + i] = 1.0;
		iCol += 2 * (NU + 1);
	}
	if (a.code) a.code[i] = 1;
}

// What ASIFrobust::filter reads of its options, and of its arguments: the policy's whole argument block (124 or 188
// bytes + 72, next to the solver's 112) instead of the 2 KB DevOptions + FilterArgs.  With the big structure the
// kernel fetched its options piecemeal -- 21 scalar loads in nine dependent groups, each a scalar-cache round trip for
// a wave that has nothing else to run -- because the scalar file cannot hold what it does not know it will need; this
// block arrives in three loads issued together.  MAXN: half-planes carried (4 when the set has no more: BASELINE's C5).
template <int MAXN>
struct RobustOpts {
	double halfPlanes[2 * MAXN]; // {a0, a1}: 1 - a.x >= 0
	double pMin, pMax, relaxCost, relaxLb, lb0, ub0, inf;
	int nHalfPlanes;
};
struct RobustArgs {
	int64_t ld;
	const double *x, *udes;
	double *uact, *relax;
	int32_t *rc;
	double *diag;
	int ndiag;
};

// examples/InvertedPendulum_Robust.cpp:53-61 (InvertedPendulumRobust::safetySet) and pendulum_point_lie above on the
// small block, every half-plane slot with a compile-time index (nothing here is addressed at run time: no scratch)
template <int MAXN>
__device__ __forceinline__ static void robust_rows_small(const RobustOpts<MAXN> &e, const double (&x)[2], double (&h)[MAXN],
                                                         double (&fl)[MAXN], double (&gl)[MAXN], double (&gh)[MAXN])
{
	double d0[MAXN], d1[MAXN];
#pragma unroll
	for (int s = 0; s < MAXN; s++) {
		h[s] = 1. - e.halfPlanes[2 * s] * x[0] - e.halfPlanes[2 * s + 1] * x[1];
		d0[s] = -e.halfPlanes[2 * s];
		d1[s] = -e.halfPlanes[2 * s + 1];
	}
	{
#pragma clang fp contract(off)
		const double f0 = (x[1] + x[1]) / 2;
		const double xs = x[0] * 0.5 + x[0] * 0.5;
		const double t = sin(xs);
		const double f1 = (t + t) / 2;
		const double gc = (e.pMax + e.pMin) / 2, gr = (e.pMax - e.pMin) / 2;
#pragma unroll
		for (int s = 0; s < MAXN; s++) {
			const double c0 = (d0[s] + d0[s]) / 2, c1 = (d1[s] + d1[s]) / 2; // centres of AAF(interval(Dh))
			const double cf = (0.0 + c0 * f0) + c1 * f1;
			fl[s] = cf - 0.0;
			const double cg = (0.0 + c0 * 0.0) + c1 * gc;
			const double r = fabs(c1 * gr);
			gl[s] = cg - r;
			gh[s] = cg + r;
		}
	}
}

template <int MAXN>
struct RobustPolicy {
	int64_t B;
	RobustOpts<MAXN> e;
	RobustArgs a;

	// Fused: every lane of the group assembles the rows of its instance from the state (a dozen flops per
	// half-plane, cheaper than staging them through HBM) and keeps its share.  Reduced row r = 2s + p is
	// [p ? hi(Lgh_s) : lo(Lgh_s), h_s] >= -lo(Lfh_s); lane g owns rows g, g+G, ...: s = (G/2) k + (g >> 1), p = g & 1.
	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
		// NC = 2 MAXN: the whole row budget, or half of it when the set has no more than half the half-planes (the
		// launcher's choice: BASELINE's C5 has four of eight, and rows that are padding cost what real ones cost)
		static_assert(NV == 2 && NC == 2 * MAXN && (G == 2 || G == 4 || G == 8), "reduced robust QP");
		constexpr int RPL = (NC + G - 1) / G, H = G / 2;
		const int N = e.nHalfPlanes;
		double x[2], h[MAXN], fl[MAXN], gl[MAXN], gh[MAXN];
		x[0] = a.x[i];
		x[1] = a.x[a.ld + i];
		robust_rows_small<MAXN>(e, x, h, fl, gl, gh);
		const bool hiRow = (g & 1) != 0;
		const int sub = g >> 1;
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			double lg = 0.0, hs = 0.0, rhs = -1e20;
#pragma unroll
			for (int q = 0; q < H; q++) {
				const int s = H * k + q; // compile-time
				if (s < MAXN) {
					const bool pick = (q == sub) && (s < N);
					lg = pick ? (hiRow ? gh[s] : gl[s]) : lg;
					hs = pick ? h[s] : hs;
					rhs = pick ? -fl[s] : rhs;
				}
			}
			qp.A[k][0] = lg;
			qp.A[k][1] = hs;
			qp.b[k] = rhs;
			qp.eq[k] = false;
		}
		// src/asif_robust.cpp:89-101,140-142 restricted to (u, delta)
		qp.Hd[0] = 1.0;
		qp.Hd[1] = e.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * e.relaxCost * e.relaxLb;
		qp.lb[0] = e.lb0;
		qp.lb[1] = e.relaxLb;
		qp.ub[0] = e.ub0;
		qp.ub[1] = e.inf;
	}
	template <int NV>
	__device__ __forceinline__ void store(int64_t i, const double (&sol)[NV], int st, int it) const
	{
		if (st == kStatusSolved) {
			a.uact[i] = fmin(fmax(sol[0], e.lb0), e.ub0);
			a.relax[i] = sol[1];
			a.rc[i] = ASIF_HIP_RC_OK;
		} else {
			a.rc[i] = ASIF_HIP_RC_QP_FAILED; // uAct, relax untouched, src/asif_robust.cpp:250-251
		}
		if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * a.ld + i] = (double)it;
	}
};

template <int MAXN>
static RobustPolicy<MAXN> robust_policy(const DevOptions &o, const FilterArgs &a)
{
	RobustPolicy<MAXN> p = {};
	p.B = a.B;
	for (int k = 0; k < 2 * MAXN; k++) p.e.halfPlanes[k] = o.halfPlanes[k];
	p.e.pMin = o.pMin;
	p.e.pMax = o.pMax;
	p.e.relaxCost = o.relaxCost;
	p.e.relaxLb = o.relaxLb;
	p.e.lb0 = o.lb[0];
	p.e.ub0 = o.ub[0];
	p.e.inf = o.inf;
	p.e.nHalfPlanes = o.nHalfPlanes;
	p.a = {a.ld, a.x, a.udes, a.uact, a.relax, a.rc, a.diag, a.ndiag};
	return p;
}

int launch_robust_ip(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                     hipStream_t stream)
{
	if (a.B <= 0) return 0;
	if (assemble_only) {
		hipLaunchKernelGGL(robust_rows_kernel, dim3(grid_for(a.B, 1, 64)), dim3(64), 0, stream, o, a);
		return (int)hipGetLastError();
	}
	int G = S.lanes_per_qp;
	if (G == 0) G = a.B >= 32768 ? 2 : (a.B >= 16384 ? 4 : 8); // enough lane groups for one wave on every SIMD
	constexpr int MAXNP = ASIF_HIP_MAX_HALFPLANES;
	const bool half = o.nHalfPlanes <= MAXNP / 2; // every real row fits half the row budget
	const RobustPolicy<MAXNP / 2> ph = robust_policy<MAXNP / 2>(o, a);
	const RobustPolicy<MAXNP> pf = robust_policy<MAXNP>(o, a);
	switch (G) {
	// one Ruiz pass by default: these rows are well scaled and a second pass only costs finish rounds
	case 2: return half ? launch_policy<2, MAXNP, 2>(S, ph, stream, 1) : launch_policy<2, 2 * MAXNP, 2>(S, pf, stream, 1);
	case 4: return half ? launch_policy<2, MAXNP, 4>(S, ph, stream, 1) : launch_policy<2, 2 * MAXNP, 4>(S, pf, stream, 1);
	case 8: return half ? launch_policy<2, MAXNP, 8>(S, ph, stream, 1) : launch_policy<2, 2 * MAXNP, 8>(S, pf, stream, 1);
	default: return ASIF_HIP_EINVAL;
	}
}

} // namespace asif
