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

constexpr int kRobustRedRows = 2 * ASIF_HIP_MAX_HALFPLANES; // reduced QP: 2 rows per half-plane

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

struct RobustPolicy {
	int64_t B;
	DevOptions o;
	FilterArgs a;

	// Fused: every lane of the group assembles the rows of its instance from the state (a dozen flops per
	// half-plane, cheaper than staging them through HBM) and keeps its share.  Reduced row r = 2s + p is
	// [p ? hi(Lgh_s) : lo(Lgh_s), h_s] >= -lo(Lfh_s); lane g owns rows g, g+G, ...: s = (G/2) k + (g >> 1), p = g & 1.
	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
		// NC: kRobustRedRows, or half of it when the set has no more than half the half-planes (the launcher's choice:
		// BASELINE's C5 has four of eight, and rows that are padding cost what real ones cost)
		static_assert(NV == 2 && (NC == kRobustRedRows || NC == kRobustRedRows / 2) && (G == 2 || G == 4 || G == 8), "reduced robust QP");
		using M = InvertedPendulumRobust;
		constexpr int MAXNP = M::MAXNP, RPL = (NC + G - 1) / G, H = G / 2;
		const int N = o.nHalfPlanes;
		double x[2], h[MAXNP], Dh[MAXNP * 2], fl[MAXNP], fh[MAXNP], gl[MAXNP], gh[MAXNP];
		x[0] = a.x[i];
		x[1] = a.x[a.ld + i];
		M::safetySet(o, x, h, Dh);
		pendulum_point_lie(o, x, Dh, N, fl, fh, gl, gh);
		const bool hiRow = (g & 1) != 0;
		const int sub = g >> 1;
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			double lg = 0.0, hs = 0.0, rhs = -1e20;
#pragma unroll
			for (int q = 0; q < H; q++) {
				const int s = H * k + q; // compile-time
				if (s < MAXNP) {
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
		qp.Hd[1] = o.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * o.relaxCost * o.relaxLb;
		qp.lb[0] = o.lb[0];
		qp.lb[1] = o.relaxLb;
		qp.ub[0] = o.ub[0];
		qp.ub[1] = o.inf;
	}
	template <int NV>
	__device__ __forceinline__ void store(int64_t i, const double (&sol)[NV], int st, int it) const
	{
		if (st == kStatusSolved) {
			a.uact[i] = fmin(fmax(sol[0], o.lb[0]), o.ub[0]);
			a.relax[i] = sol[1];
			a.rc[i] = ASIF_HIP_RC_OK;
		} else {
			a.rc[i] = ASIF_HIP_RC_QP_FAILED; // uAct, relax untouched, src/asif_robust.cpp:250-251
		}
		if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * a.ld + i] = (double)it;
	}
};

int launch_robust_ip(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                     hipStream_t stream)
{
	if (a.B <= 0) return 0;
	if (assemble_only) {
		hipLaunchKernelGGL(robust_rows_kernel, dim3(grid_for(a.B, 1, 64)), dim3(64), 0, stream, o, a);
		return (int)hipGetLastError();
	}
	const RobustPolicy p = {a.B, o, a};
	int G = S.lanes_per_qp;
	if (G == 0) G = a.B >= 32768 ? 2 : (a.B >= 16384 ? 4 : 8); // enough lane groups for one wave on every SIMD
	const bool half = 2 * o.nHalfPlanes <= kRobustRedRows / 2; // every real row fits half the row budget
	switch (G) {
	// one Ruiz pass by default: these rows are well scaled and a second pass only costs finish rounds
	case 2: return half ? launch_policy<2, kRobustRedRows / 2, 2>(S, p, stream, 1) : launch_policy<2, kRobustRedRows, 2>(S, p, stream, 1);
	case 4: return half ? launch_policy<2, kRobustRedRows / 2, 4>(S, p, stream, 1) : launch_policy<2, kRobustRedRows, 4>(S, p, stream, 1);
	case 8: return half ? launch_policy<2, kRobustRedRows / 2, 8>(S, p, stream, 1) : launch_policy<2, kRobustRedRows, 8>(S, p, stream, 1);
	default: return ASIF_HIP_EINVAL;
	}
}

} // namespace asif
