// k_robust.hip -- robust explicit filter: ASIFrobust::filter, src/asif_robust.cpp:218-252.
//
// Stage 1  robust_rows_kernel, one instance per lane: ASIFrobust::updateConstraints (:275-367).
//   State and safety-set gradient are lifted to affine forms (:282-284,322-324), the model dynamics
//   run in affine arithmetic, Lfh = Dh f and Lgh = Dh g are formed with the reference's matmul order
//   (include/asif_utils.h:22-62 instantiated on AAF), converted to intervals and scattered into the
//   fixed sparsity pattern initialize() lays down (:103-133): per safety row s
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
#include "affine_dev.hpp"
#include "qp_kernel.hpp"

namespace asif {

// examples/InvertedPendulum_Robust.cpp:62-69: f = (x1, sin x0), g = (0, [pMin,pMax]).
// Symbol creation order is the statement order: sin's symbol, then g[1]'s.
__device__ static void pendulum_dynamics_affine(const DevOptions &o, AfCtx &cx, const Af (&x)[2], Af (&f)[2], Af (&g)[2])
{
	f[0] = x[1];
	af_sin(cx, x[0], f[1]);
	af_const(g[0], 0.);
	af_interval(cx, g[1], o.pMin, o.pMax);
}

constexpr int kRobustRedRows = 2 * ASIF_HIP_MAX_HALFPLANES; // reduced QP: 2 rows per half-plane

__global__ __launch_bounds__(64) void robust_rows_kernel(DevOptions o, FilterArgs a, bool reduced)
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

	AfCtx cx = {0u, false};
	Af xI[NX], f[NX], g[NX * NU];
	for (int k = 0; k < NX; k++) af_interval(cx, xI[k], x[k], x[k]);
	for (int k = 0; k < NX; k++) af_const(f[k], 0.0);
	for (int k = 0; k < NX * NU; k++) af_const(g[k], 0.0);
	pendulum_dynamics_affine(o, cx, xI, f, g);
	Af DhI[MAXNP * NX];
	for (int e = 0; e < N * NX; e++) af_interval(cx, DhI[e], Dh[e], Dh[e]);
	double fl[MAXNP], fh[MAXNP], gl[MAXNP], gh[MAXNP];
	for (int s = 0; s < N; s++) { // Lfh, include/asif_utils.h:46-62 on AAF
		Af acc, t;
		af_const(acc, 0.0);
		for (int k = 0; k < NX; k++) {
			af_mul(cx, DhI[s + k * N], f[k], t);
			af_add(cx, acc, t, acc);
		}
		af_convert(acc, fl[s], fh[s]);
	}
	for (int s = 0; s < N; s++) { // Lgh (nu == 1), include/asif_utils.h:22-44 on AAF
		Af acc, t;
		af_const(acc, 0.0);
		for (int k = 0; k < NX; k++) {
			af_mul(cx, DhI[s + k * N], g[k], t);
			af_add(cx, acc, t, acc);
		}
		af_convert(acc, gl[s], gh[s]);
	}
	if (cx.overflow) { // capacity exceeded: poison the rows so that nothing downstream can look valid
		for (int s = 0; s < N; s++) fl[s] = fh[s] = gl[s] = gh[s] = __builtin_nan("");
	}
	if (reduced) {
		for (int r = 0; r < kRobustRedRows; r++) {
			const int s = r >> 1;
			const bool valid = s < N;
			const double lg = valid ? ((r & 1) ? gh[s] : gl[s]) : 0.0;
			a.A[(int64_t)(r + 0 * kRobustRedRows) * ld + i] = lg;
			a.A[(int64_t)(r + 1 * kRobustRedRows) * ld + i] = valid ? h[s] : 0.0;
			a.b[(int64_t)r * ld + i] = valid ? -fl[s] : -1e20;
		}
		return;
	}
	// full rows: fixed structure (src/asif_robust.cpp:103-133) + interval entries (:340-358)
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
	if (a.code) a.code[i] = cx.overflow ? -100 : 1;
}

struct RobustPolicy {
	int64_t B;
	DevOptions o;
	FilterArgs a; // a.A / a.b = staged reduced rows

	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
		static_assert(NV == 2 && NC == kRobustRedRows, "reduced robust QP");
		// src/asif_robust.cpp:89-101,140-142 restricted to (u, delta)
		qp.Hd[0] = 1.0;
		qp.Hd[1] = o.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * o.relaxCost * o.relaxLb;
		qp.lb[0] = o.lb[0];
		qp.lb[1] = o.relaxLb;
		qp.ub[0] = o.ub[0];
		qp.ub[1] = o.inf;
		load_rows<NV, NC, G>(a.A, a.b, a.ld, i, g, 0ull, qp);
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
	hipLaunchKernelGGL(robust_rows_kernel, dim3(grid_for(a.B, 1, 64)), dim3(64), 0, stream, o, a, !assemble_only);
	int e = (int)hipGetLastError();
	if (e || assemble_only) return e;
	const RobustPolicy p = {a.B, o, a};
	switch (S.lanes_per_qp) {
	case 0:
	case 2: return launch_policy<2, kRobustRedRows, 2>(S, p, stream);
	case 4: return launch_policy<2, kRobustRedRows, 4>(S, p, stream);
	default: return ASIF_HIP_EINVAL;
	}
}

} // namespace asif
