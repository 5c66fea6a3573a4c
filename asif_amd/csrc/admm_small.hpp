// admm_small.hpp -- in-register ADMM for the tiny dense QPs of the safety filter (nv <= 4).
//
// Replaces the reference's OSQP call chain (src/qpwrapper_osqp.cpp:55-261 -> osqp_setup/osqp_solve)
// for one QP per lane group.  Nothing is ported from OSQP: the splitting is the published ADMM of
// Stellato et al. (OSQP, Math. Prog. Comp. 2020) laid out for CDNA4:
//   * G lanes cooperate on one QP (G = 1, 2, 4, ... 64, compile time).  The NC general rows are dealt
//     round-robin to the G lanes (RPL rows per lane, in VGPRs); the nv variables, the nv bound rows and
//     the nv x nv reduced KKT factor are replicated in every lane of the group, so the only cross-lane
//     traffic per iteration is one nv-vector sum (A'(rho z - y)).
//   * The quasi-definite (nv+m) KKT system of the paper is eliminated to its nv x nv Schur complement
//     P + sigma I + A' diag(rho) A, factored LDL' in registers (nv <= 4: a handful of FMAs).  The
//     row-outer-product sum is formed once; a rho change refactors without any cross-lane traffic.
//   * Ruiz equilibration uses power-of-two factors (exponent arithmetic, exact to apply and undo).
//   * Every `check_interval` iterations: an active-set finish seeded by the iterates (exact solve on
//     the guessed working set, primal-dual corrections, accepted only when KKT-valid or when it
//     carries an exact Farkas certificate), then the unscaled residual test (OSQP's criteria),
//     OSQP's primal/dual infeasibility certificates, and rho adaptation.
// Form translation follows the wrapper the reference puts in front of OSQP (src/qpwrapper_osqp.cpp:263-376):
//   P = 2H, q = c, rows [A; I], l = [b; lb], u = [+inf | b where be; ub].
// General rows are one-sided (A x >= b) or equalities; `b` is data and always finite (the reference's
// own "infinity" is 1e20, include/asif.h:16, far below OSQP's 1e30), so they are never "loose" rows.
#pragma once
#include <hip/hip_runtime.h>
#include "asif_hip.h"
#include "qp_lane.hpp"
#include "gi_small.hpp"

namespace asif {

constexpr double kInfty = 1e30;
constexpr double kMinScaling = 1e-4;
constexpr double kMaxScaling = 1e4;
constexpr double kRhoMin = 1e-6;
constexpr double kRhoMax = 1e6;
constexpr double kRhoEqOverIneq = 1e3;
constexpr double kRhoTol = 1e-4;
constexpr double kPolishDelta = 1e-9;     // 1/penalty of the working-set solve (see finish())
constexpr double kPolishPrimalReg = 1e-6; // curvature lent to variables the cost does not touch
constexpr double kPolishKktTol = 1e-9;

constexpr int kStatusSolved = 1;
constexpr int kStatusSolvedInaccurate = 2;
constexpr int kStatusPrimalInfInaccurate = 3;
constexpr int kStatusDualInfInaccurate = 4;
constexpr int kStatusMaxIter = -2;
constexpr int kStatusPrimalInf = -3;
constexpr int kStatusDualInf = -4;

__device__ __forceinline__ double limit_scaling(double v)
{
	v = v < kMinScaling ? 1.0 : v;
	v = v > kMaxScaling ? kMaxScaling : v;
	return v;
}
// power-of-two stand-in for 1/sqrt(v): v = f*2^e, f in [0.5,1) -> 2^-(e>>1)
__device__ __forceinline__ double pow2_rsqrt(double v)
{
	int e;
	(void)frexp(v, &e);
	return ldexp(1.0, -(e >> 1));
}
__device__ __forceinline__ double pow2_floor(double v)
{
	int e;
	(void)frexp(v, &e);
	return ldexp(1.0, e - 1);
}
// largest power of two <= 1/v, without the division: v = f*2^e, f in [0.5,1) -> 2^-e, or 2^(1-e) when f == 0.5
__device__ __forceinline__ double pow2_floor_inv(double v)
{
	int e;
	const double f = frexp(v, &e);
	return ldexp(1.0, (f == 0.5 ? 1 : 0) - e);
}
// exact reciprocal of a power of two
__device__ __forceinline__ double pow2_inv(double p)
{
	int e;
	(void)frexp(p, &e);
	return ldexp(1.0, 1 - e);
}

template <int NV, int RPL, int G>
struct AdmmSmall {
	// scaled problem
	double P[NV], q[NV], D[NV], cs;
	double A[RPL][NV], l[RPL], E[RPL];
	bool eqr[RPL];
	double Ab[NV], lbs[NV], ubs[NV], Eb[NV]; // bound rows: Ab = scaled identity entry
	int clsb[NV];                            // bound-row class: -1 loose, 0 inequality, 1 equality
	double S[NV][NV];                        // sum_r w_r a_r a_r' over the general rows (w = 1 | 1e3)
	// iterates
	double x[NV], z[RPL], y[RPL], zb[NV], yb[NV];
	double dy[RPL], dyb[NV], dx[NV];
	// rho + factor
	double rho, rinv, rinv_eq, rb[NV], rbinv[NV];
	double L[NV][NV], Dinv[NV];
	// diagnostics: working-set solves and certificate iterations this lane went through
	int stat_rounds, stat_farkas;
	// working set of the last problem this object solved to optimality (see finish()); survives solve() calls
	int ws_act[RPL], ws_actb[NV];
	bool ws_valid, ws_stored;

	__device__ __forceinline__ void load_and_scale(const QpLaneData<NV, RPL> &in, int iters)
	{
#pragma unroll
		for (int j = 0; j < NV; j++) {
			P[j] = 2.0 * in.Hd[j];
			q[j] = in.c[j];
			D[j] = 1.0;
			Ab[j] = 1.0;
			Eb[j] = 1.0;
		}
#pragma unroll
		for (int r = 0; r < RPL; r++) {
			E[r] = 1.0;
#pragma unroll
			for (int j = 0; j < NV; j++) A[r][j] = in.A[r][j];
		}
		cs = 1.0;
#pragma unroll 1
		for (int it = 0; it < iters; it++) {
			double Dt[NV];
#pragma unroll
			for (int j = 0; j < NV; j++) {
				double v = 0.0;
#pragma unroll
				for (int r = 0; r < RPL; r++) v = fmax(v, fabs(A[r][j]));
				v = gmax<G>(v);
				v = fmax(v, fmax(fabs(P[j]), fabs(Ab[j])));
				Dt[j] = pow2_rsqrt(limit_scaling(v));
			}
#pragma unroll
			for (int r = 0; r < RPL; r++) {
				double v = 0.0;
#pragma unroll
				for (int j = 0; j < NV; j++) v = fmax(v, fabs(A[r][j]));
				const double Et = pow2_rsqrt(limit_scaling(v));
				E[r] *= Et;
#pragma unroll
				for (int j = 0; j < NV; j++) A[r][j] *= Et * Dt[j];
			}
			double cm = 0.0, qn = 0.0;
#pragma unroll
			for (int j = 0; j < NV; j++) {
				const double Etb = pow2_rsqrt(limit_scaling(fabs(Ab[j])));
				Eb[j] *= Etb;
				Ab[j] *= Etb * Dt[j];
				P[j] *= Dt[j] * Dt[j];
				q[j] *= Dt[j];
				D[j] *= Dt[j];
				cm += fabs(P[j]);
				qn = fmax(qn, fabs(q[j]));
			}
			cm *= (1.0 / NV);
			qn = limit_scaling(qn);
			const double ct = pow2_floor_inv(limit_scaling(fmax(cm, qn)));
#pragma unroll
			for (int j = 0; j < NV; j++) {
				P[j] *= ct;
				q[j] *= ct;
			}
			cs *= ct;
		}
#pragma unroll
		for (int r = 0; r < RPL; r++) {
			l[r] = in.b[r] * E[r];
			eqr[r] = in.eq[r];
		}
#pragma unroll
		for (int j = 0; j < NV; j++) {
			lbs[j] = in.lb[j] * Eb[j];
			ubs[j] = in.ub[j] * Eb[j];
			if (lbs[j] < -kInfty * kMinScaling && ubs[j] > kInfty * kMinScaling) clsb[j] = -1;
			else if (ubs[j] - lbs[j] < kRhoTol) clsb[j] = 1;
			else clsb[j] = 0;
		}
		// the rho-independent part of the reduced KKT matrix
#pragma unroll
		for (int a = 0; a < NV; a++)
#pragma unroll
			for (int b = 0; b <= a; b++) {
				double s = 0.0;
#pragma unroll
				for (int r = 0; r < RPL; r++) s += (eqr[r] ? kRhoEqOverIneq : 1.0) * A[r][a] * A[r][b];
				S[a][b] = gsum<G>(s);
			}
	}

	// per-row rho (OSQP's rho_vec) and the LDL' of  P + sigma I + A' diag(rho) A ; lane-local.
	__device__ __forceinline__ bool set_rho_and_factor(double rho0, double sigma)
	{
		rho = rho0;
		rinv = 1.0 / rho0;
		rinv_eq = rinv * (1.0 / kRhoEqOverIneq);
#pragma unroll
		for (int j = 0; j < NV; j++) {
			rb[j] = clsb[j] < 0 ? kRhoMin : (clsb[j] > 0 ? kRhoEqOverIneq * rho0 : rho0);
			rbinv[j] = clsb[j] < 0 ? 1.0 / kRhoMin : (clsb[j] > 0 ? rinv_eq : rinv);
		}
#pragma unroll
		for (int a = 0; a < NV; a++)
#pragma unroll
			for (int b = 0; b <= a; b++) {
				double s = rho0 * S[a][b];
				if (a == b) s += P[a] + sigma + rb[a] * Ab[a] * Ab[a];
				L[a][b] = s;
			}
		return ldl_factor<NV>(L, Dinv);
	}

	// one ADMM iteration (x~/z~ solve, relaxation, projection, dual update)
	__device__ __forceinline__ void iterate(double sigma, double alpha)
	{
		double rhs[NV];
		const double rho_eq = kRhoEqOverIneq * rho;
#pragma unroll
		for (int j = 0; j < NV; j++) rhs[j] = 0.0;
#pragma unroll
		for (int r = 0; r < RPL; r++) {
			const double t = (eqr[r] ? rho_eq : rho) * z[r] - y[r];
#pragma unroll
			for (int j = 0; j < NV; j++) rhs[j] += A[r][j] * t;
		}
#pragma unroll
		for (int j = 0; j < NV; j++) {
			rhs[j] = gsum<G>(rhs[j]);
			rhs[j] += sigma * x[j] - q[j] + Ab[j] * (rb[j] * zb[j] - yb[j]);
		}
		ldl_solve<NV>(L, Dinv, rhs); // rhs = x~
		const double oma = 1.0 - alpha;
#pragma unroll
		for (int r = 0; r < RPL; r++) {
			double zt = 0.0;
#pragma unroll
			for (int j = 0; j < NV; j++) zt += A[r][j] * rhs[j];
			const double zr = alpha * zt + oma * z[r];
			const double zn = eqr[r] ? l[r] : fmax(zr + y[r] * rinv, l[r]);
			dy[r] = (eqr[r] ? rho_eq : rho) * (zr - zn);
			y[r] += dy[r];
			z[r] = zn;
		}
#pragma unroll
		for (int j = 0; j < NV; j++) {
			const double zt = Ab[j] * rhs[j];
			const double zr = alpha * zt + oma * zb[j];
			const double zn = fmin(fmax(zr + yb[j] * rbinv[j], lbs[j]), ubs[j]);
			dyb[j] = rb[j] * (zr - zn);
			yb[j] += dyb[j];
			zb[j] = zn;
			const double xn = alpha * rhs[j] + oma * x[j];
			dx[j] = xn - x[j];
			x[j] = xn;
		}
	}

	// ---- Active-set finish.  The ADMM iterates only have to point at the right working set; this
	// routine then solves the equality-constrained QP on that set exactly (regularised primal Schur
	// system + refinement, the idea of OSQP's polish step), classifies every row at the result, and
	// corrects the set primal-dual-active-set style: rows found violated join, rows with a wrongly
	// signed multiplier leave.  Nothing is accepted on trust:
	//   return 1  the point satisfies the KKT conditions of the FULL problem (optimal; xp filled),
	//   return 2  the least-squares residual of the working rows is a Farkas certificate
	//             (all residuals have the sign of a violation, A_S' v = 0 by construction, and then
	//             sum r_i v_i = |v|^2 > 0): the QP is infeasible,
	//   return 0  undecided -> ADMM keeps iterating and the next check tries again.
	// act: 0 inactive, -1 at lower bound, +1 at upper bound, 2 equality (always in, multiplier free).
	// idl: penalty (1/delta) of the working-set solve; the caller raises it after an undecided attempt.
	__device__ __forceinline__ int finish(double (&xp)[NV], int rounds, int refine, double idl)
	{
		int act[RPL], actb[NV];
		double nu[RPL], nub[NV], rtb[NV];
#pragma unroll
		for (int r = 0; r < RPL; r++) act[r] = eqr[r] ? 2 : ((z[r] - l[r] < -y[r]) ? -1 : 0);
#pragma unroll
		for (int j = 0; j < NV; j++) {
			actb[j] = 0;
			if (clsb[j] > 0) actb[j] = 2;
			else if (zb[j] - lbs[j] < -yb[j]) actb[j] = -1;
			else if (ubs[j] - zb[j] < yb[j]) actb[j] = 1;
		}
		// warm start (closed-loop rollouts): the first attempt starts from the working set the previous control
		// step ended with -- between consecutive steps it rarely changes
		if (ws_valid) {
#pragma unroll
			for (int r = 0; r < RPL; r++) act[r] = eqr[r] ? 2 : ws_act[r];
#pragma unroll
			for (int j = 0; j < NV; j++) actb[j] = clsb[j] > 0 ? 2 : ws_actb[j];
		}
		ws_valid = false;
#pragma unroll
		for (int j = 0; j < NV; j++) rtb[j] = actb[j] == 1 ? ubs[j] : lbs[j];
		int verdict = 0;
#pragma unroll 1
		for (int round = 0; round <= rounds; round++) {
			stat_rounds++;
			// -- equality-constrained solve on the working set
			// Method of multipliers on the working rows with penalty idl = 1/delta, `refine` steps from zero.  Its
			// contraction factor is about delta / (delta + s^2), s the smallest singular value of the (scaled)
			// working rows seen through P^-1/2.  With 1/delta = 1e9 two steps meet the rows unless they are parallel
			// to ~1e-4; the corrections below usually cope with an unconverged solve (one of two parallel rows
			// leaves); where they do not, the attempt ends undecided and solve() repeats it with 1000x the penalty.
			double Mp[NV][NV], Mi[NV];
#pragma unroll
			for (int a = 0; a < NV; a++)
#pragma unroll
				for (int b = 0; b <= a; b++) {
					double s = 0.0;
#pragma unroll
					for (int r = 0; r < RPL; r++) s += act[r] ? A[r][a] * A[r][b] : 0.0;
					s = gsum<G>(s) * idl;
					// primal regularisation only where the cost has no curvature of its own
					if (a == b) s += (P[a] > 0.0 ? P[a] : kPolishPrimalReg) + (actb[a] ? Ab[a] * Ab[a] * idl : 0.0);
					Mp[a][b] = s;
				}
			bool ok = ldl_factor<NV>(Mp, Mi);
#pragma unroll
			for (int j = 0; j < NV; j++) { xp[j] = 0.0; nub[j] = 0.0; }
#pragma unroll
			for (int r = 0; r < RPL; r++) nu[r] = 0.0;
#pragma unroll 1
			for (int it = 0; it < refine; it++) {
				double rhs[NV], e2[RPL], e2b[NV];
#pragma unroll
				for (int j = 0; j < NV; j++) rhs[j] = 0.0;
#pragma unroll
				for (int r = 0; r < RPL; r++) {
					double ax = 0.0;
#pragma unroll
					for (int j = 0; j < NV; j++) ax += A[r][j] * xp[j];
					e2[r] = l[r] - ax;
					const double w = act[r] ? (e2[r] * idl - nu[r]) : 0.0;
#pragma unroll
					for (int j = 0; j < NV; j++) rhs[j] += A[r][j] * w;
				}
#pragma unroll
				for (int j = 0; j < NV; j++) {
					rhs[j] = gsum<G>(rhs[j]);
					e2b[j] = rtb[j] - Ab[j] * xp[j];
					rhs[j] += -q[j] - P[j] * xp[j] + (actb[j] ? Ab[j] * (e2b[j] * idl - nub[j]) : 0.0);
				}
				ldl_solve<NV>(Mp, Mi, rhs); // rhs = dx
#pragma unroll
				for (int r = 0; r < RPL; r++) {
					double adx = 0.0;
#pragma unroll
					for (int j = 0; j < NV; j++) adx += A[r][j] * rhs[j];
					nu[r] += act[r] ? (adx - e2[r]) * idl : 0.0;
				}
#pragma unroll
				for (int j = 0; j < NV; j++) {
					nub[j] += actb[j] ? (Ab[j] * rhs[j] - e2b[j]) * idl : 0.0;
					xp[j] += rhs[j];
				}
			}
			// -- classify every row at xp; build the corrected working set
			int nact[RPL], nactb[NV], sideb[NV];
			bool viol_active = false, changed = false;
			double g[NV], gm[NV], vm[RPL], vmb[NV], worst = 0.0;
#pragma unroll
			for (int j = 0; j < NV; j++) { g[j] = 0.0; gm[j] = 0.0; }
#pragma unroll
			for (int r = 0; r < RPL; r++) {
				double ax = 0.0;
#pragma unroll
				for (int j = 0; j < NV; j++) ax += A[r][j] * xp[j];
				const double tol = kPolishKktTol * (1.0 + fabs(ax));
				const bool below = ax < l[r] - tol, above = eqr[r] && ax > l[r] + tol;
				const bool wrong = act[r] == -1 && nu[r] > kPolishKktTol * (1.0 + fabs(nu[r]));
				ok = ok && !below && !above && !wrong;
				viol_active = viol_active || ((below || above) && act[r] != 0);
				nact[r] = eqr[r] ? 2 : (wrong ? 0 : act[r]);
				changed = changed || (nact[r] != act[r]);
				vm[r] = (below && act[r] == 0) ? l[r] - ax : 0.0; // candidate to join, by violation
				worst = fmax(worst, vm[r]);
#pragma unroll
				for (int j = 0; j < NV; j++) {
					const double t = act[r] ? A[r][j] * nu[r] : 0.0;
					g[j] += t;
					gm[j] += fabs(t);
				}
			}
#pragma unroll
			for (int j = 0; j < NV; j++) {
				const double ax = Ab[j] * xp[j];
				const double tol = kPolishKktTol * (1.0 + fabs(ax));
				const bool below = ax < lbs[j] - tol, above = ax > ubs[j] + tol;
				const double nt = kPolishKktTol * (1.0 + fabs(nub[j]));
				const bool wrong = (actb[j] == -1 && nub[j] > nt) || (actb[j] == 1 && nub[j] < -nt);
				ok = ok && !below && !above && !wrong;
				viol_active = viol_active || ((below || above) && actb[j] != 0);
				nactb[j] = actb[j] == 2 ? 2 : (wrong ? 0 : actb[j]);
				changed = changed || (nactb[j] != actb[j]);
				vmb[j] = (actb[j] == 0 && below) ? lbs[j] - ax : ((actb[j] == 0 && above) ? ax - ubs[j] : 0.0);
				sideb[j] = above ? 1 : -1;
				worst = fmax(worst, vmb[j]);
				// stationarity, relative to the size of its own terms (a column can be scaled very small)
				const double tb = actb[j] ? Ab[j] * nub[j] : 0.0;
				const double gj = gsum<G>(g[j]) + P[j] * xp[j] + q[j] + tb;
				const double mag = gsum<G>(gm[j]) + fabs(P[j] * xp[j]) + fabs(q[j]) + fabs(tb);
				ok = ok && !(fabs(gj) > 1e-10 * mag + 1e-300);
			}
			if (gand<G>(ok ? 1 : 0)) {
				verdict = 1;
#pragma unroll
				for (int r = 0; r < RPL; r++) ws_act[r] = act[r];
#pragma unroll
				for (int j = 0; j < NV; j++) ws_actb[j] = actb[j];
				ws_stored = true;
				break;
			}
			// -- infeasibility: least-squares point of the working rows.  Rows met with slack there
			//    leave (removal only -> monotone), the rest must all show the sign of a violation.
			if (gand<G>(viol_active ? 0 : 1) == 0) {
				int fa[RPL], fab[NV];
#pragma unroll
				for (int r = 0; r < RPL; r++) fa[r] = act[r];
#pragma unroll
				for (int j = 0; j < NV; j++) fab[j] = actb[j];
				bool cert = false;
#pragma unroll 1
				for (int fi = 0; fi <= 2; fi++) {
					stat_farkas++;
					double Ms[NV][NV], Msi[NV], wv[NV], dmax = 0.0;
#pragma unroll
					for (int a = 0; a < NV; a++)
#pragma unroll
						for (int b = 0; b <= a; b++) {
							double s = 0.0;
#pragma unroll
							for (int r = 0; r < RPL; r++) s += fa[r] ? A[r][a] * A[r][b] : 0.0;
							s = gsum<G>(s);
							if (a == b) {
								s += fab[a] ? Ab[a] * Ab[a] : 0.0;
								dmax = fmax(dmax, s);
							}
							Ms[a][b] = s;
						}
#pragma unroll
					for (int a = 0; a < NV; a++) Ms[a][a] += 1e-14 * dmax + 1e-300;
					bool fk = ldl_factor<NV>(Ms, Msi);
#pragma unroll
					for (int j = 0; j < NV; j++) {
						double s = 0.0;
#pragma unroll
						for (int r = 0; r < RPL; r++) s += fa[r] ? A[r][j] * l[r] : 0.0;
						wv[j] = gsum<G>(s) + (fab[j] ? Ab[j] * rtb[j] : 0.0);
					}
					ldl_solve<NV>(Ms, Msi, wv);
					{ // one refinement step removes the footprint of the regularisation from A_S' v
						double g2[NV];
#pragma unroll
						for (int j = 0; j < NV; j++) g2[j] = 0.0;
#pragma unroll
						for (int r = 0; r < RPL; r++) {
							double ax = 0.0;
#pragma unroll
							for (int j = 0; j < NV; j++) ax += A[r][j] * wv[j];
							const double v = fa[r] ? l[r] - ax : 0.0;
#pragma unroll
							for (int j = 0; j < NV; j++) g2[j] += A[r][j] * v;
						}
#pragma unroll
						for (int j = 0; j < NV; j++)
							g2[j] = gsum<G>(g2[j]) + (fab[j] ? Ab[j] * (rtb[j] - Ab[j] * wv[j]) : 0.0);
						ldl_solve<NV>(Ms, Msi, g2);
#pragma unroll
						for (int j = 0; j < NV; j++) wv[j] += g2[j];
					}
					bool signs_ok = true;
					double vmax = 0.0, atv[NV];
#pragma unroll
					for (int j = 0; j < NV; j++) atv[j] = 0.0;
#pragma unroll
					for (int r = 0; r < RPL; r++) {
						double ax = 0.0;
#pragma unroll
						for (int j = 0; j < NV; j++) ax += A[r][j] * wv[j];
						double v = fa[r] ? l[r] - ax : 0.0;
						if (fa[r] == -1 && v < -1e-12) { // met with slack: leaves the set
							signs_ok = false;
							fa[r] = 0;
							v = 0.0;
						}
						vmax = fmax(vmax, fabs(v));
#pragma unroll
						for (int j = 0; j < NV; j++) atv[j] += A[r][j] * v;
					}
					double vb[NV];
#pragma unroll
					for (int j = 0; j < NV; j++) {
						vb[j] = fab[j] ? rtb[j] - Ab[j] * wv[j] : 0.0;
						if ((fab[j] == -1 && vb[j] < -1e-12) || (fab[j] == 1 && vb[j] > 1e-12)) {
							signs_ok = false;
							fab[j] = 0;
							vb[j] = 0.0;
						}
						vmax = fmax(vmax, fabs(vb[j]));
					}
					vmax = gmax<G>(vmax);
					if (gand<G>(signs_ok ? 1 : 0)) {
						fk = fk && (vmax > 1e-7);
#pragma unroll
						for (int j = 0; j < NV; j++) {
							const double gj = gsum<G>(atv[j]) + Ab[j] * vb[j];
							fk = fk && !(fabs(gj) > 1e-9 * vmax);
						}
						cert = gand<G>(fk ? 1 : 0) != 0;
						break;
					}
				}
				if (cert) { verdict = 2; break; }
			}
			// of the rows found violated only the most violated one(s) join: the working set then stays
			// consistent on feasible problems and an inconsistency points at a genuine conflict
			worst = gmax<G>(worst);
			if (gand<G>((changed || worst > 0.0) ? 0 : 1)) break; // nothing to correct: leave it to ADMM
#pragma unroll
			for (int r = 0; r < RPL; r++) act[r] = (worst > 0.0 && vm[r] == worst) ? -1 : nact[r];
#pragma unroll
			for (int j = 0; j < NV; j++) {
				actb[j] = (worst > 0.0 && vmb[j] == worst) ? sideb[j] : nactb[j];
				rtb[j] = actb[j] == 1 ? ubs[j] : lbs[j];
			}
		}
		return verdict;
	}

	// Cold-start solve.  `status` follows QPWrapperOsqp::solve() (src/qpwrapper_osqp.cpp:225-238):
	// 1 when solved, the raw OSQP-style code otherwise.  xout is unscaled.  Must be called by every
	// lane of the wave (group reductions and the wave-uniform exit test).
	// finish_first (asif_hip_solver::polish == 2): before any iteration the lane group runs the dual active-set
	// method of gi_small.hpp on the unscaled problem; it decides strictly convex problems of this size outright
	// (optimal point or proof of infeasibility) and the iterations below only run for what it leaves undecided.
	// warm = true: the first finish attempt of the ITERATIONS starts from the working set the previous solve() of
	// this object ended with (if the iterations ended it at an optimum); iterates still start from zero.
	__device__ __forceinline__ void solve(const QpLaneData<NV, RPL> &in, const asif_hip_solver &S_, double (&xout)[NV],
	                                      int &status, int &iters, bool warm = false, bool finish_first = false)
	{
		ws_valid = warm && ws_stored;
		ws_stored = false;
		status = 0;
		iters = 0;
		stat_rounds = 0;
		stat_farkas = 0;
#pragma unroll
		for (int j = 0; j < NV; j++) xout[j] = 0.0;
		// non-finite data (a NaN / inf state): the reference's solver never converges on it and returns max_iter
		// (qp_lane.hpp: qp_data_nonfinite); latched here for every solver mode, the iterations skip a latched lane
		const bool nonfinite = qp_data_nonfinite<NV, RPL, G>(in.Hd, in.c, in.lb, in.ub, in.A, in.b);
		if (nonfinite) status = kStatusMaxIter;
		// polish == 0 (the iterations alone): the same stage runs first as well, but its verdict is only kept for the lanes
		// the iterations leave at max_iter -- the policy of the wave kernels (k_qp.hip: launch_qp_wave): a problem the
		// iterations do not finish is answered by the exact method instead of a MAX_ITER status (profiles/r03/
		// soak_parity.txt: 4e-5 of C3's instances ran to max_iter although an optimum exists)
		int gi_kept = kGiUndecided;
		double gi_x[NV];
#pragma unroll
		for (int j = 0; j < NV; j++) gi_x[j] = 0.0;
		if constexpr (NV <= 3) {
			if (finish_first) {
				double xg[NV];
				int gsteps;
				const int v = GiSmall<NV, RPL, G>::solve_unchecked(in, (int)(threadIdx.x % G), 8 * NV + 4, xg, gsteps);
				if (S_.polish == 0) {
					gi_kept = v;
#pragma unroll
					for (int j = 0; j < NV; j++) gi_x[j] = xg[j];
				} else {
					stat_rounds = gsteps;
					if (nonfinite) {
					} else if (v == kGiFailed) { // overflow inside the stage: nothing the iterations could do better
						status = kStatusMaxIter;
					} else if (v == kGiOptimal) {
						status = kStatusSolved;
#pragma unroll
						for (int j = 0; j < NV; j++) xout[j] = xg[j];
					} else if (v == kGiInfeasible) {
						status = kStatusPrimalInf;
					}
				}
			}
		}
		if (__all(status != 0)) return; // the usual case: nothing left for the iterations
		load_and_scale(in, S_.scaling_iters);
#pragma unroll
		for (int j = 0; j < NV; j++) { x[j] = 0.0; zb[j] = 0.0; yb[j] = 0.0; dyb[j] = 0.0; dx[j] = 0.0; }
#pragma unroll
		for (int r = 0; r < RPL; r++) { z[r] = 0.0; y[r] = 0.0; dy[r] = 0.0; }
		bool fact_ok = set_rho_and_factor(S_.rho, S_.sigma);
		const double cinv = pow2_inv(cs);
		int it = 0;
		double penalty = 1.0 / kPolishDelta; // of the finish's working-set solves; per problem, see finish()
		const int K = S_.check_interval > 0 ? S_.check_interval : 10;
		// rho is re-estimated every R iterations, NOT at every check: with checks every 1-2 iterations the estimate
		// chases its own transient and a per-check update never settles (1.3 % of the feasible C2 problems ran into
		// max_iter that way); OSQP ties it to a multiple of its 25-iteration check period
		const int R = S_.adaptive_rho_interval > 0 ? S_.adaptive_rho_interval : 25;
		int next_rho = R;
		while (it < S_.max_iter) {
			if (__all(status != 0)) break; // wave-uniform: every lane has latched its result
#pragma unroll 1
			for (int k = 0; k < K; k++) iterate(S_.sigma, S_.alpha);
			it += K;
			const bool last = it >= S_.max_iter;

			int st = 0;
			double xs[NV]; // candidate solution, scaled
#pragma unroll
			for (int j = 0; j < NV; j++) xs[j] = x[j];
			if (S_.polish) {
				double xp[NV];
				const int v = finish(xp, S_.active_set_rounds, S_.refine_steps, penalty);
				if (v == 1) {
					st = kStatusSolved;
#pragma unroll
					for (int j = 0; j < NV; j++) xs[j] = xp[j];
				} else if (v == 2) {
					st = kStatusPrimalInf;
				} else if (status == 0 && penalty < 1e14) {
					penalty *= 1e3; // undecided: the next attempt, K iterations on, solves its working sets more stiffly
				}
			}
			// The residual tests below are only needed by lanes the finish left undecided
			// (wave-uniform branch: the block contains group reductions).
			if (__any(st == 0 && status == 0)) {
				// ---- residual norms, unscaled (termination) and scaled (rho estimate)
				double aty[NV], pri = 0.0, nz = 0.0, nax = 0.0, pri_s = 0.0, nz_s = 0.0, nax_s = 0.0;
#pragma unroll
				for (int j = 0; j < NV; j++) aty[j] = 0.0;
#pragma unroll
				for (int r = 0; r < RPL; r++) {
					double ax = 0.0;
#pragma unroll
					for (int j = 0; j < NV; j++) {
						ax += A[r][j] * x[j];
						aty[j] += A[r][j] * y[r];
					}
					const double ei = pow2_inv(E[r]);
					pri = fmax(pri, fabs(ei * (ax - z[r])));
					nz = fmax(nz, fabs(ei * z[r]));
					nax = fmax(nax, fabs(ei * ax));
					pri_s = fmax(pri_s, fabs(ax - z[r]));
					nz_s = fmax(nz_s, fabs(z[r]));
					nax_s = fmax(nax_s, fabs(ax));
				}
				double dua = 0.0, nq = 0.0, naty = 0.0, npx = 0.0, dua_s = 0.0, nq_s = 0.0, naty_s = 0.0, npx_s = 0.0;
#pragma unroll
				for (int j = 0; j < NV; j++) {
					const double ax = Ab[j] * x[j];
					const double ei = pow2_inv(Eb[j]);
					pri = fmax(pri, fabs(ei * (ax - zb[j])));
					nz = fmax(nz, fabs(ei * zb[j]));
					nax = fmax(nax, fabs(ei * ax));
					pri_s = fmax(pri_s, fabs(ax - zb[j]));
					nz_s = fmax(nz_s, fabs(zb[j]));
					nax_s = fmax(nax_s, fabs(ax));
					const double at = gsum<G>(aty[j]) + Ab[j] * yb[j];
					const double px = P[j] * x[j];
					const double di = pow2_inv(D[j]);
					const double rd = px + q[j] + at;
					dua = fmax(dua, fabs(di * rd));
					nq = fmax(nq, fabs(di * q[j]));
					naty = fmax(naty, fabs(di * at));
					npx = fmax(npx, fabs(di * px));
					dua_s = fmax(dua_s, fabs(rd));
					nq_s = fmax(nq_s, fabs(q[j]));
					naty_s = fmax(naty_s, fabs(at));
					npx_s = fmax(npx_s, fabs(px));
				}
				pri = gmax<G>(pri); nz = gmax<G>(nz); nax = gmax<G>(nax);
				pri_s = gmax<G>(pri_s); nz_s = gmax<G>(nz_s); nax_s = gmax<G>(nax_s);
				dua *= cinv;
				// ---- primal-infeasibility certificate from the last delta_y (projected on the polar cone)
				double ndy = 0.0, lhs = 0.0, atdy[NV];
#pragma unroll
				for (int j = 0; j < NV; j++) atdy[j] = 0.0;
#pragma unroll
				for (int r = 0; r < RPL; r++) {
					const double v = eqr[r] ? dy[r] : fmin(dy[r], 0.0);
					ndy = fmax(ndy, fabs(E[r] * v));
					lhs += l[r] * v;
#pragma unroll
					for (int j = 0; j < NV; j++) atdy[j] += A[r][j] * v;
				}
				ndy = gmax<G>(ndy);
				lhs = gsum<G>(lhs);
				double natdy = 0.0;
				{
					double vb[NV];
#pragma unroll
					for (int j = 0; j < NV; j++) {
						double v = dyb[j];
						if (ubs[j] > kInfty * kMinScaling) v = (lbs[j] < -kInfty * kMinScaling) ? 0.0 : fmin(v, 0.0);
						else if (lbs[j] < -kInfty * kMinScaling) v = fmax(v, 0.0);
						vb[j] = v;
						ndy = fmax(ndy, fabs(Eb[j] * v));
						lhs += ubs[j] * fmax(v, 0.0) + lbs[j] * fmin(v, 0.0);
					}
#pragma unroll
					for (int j = 0; j < NV; j++) {
						const double t = (gsum<G>(atdy[j]) + Ab[j] * vb[j]) * pow2_inv(D[j]);
						natdy = fmax(natdy, fabs(t));
					}
				}
				// ---- dual-infeasibility certificate pieces (delta_x)
				double ndx = 0.0, qdx = 0.0, npdx = 0.0;
#pragma unroll
				for (int j = 0; j < NV; j++) {
					ndx = fmax(ndx, fabs(D[j] * dx[j]));
					qdx += q[j] * dx[j];
					npdx = fmax(npdx, fabs(P[j] * dx[j] * pow2_inv(D[j])));
				}
#pragma unroll 1
				for (int approx = 0; approx <= (last ? 1 : 0); approx++) {
					if (st) break;
					const double k = approx ? 10.0 : 1.0;
					const double ea = k * S_.eps_abs, er = k * S_.eps_rel;
					const double epi = k * S_.eps_prim_inf, edi = k * S_.eps_dual_inf;
					const double eps_pri = ea + er * fmax(nz, nax);
					const double eps_dua = ea + er * cinv * fmax(nq, fmax(naty, npx));
					const bool prim_ok = pri < eps_pri, dual_ok = dua < eps_dua;
					if (prim_ok && dual_ok) st = approx ? kStatusSolvedInaccurate : kStatusSolved;
					else if (!prim_ok && ndy > epi && lhs < -epi * ndy && natdy < epi * ndy)
						st = approx ? kStatusPrimalInfInaccurate : kStatusPrimalInf;
					else if (!dual_ok && ndx > edi && qdx < -cs * edi * ndx && npdx < cs * edi * ndx) {
						bool cone = true;
#pragma unroll
						for (int r = 0; r < RPL; r++) {
							double a = 0.0;
#pragma unroll
							for (int j = 0; j < NV; j++) a += A[r][j] * dx[j];
							a *= pow2_inv(E[r]);
							cone = cone && !((eqr[r] && a > edi * ndx) || a < -edi * ndx);
						}
#pragma unroll
						for (int j = 0; j < NV; j++) {
							const double a = Ab[j] * dx[j] * pow2_inv(Eb[j]);
							cone = cone && !((ubs[j] < kInfty * kMinScaling && a > edi * ndx) ||
							                 (lbs[j] > -kInfty * kMinScaling && a < -edi * ndx));
						}
						if (gand<G>(cone ? 1 : 0)) st = approx ? kStatusDualInfInaccurate : kStatusDualInf;
					}
				}
				// ---- rho adaptation (OSQP's estimate on the scaled residuals); lane-local refactor
				if (S_.adaptive_rho && !last && it >= next_rho) {
					next_rho = it + R;
					const double pr = pri_s / (fmax(nz_s, nax_s) + 1e-10);
					const double dr = dua_s / (fmax(nq_s, fmax(naty_s, npx_s)) + 1e-10);
					double rn = rho * sqrt(pr / (dr + 1e-10));
					rn = fmin(fmax(rn, kRhoMin), kRhoMax);
					if (rn > rho * S_.adaptive_rho_tolerance || rn < rho / S_.adaptive_rho_tolerance)
						fact_ok = set_rho_and_factor(rn, S_.sigma) && fact_ok;
				}
			}
			if (!st && (last || !fact_ok)) st = kStatusMaxIter;
			if (status == 0 && st != 0) { // latch the first verdict
				status = st;
				iters = it;
#pragma unroll
				for (int j = 0; j < NV; j++) xout[j] = D[j] * xs[j];
			}
		}
		if (status == 0) { // max_iter == 0
			status = kStatusMaxIter;
			iters = it;
#pragma unroll
			for (int j = 0; j < NV; j++) xout[j] = D[j] * x[j];
		}
		if (S_.polish == 0 && status == kStatusMaxIter && !nonfinite) { // see gi_kept above
			if (gi_kept == kGiOptimal) {
				status = kStatusSolved;
#pragma unroll
				for (int j = 0; j < NV; j++) xout[j] = gi_x[j];
			} else if (gi_kept == kGiInfeasible) {
				status = kStatusPrimalInf;
			}
		}
		if (status == kStatusSolvedInaccurate) status = kStatusSolved; // src/qpwrapper_osqp.cpp:225
	}
};

} // namespace asif
