// qp_alm_host.cpp -- the wave-level kernels' method (asif_amd/csrc/qp_lds.hpp / qp_inv.hpp) on the calling thread, for
// ASIF::QPWrapperHost on shapes beyond the in-register stage (nv > 3, or a full cost matrix): the lifted problems
// ASIFrobust (18 x 12, 22 x 15) and ASIFrealizable (38 x 29 ... 86 x 65) hand their solver for ONE agent.
//
// Same method, same constants, same schedule as the kernels -- proximal method of multipliers whose inner problems
//   min_x  1/2 x'Px + q'x + |x - xhat|^2 / (2 gamma) + sum_i mu_i / 2 dist^2(a_i x + y_i / mu_i, [l_i, u_i])
// are solved exactly by a semismooth Newton iteration (generalised Hessian K_J = P + I/gamma + sum_{i in J} mu_i a_i a_i',
// J = rows outside their interval) with an exact line search on the piecewise-linear derivative; outer update
// y <- mu (s - proj s), xhat <- x; termination on scaled residuals (1e-10), OSQP's primal-infeasibility certificate on
// the dual increment; penalties raised by the factor the observed contraction asks for (penalty_jump).  What differs is
// the linear algebra -- a dense Cholesky factor of K_J here, rebuilt when J changes, where the kernels keep an inverse
// by rank-one steps or a factor in LDS -- and the bookkeeping around it (no polish stage, a scalar line search over
// sorted breakpoints); iterates are therefore not bit-identical to the kernels', the bar is the same (status on every
// instance, 1e-6 on u: tests/test_alm_host.py).  Selected BY NAME (QPSOLVER::HOST), never a fallback, never the oracle.
// Form translation as QPWrapperOsqp (src/qpwrapper_osqp.cpp:263-376): P = 2H, rows [A; I], l = [b; lb], u = [inf | b; ub].
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ASIF {
namespace hostqp {

namespace {

constexpr double kGamma = 1e7, kMu0 = 10.0, kMuMax = 1e4, kRhoTarget = 1e-3, kInfty = 1e30, kMinScaling = 1e-4,
                 kMaxScaling = 1e4, kRhoTol = 1e-4;
constexpr int kMaxOuter = 80, kMaxInner = 60;

double limit_scaling(double v) { return v < kMinScaling ? 1.0 : (v > kMaxScaling ? kMaxScaling : v); }
// 2^-ceil(log2(sqrt v)) style power-of-two scalings, as the kernels take them (exact products, no rounding)
double pow2_rsqrt(double v)
{
	int e;
	(void)std::frexp(v, &e);
	return std::ldexp(1.0, -(e >> 1));
}
double pow2_floor_inv(double v)
{
	int e;
	(void)std::frexp(1.0 / v, &e);
	return std::ldexp(1.0, e - 1);
}
double penalty_jump(double pri, double pri_prev)
{
	const double rho = pri < 0.999 * pri_prev ? pri / pri_prev : 0.999;
	const double f = (1.0 / kRhoTarget - 1.0) / (1.0 / rho - 1.0);
	return f < 10.0 ? 10.0 : (f > 1e6 ? 1e6 : f);
}
double clip(double v, double lo, double hi) { return std::min(std::max(v, lo), hi); }

// in-place Cholesky of the n x n symmetric positive definite K (row-major, lower triangle); false if a pivot fails
bool cholesky(std::vector<double> &K, int n)
{
	for (int j = 0; j < n; j++) {
		double d = K[j * n + j];
		for (int k = 0; k < j; k++) d -= K[j * n + k] * K[j * n + k];
		if (!(d > 0.0)) return false;
		const double l = std::sqrt(d);
		K[j * n + j] = l;
		for (int i = j + 1; i < n; i++) {
			double s = K[i * n + j];
			for (int k = 0; k < j; k++) s -= K[i * n + k] * K[j * n + k];
			K[i * n + j] = s / l;
		}
	}
	return true;
}
void chol_solve(const std::vector<double> &L, int n, std::vector<double> &v)
{
	for (int i = 0; i < n; i++) {
		double s = v[i];
		for (int k = 0; k < i; k++) s -= L[i * n + k] * v[k];
		v[i] = s / L[i * n + i];
	}
	for (int i = n - 1; i >= 0; i--) {
		double s = v[i];
		for (int k = i + 1; k < n; k++) s -= L[k * n + i] * v[k];
		v[i] = s / L[i * n + i];
	}
}

} // namespace

// Returns QPWrapperOsqp::solve's value: 1, -3 (primal infeasible), -2 (iteration budget / data outside the domain).
// H: nv x nv column-major (only the diagonal is read when diag), A: nc x nv column-major.
// warm_x[nv], warm_y[nc + nv] (may be null): the iterate and the multipliers of [A; I] in the caller's units, written
// by every call (zeros unless the verdict is 1) and, with warm_in, read as the start -- asif_hip_qp_solve_batch_warm's
// contract (include/asif_hip.h), OSQP's warm_start = 1 between two solve() calls of one workspace.
int solve_alm(int nv, int nc, bool diag, const double *H, const double *c, const double *Acm, const double *b,
              const double *lb, const double *ub, const bool *be, double eps_rel, int max_newton, double *sol, int *newton_out,
              double *warm_x, double *warm_y, bool warm_in)
{
	const int n = nv, m = nc + nv;
	std::vector<double> P((size_t)n * n, 0.0), q(c, c + n), A((size_t)m * n, 0.0), l(m), u(m);
	for (int j = 0; j < n; j++)
		for (int i = 0; i < n; i++) {
			const double h = diag ? (i == j ? H[i + (size_t)j * n] : 0.0) : (i <= j ? H[i + (size_t)j * n] : H[j + (size_t)i * n]);
			P[(size_t)i * n + j] = 2.0 * h; // the upper triangle is what the wrapper hands on (:136-153,276-309)
		}
	for (int i = 0; i < nc; i++) {
		for (int j = 0; j < n; j++) A[(size_t)i * n + j] = Acm[i + (size_t)j * nc];
		l[i] = b[i];
		u[i] = (be && be[i]) ? b[i] : kInfty;
	}
	for (int j = 0; j < n; j++) {
		A[(size_t)(nc + j) * n + j] = 1.0;
		l[nc + j] = lb[j];
		u[nc + j] = ub[j];
	}
	// data outside the solvers' domain (qp_lane.hpp: qp_data_nonfinite): the reference's solver runs it to max_iter
	{
		double dom = 0.0;
		bool nanb = false;
		for (double v : P) dom = std::fma(v, 1e160, dom);
		for (double v : q) dom = std::fma(v, 1e160, dom);
		for (int i = 0; i < nc; i++) {
			for (int j = 0; j < n; j++) dom = std::fma(A[(size_t)i * n + j], 1e160, dom);
			dom = std::fma(b[i], 1e160, dom);
		}
		for (int j = 0; j < n; j++) nanb = nanb || lb[j] != lb[j] || ub[j] != ub[j];
		if (nanb || !(std::fabs(dom) < HUGE_VAL)) {
			for (int j = 0; j < n; j++) sol[j] = 0.0;
			if (newton_out) *newton_out = 0;
			if (warm_x && warm_y) {
				for (int j = 0; j < n; j++) warm_x[j] = 0.0;
				for (int i = 0; i < m; i++) warm_y[i] = 0.0;
			}
			return -2;
		}
	}
	// power-of-two Ruiz equilibration of [P A'; A 0] and cost normalisation (qp_lds.hpp: scale), four passes
	std::vector<double> D(n, 1.0), E(m, 1.0);
	double cs = 1.0; // the cost's own scale factor: the multipliers' unit
	for (int it = 0; it < 4; it++) {
		std::vector<double> Dt(n), Et(m);
		for (int j = 0; j < n; j++) {
			double v = 0.0;
			for (int i = 0; i < n; i++) v = std::max(v, std::fabs(P[(size_t)i * n + j]));
			for (int i = 0; i < m; i++) v = std::max(v, std::fabs(A[(size_t)i * n + j]));
			Dt[j] = pow2_rsqrt(limit_scaling(v));
		}
		for (int i = 0; i < m; i++) {
			double v = 0.0;
			for (int j = 0; j < n; j++) v = std::max(v, std::fabs(A[(size_t)i * n + j]));
			Et[i] = pow2_rsqrt(limit_scaling(v));
		}
		for (int i = 0; i < n; i++)
			for (int j = 0; j < n; j++) P[(size_t)i * n + j] *= Dt[i] * Dt[j];
		for (int i = 0; i < m; i++)
			for (int j = 0; j < n; j++) A[(size_t)i * n + j] *= Et[i] * Dt[j];
		double cm = 0.0, qn = 0.0;
		for (int j = 0; j < n; j++) {
			q[j] *= Dt[j];
			D[j] *= Dt[j];
			double colmax = 0.0;
			for (int i = 0; i < n; i++) colmax = std::max(colmax, std::fabs(P[(size_t)i * n + j]));
			cm += colmax;
			qn = std::max(qn, std::fabs(q[j]));
		}
		for (int i = 0; i < m; i++) E[i] *= Et[i];
		const double ct = pow2_floor_inv(limit_scaling(std::max(cm / n, limit_scaling(qn))));
		for (double &v : P) v *= ct;
		for (double &v : q) v *= ct;
		cs *= ct;
	}
	for (int i = 0; i < m; i++) {
		l[i] *= E[i];
		u[i] *= E[i];
	}
	const double tol = std::max(eps_rel, 1e-10) * 1e-2, ig = 1.0 / kGamma, big = kInfty * kMinScaling;
	std::vector<double> x(n, 0.0), xh(n, 0.0), y(m, 0.0), mu(m), s(m), r(m), g(n), d(n), dl(m), K((size_t)n * n), Ax(m), ynew(m);
	for (int i = 0; i < m; i++) mu[i] = (u[i] - l[i] < kRhoTol) ? 100.0 * kMu0 : kMu0;
	std::vector<char> J(m, 0), Jf(m, 0);
	const bool warm = warm_in && warm_x && warm_y;
	if (warm) {
		auto sane = [](double v) { return std::fabs(v) < 1e100 ? v : 0.0; }; // NaN, inf, nonsense: no start
		for (int j = 0; j < n; j++) xh[j] = x[j] = sane(warm_x[j]) / D[j];
		for (int i = 0; i < m; i++) y[i] = sane(warm_y[i]) * cs / E[i];
	}
	int met_in_a_row = 0;
	bool have_factor = false;
	int newton = 0, status = 0;
	double pri_prev = -1.0, best_res = 1e300;
	auto mulA = [&](const std::vector<double> &v, std::vector<double> &out) {
		for (int i = 0; i < m; i++) {
			double t = 0.0;
			for (int j = 0; j < n; j++) t += A[(size_t)i * n + j] * v[j];
			out[i] = t;
		}
	};
	for (int outer = 0; outer < kMaxOuter && status == 0; outer++) {
		double gfloor = 0.0, gscale = 1.0;
		for (int inner = 0; inner < kMaxInner; inner++) {
			mulA(x, Ax);
			double gn = 0.0;
			for (int i = 0; i < m; i++) {
				s[i] = Ax[i] + y[i] / mu[i];
				r[i] = mu[i] * (s[i] - clip(s[i], l[i], u[i]));
				J[i] = (s[i] < l[i] || s[i] > u[i]) ? 1 : 0;
			}
			double gs = 0.0, fl = 0.0;
			for (int j = 0; j < n; j++) {
				double px = 0.0, pxa = 0.0, atr = 0.0, atra = 0.0, flj = 0.0;
				for (int k = 0; k < n; k++) {
					px += P[(size_t)j * n + k] * x[k];
					pxa += std::fabs(P[(size_t)j * n + k] * x[k]);
				}
				for (int i = 0; i < m; i++) {
					const double a = A[(size_t)i * n + j];
					atr += a * r[i];
					atra += std::fabs(a * r[i]);
					const double bd = s[i] < l[i] ? std::fabs(l[i]) : (s[i] > u[i] ? std::fabs(u[i]) : 0.0);
					flj += std::fabs(a) * 2.2e-16 * mu[i] * (std::fabs(Ax[i]) + std::fabs(y[i]) / mu[i] + bd);
				}
				g[j] = px + q[j] + (x[j] - xh[j]) * ig + atr;
				gn = std::max(gn, std::fabs(g[j]));
				gs = std::max(gs, std::max(pxa, std::max(std::fabs(q[j]), atra)));
				fl = std::max(fl, flj);
			}
			if (inner == 0) { // scale of the gradient's own terms and its rounding floor, once per inner solve
				gscale = 1.0 + gs;
				gfloor = fl + 2.2e-16 * gscale;
			}
			if (gn <= 0.1 * tol * gscale || gn <= 8.0 * gfloor || newton >= max_newton) break;
			if (!have_factor || J != Jf) {
				for (int i = 0; i < n; i++)
					for (int j = 0; j <= i; j++) K[(size_t)i * n + j] = P[(size_t)i * n + j] + (i == j ? ig : 0.0);
				for (int k = 0; k < m; k++)
					if (J[k]) {
						const double *a = &A[(size_t)k * n];
						for (int i = 0; i < n; i++) {
							if (a[i] == 0.0) continue;
							const double t = mu[k] * a[i];
							for (int j = 0; j <= i; j++) K[(size_t)i * n + j] += t * a[j];
						}
					}
				if (!cholesky(K, n)) { // (a cost that is not positive semidefinite: not this wrapper's problem class)
					status = -2;
					break;
				}
				Jf = J;
				have_factor = true;
			}
			for (int j = 0; j < n; j++) d[j] = -g[j];
			chol_solve(K, n, d);
			newton++;
			// exact line search: phi'(t) = a0 + t a1 + sum_i mu_i dl_i (s_i + t dl_i - proj(s_i + t dl_i) - (s_i - proj s_i))
			mulA(d, dl);
			double a0 = 0.0, a1 = 0.0;
			for (int j = 0; j < n; j++) {
				double pd = 0.0;
				for (int k = 0; k < n; k++) pd += P[(size_t)j * n + k] * d[k];
				a0 += g[j] * d[j];
				a1 += d[j] * (pd + ig * d[j]);
			}
			auto dphi = [&](double t) {
				double f = a0 + t * a1;
				for (int i = 0; i < m; i++) {
					const double st = s[i] + t * dl[i];
					f += mu[i] * dl[i] * ((st - clip(st, l[i], u[i])) - (s[i] - clip(s[i], l[i], u[i])));
				}
				return f;
			};
			double t = 1.0;
			if (!(dphi(1.0) < 0.0)) {
				std::vector<double> bp;
				for (int i = 0; i < m; i++) {
					if (dl[i] == 0.0) continue;
					const double t1 = (l[i] - s[i]) / dl[i], t2 = (u[i] - s[i]) / dl[i];
					if (t1 > 0.0 && t1 <= 1.0) bp.push_back(t1);
					if (t2 > 0.0 && t2 <= 1.0) bp.push_back(t2);
				}
				bp.push_back(1.0);
				std::sort(bp.begin(), bp.end());
				double tlo = 0.0, flo = dphi(0.0);
				t = 1.0;
				if (!(flo < 0.0)) { // not a descent direction (never seen with an exact factor): no step, refactor
					t = 0.0;
					have_factor = false;
				} else {
					for (double tb : bp) {
						const double fb = dphi(tb);
						if (fb >= 0.0) {
							t = fb > flo ? tlo - flo * (tb - tlo) / (fb - flo) : tlo;
							break;
						}
						tlo = tb;
						flo = fb;
					}
				}
			}
			for (int j = 0; j < n; j++) x[j] += t * d[j];
		}
		if (status != 0) break;
		// ---- multiplier update, residuals, certificates
		mulA(x, Ax);
		double pri = 0.0, nax = 0.0, ndy = 0.0, lhs = 0.0;
		std::vector<double> v(m);
		for (int i = 0; i < m; i++) {
			const double sv = Ax[i] + y[i] / mu[i];
			ynew[i] = mu[i] * (sv - clip(sv, l[i], u[i]));
			pri = std::max(pri, std::fabs(Ax[i] - clip(Ax[i], l[i], u[i])));
			nax = std::max(nax, std::fabs(Ax[i]));
			double vi = ynew[i] - y[i];
			if (u[i] > big) vi = (l[i] < -big) ? 0.0 : std::min(vi, 0.0);
			else if (l[i] < -big) vi = std::max(vi, 0.0);
			v[i] = vi;
			ndy = std::max(ndy, std::fabs(vi));
			lhs += vi > 0.0 ? u[i] * vi : (vi < 0.0 ? l[i] * vi : 0.0);
		}
		double dua = 0.0, nd = 0.0, natv = 0.0;
		for (int j = 0; j < n; j++) {
			double px = 0.0, aty = 0.0, atya = 0.0, atv = 0.0;
			for (int k = 0; k < n; k++) px += P[(size_t)j * n + k] * x[k];
			for (int i = 0; i < m; i++) {
				const double a = A[(size_t)i * n + j];
				aty += a * ynew[i];
				atya += std::fabs(a * ynew[i]);
				atv += a * v[i];
			}
			dua = std::max(dua, std::fabs(px + q[j] + aty));
			nd = std::max(nd, std::max(std::fabs(px), std::max(std::fabs(q[j]), atya)));
			natv = std::max(natv, std::fabs(atv));
		}
		y = ynew;
		xh = x;
		const double rp = pri / (1.0 + nax), rd = dua / (1.0 + nd);
		best_res = std::min(best_res, std::max(rp, rd));
		// From a cold start the test is first met after the multipliers have gone through at least one update on a settled
		// active set, and the point it accepts is far better than the test asks (|u - u_ref| 4e-9 ... 1e-7).  A warm start
		// can meet it at once, at a point that is only as good as the relative test (1e-10 of multipliers of size 1e4-1e5:
		// 2e-5 in u on the realizable filter's problems): a warm solve is done when three updates in a row meet it
		// (two: 7e-7; three: 6e-8, the cold start's own accuracy, for 0.3-0.6 Newton steps more).
		met_in_a_row = (rp <= tol && rd <= tol) ? met_in_a_row + 1 : 0;
		if (met_in_a_row >= (warm ? 3 : 1)) status = 1;
		else if (ndy > 1e-4 && lhs < -1e-6 * ndy && natv < 1e-6 * ndy) status = -3;
		else if (newton >= max_newton) status = -2;
		if (status == 0) {
			const double mumin = *std::min_element(mu.begin(), mu.end()), mumax = *std::max_element(mu.begin(), mu.end());
			double f = 1.0, cap = kMuMax;
			if (rp <= tol) {
				if (mumax > 100.0 * kMu0) f = 0.1;
			} else if (pri_prev >= 0.0 && pri > 0.5 * pri_prev && mumin >= kMuMax) {
				f = 10.0;
				cap = 1e8;
			} else if (pri_prev >= 0.0 && pri > 0.1 * pri_prev) {
				f = penalty_jump(pri, pri_prev);
			}
			if (f != 1.0) {
				for (double &mi : mu) mi = f > 1.0 ? std::min(mi * f, std::max(mi, cap)) : std::max(mi * f, kMu0);
				have_factor = false;
			}
			pri_prev = pri;
		}
	}
	if (status == 0 || status == -2) status = best_res <= 1e3 * tol ? 1 : -2; // OSQP's "solved inaccurate" counts as solved (:225)
	for (int j = 0; j < n; j++) sol[j] = D[j] * x[j];
	if (newton_out) *newton_out = newton;
	if (warm_x && warm_y) { // a problem without a solution leaves a cold start behind
		for (int j = 0; j < n; j++) warm_x[j] = status == 1 ? D[j] * x[j] : 0.0;
		for (int i = 0; i < m; i++) warm_y[i] = status == 1 ? E[i] * y[i] / cs : 0.0;
	}
	return status;
}

} // namespace hostqp
} // namespace ASIF
