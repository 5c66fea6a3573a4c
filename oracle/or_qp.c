/*
 * or_qp.c -- the two CPU QP solvers of the oracle.  TEST INFRASTRUCTURE (see or_oracle.h).
 *
 *  (1) or_qp_exact_small: exact optimum by active-set enumeration in long double.  This is
 *      u_ref of the parity metric (SURVEY 8c "Resolution of max|u*-u_ref|").
 *  (2) or_qp_admm: the reference's solve path restated: QPWrapperOsqp's form translation
 *      (src/qpwrapper_osqp.cpp:263-376) followed by an OSQP-0.6-style ADMM.
 *      OSQP is a third-party dependency that is NOT in /root/reference and NOT in this image
 *      (CMake package "osqp", no version pinned; the API used pins the 0.6.x line).  Its
 *      algorithm is restated from the published description (Stellato, Banjac, Goulart,
 *      Bemporad, Boyd: "OSQP: an operator splitting solver for quadratic programs", Math. Prog.
 *      Comp. 2020) and from memory of the 0.6.x defaults.  PARITY UNPINNED at this boundary:
 *      no reference test, golden vector or runnable OSQP exists to check it against.
 */
#include "or_internal.h"
#include <math.h>
#include <string.h>

/* =============================================================== exact solver */
typedef struct {
	long double a[3];
	long double r;
	int eq;
} cand_t;

/* Solve k x k (k<=3) by Gaussian elimination with partial pivoting; 0 if singular. */
static int solve_small(int k, long double G[3][3], long double *d, long double *mu)
{
	long double M[3][4];
	long double scale = 0;
	for (int i = 0; i < k; i++)
		for (int j = 0; j < k; j++) {
			M[i][j] = G[i][j];
			if (fabsl(G[i][j]) > scale) scale = fabsl(G[i][j]);
		}
	for (int i = 0; i < k; i++) M[i][k] = d[i];
	if (scale == 0) return 0;
	for (int c = 0; c < k; c++) {
		int p = c;
		for (int i = c + 1; i < k; i++)
			if (fabsl(M[i][c]) > fabsl(M[p][c])) p = i;
		if (fabsl(M[p][c]) < 1e-15L * scale) return 0;
		if (p != c)
			for (int j = 0; j <= k; j++) {
				long double t = M[c][j];
				M[c][j] = M[p][j];
				M[p][j] = t;
			}
		for (int i = c + 1; i < k; i++) {
			long double fct = M[i][c] / M[c][c];
			for (int j = c; j <= k; j++) M[i][j] -= fct * M[c][j];
		}
	}
	for (int i = k - 1; i >= 0; i--) {
		long double s = M[i][k];
		for (int j = i + 1; j < k; j++) s -= M[i][j] * mu[j];
		mu[i] = s / M[i][i];
	}
	return 1;
}

/* Try working set W (indices into cand); 1 if it is the optimum (x filled). */
static int try_set(int nv, const long double *Dinv, const long double *c, const cand_t *cand, int ncand,
                   const int *W, int k, long double *x)
{
	long double G[3][3], d[3], mu[3] = {0, 0, 0};
	for (int p = 0; p < k; p++) {
		const cand_t *cp = &cand[W[p]];
		long double s = cp->r;
		for (int j = 0; j < nv; j++) s += cp->a[j] * Dinv[j] * c[j];
		d[p] = s;
		for (int q = 0; q < k; q++) {
			long double g = 0;
			for (int j = 0; j < nv; j++) g += cp->a[j] * Dinv[j] * cand[W[q]].a[j];
			G[p][q] = g;
		}
	}
	if (k > 0 && !solve_small(k, G, d, mu)) return 0;
	for (int p = 0; p < k; p++)
		if (!cand[W[p]].eq && mu[p] < -1e-13L * (1 + fabsl(mu[p]))) return 0;
	for (int j = 0; j < nv; j++) {
		long double s = -c[j];
		for (int p = 0; p < k; p++) s += cand[W[p]].a[j] * mu[p];
		x[j] = Dinv[j] * s;
	}
	for (int i = 0; i < ncand; i++) {
		long double ax = 0, mag = fabsl(cand[i].r);
		for (int j = 0; j < nv; j++) {
			ax += cand[i].a[j] * x[j];
			mag += fabsl(cand[i].a[j] * x[j]);
		}
		const long double viol = cand[i].r - ax; /* want <= 0 (== 0 for eq) */
		const long double tol = 1e-12L * (1 + mag);
		if (viol > tol) return 0;
		if (cand[i].eq && viol < -tol) return 0;
	}
	return 1;
}

int or_qp_exact_small(const or_qp *qp, double *xout)
{
	const int nv = qp->nv, nc = qp->nc;
	if (nv < 1 || nv > 3 || nc > 250) return -100;
	/* Data outside the domain: NaN, infinite, or beyond 1e148 in magnitude (a NaN / inf / overflowing state reaches the
	 * rows through h, Lfh, Lgh).  OSQP 0.6 validates only l <= u at set-up, osqp_update_A / _lin_cost take the values
	 * as they are, and every residual comparison of its termination test is false on NaN -- it runs to max_iter and
	 * QPWrapperOsqp::solve hands back the raw status (src/qpwrapper_osqp.cpp:225-238); on 1e150-sized entries its
	 * double-precision iterates overflow into the same state.  The exact solver stands in for it with that verdict;
	 * filter() then returns -1 with uAct untouched (src/asif.cpp:199-209).  Same rule on the device (qp_lane.hpp). */
#define OR_OUT_OF_DOMAIN(v) (!(fabs(v) <= 1e148))
	for (int j = 0; j < nv; j++) {
		if (OR_OUT_OF_DOMAIN(qp->Hd[j]) || OR_OUT_OF_DOMAIN(qp->c[j]) || isnan(qp->lb[j]) || isnan(qp->ub[j]))
			return OR_OSQP_MAX_ITER_REACHED;
		for (int i = 0; i < nc; i++)
			if (OR_OUT_OF_DOMAIN(qp->A[i + j * nc])) return OR_OSQP_MAX_ITER_REACHED;
	}
	for (int i = 0; i < nc; i++)
		if (OR_OUT_OF_DOMAIN(qp->b[i])) return OR_OSQP_MAX_ITER_REACHED;
	cand_t cand[256 + 8];
	int ncand = 0;
	long double Dinv[3], c[3], x[3];
	for (int j = 0; j < nv; j++) {
		if (!(qp->Hd[j] > 0)) return -100;
		Dinv[j] = 1.0L / (2.0L * qp->Hd[j]);
		c[j] = qp->c[j];
	}
	int eqs[8], neq = 0, ineq[264], nineq = 0;
	for (int i = 0; i < nc; i++) {
		cand_t *cp = &cand[ncand];
		for (int j = 0; j < nv; j++) cp->a[j] = qp->A[i + j * nc];
		cp->r = qp->b[i];
		cp->eq = qp->be ? qp->be[i] : 0;
		if (cp->eq) { if (neq < 8) eqs[neq++] = ncand; } else ineq[nineq++] = ncand;
		ncand++;
	}
	for (int j = 0; j < nv; j++) {
		if (qp->lb[j] == qp->ub[j]) {
			cand_t *cp = &cand[ncand];
			for (int q = 0; q < nv; q++) cp->a[q] = (q == j);
			cp->r = qp->lb[j];
			cp->eq = 1;
			if (neq < 8) eqs[neq++] = ncand;
			ncand++;
		} else {
			cand_t *cp = &cand[ncand];
			for (int q = 0; q < nv; q++) cp->a[q] = (q == j);
			cp->r = qp->lb[j];
			cp->eq = 0;
			ineq[nineq++] = ncand++;
			cp = &cand[ncand];
			for (int q = 0; q < nv; q++) cp->a[q] = -(long double)(q == j);
			cp->r = -(long double)qp->ub[j];
			cp->eq = 0;
			ineq[nineq++] = ncand++;
		}
	}
	if (neq > nv) return -100; /* not produced by any variant */
	int W[3];
	for (int p = 0; p < neq; p++) W[p] = eqs[p];
	const int room = nv - neq;
	/* subsets of the inequalities of size 0..room, smallest first */
	if (try_set(nv, Dinv, c, cand, ncand, W, neq, x)) goto found;
	if (room >= 1)
		for (int i0 = 0; i0 < nineq; i0++) {
			W[neq] = ineq[i0];
			if (try_set(nv, Dinv, c, cand, ncand, W, neq + 1, x)) goto found;
		}
	if (room >= 2)
		for (int i0 = 0; i0 < nineq; i0++)
			for (int i1 = i0 + 1; i1 < nineq; i1++) {
				W[neq] = ineq[i0];
				W[neq + 1] = ineq[i1];
				if (try_set(nv, Dinv, c, cand, ncand, W, neq + 2, x)) goto found;
			}
	if (room >= 3)
		for (int i0 = 0; i0 < nineq; i0++)
			for (int i1 = i0 + 1; i1 < nineq; i1++)
				for (int i2 = i1 + 1; i2 < nineq; i2++) {
					W[neq] = ineq[i0];
					W[neq + 1] = ineq[i1];
					W[neq + 2] = ineq[i2];
					if (try_set(nv, Dinv, c, cand, ncand, W, neq + 3, x)) goto found;
				}
	return 0; /* a strictly convex QP over a nonempty closed set always has a KKT point: infeasible */
found:
	for (int j = 0; j < nv; j++) xout[j] = (double)x[j];
	return 1;
}

/* ================================================================ OSQP-style ADMM */
#define OSQP_INFTY 1e30
#define MIN_SCALING 1e-04
#define MAX_SCALING 1e+04
#define RHO_MIN 1e-06
#define RHO_MAX 1e+06
#define RHO_EQ_OVER_RHO_INEQ 1e+03
#define RHO_TOL 1e-04
#define MAXN 96  /* realizable filter on the 10 Hz kernel: n = 86, m = 65 + 86 */
#define MAXM 160

void or_admm_default_settings(or_admm_settings *s)
{
	s->rho = 0.1;
	s->sigma = 1e-6;
	s->alpha = 1.6;
	s->eps_abs = 1e-3;
	s->eps_rel = 1e-3;
	s->eps_prim_inf = 1e-4;
	s->eps_dual_inf = 1e-4;
	s->scaling = 10;
	s->adaptive_rho = 1;
	/* OSQP's default 0 means "choose from setup/solve wall-clock" (timing dependent);
	 * for problems this small that rule lands on its floor, one check_termination period. */
	s->adaptive_rho_interval = 25;
	s->adaptive_rho_tolerance = 5.0;
	s->check_termination = 25;
	s->max_iter = 2000; /* src/qpwrapper_osqp.cpp:69 */
	s->reduced_kkt = 0;
	s->polish = 0;
	s->scaling_pow2 = 0;
}

typedef struct {
	int n, m;
	double P[MAXN];        /* diagonal of P (scaled) */
	double q[MAXN];
	double A[MAXM * MAXN]; /* row-major m x n (scaled) */
	double l[MAXM], u[MAXM];
	double D[MAXN], E[MAXM], Dinv[MAXN], Einv[MAXM], c, cinv;
	double rho, rho_vec[MAXM], rho_inv[MAXM];
	int ctype[MAXM];
	/* factorisation */
	int dim;
	double L[(MAXN + MAXM) * (MAXN + MAXM)];
	double Dg[MAXN + MAXM];
} ws_t;

static double limit_scaling(double v)
{
	v = v < MIN_SCALING ? 1.0 : v;
	v = v > MAX_SCALING ? MAX_SCALING : v;
	return v;
}

/* Ruiz equilibration of the KKT matrix + cost normalisation (OSQP scaling.c, scale_data) */
static double inv_sqrt_scale(double v, int pow2)
{
	if (!pow2) return 1.0 / sqrt(v);
	/* power-of-two stand-in for 1/sqrt(v): exact to apply and to undo (device scaling) */
	int e;
	frexp(v, &e); /* v = f * 2^e, f in [0.5,1) */
	return ldexp(1.0, -(e >> 1));
}

static void scale_data(ws_t *w, int iters, int pow2)
{
	const int n = w->n, m = w->m;
	for (int j = 0; j < n; j++) w->D[j] = 1.0;
	for (int i = 0; i < m; i++) w->E[i] = 1.0;
	w->c = 1.0;
	double Dt[MAXN], Et[MAXM];
	for (int it = 0; it < iters; it++) {
		for (int j = 0; j < n; j++) {
			double v = fabs(w->P[j]);
			for (int i = 0; i < m; i++) {
				const double a = fabs(w->A[i * n + j]);
				if (a > v) v = a;
			}
			Dt[j] = inv_sqrt_scale(limit_scaling(v), pow2);
		}
		for (int i = 0; i < m; i++) {
			double v = 0;
			for (int j = 0; j < n; j++) {
				const double a = fabs(w->A[i * n + j]);
				if (a > v) v = a;
			}
			Et[i] = inv_sqrt_scale(limit_scaling(v), pow2);
		}
		for (int j = 0; j < n; j++) {
			w->P[j] = Dt[j] * w->P[j] * Dt[j];
			w->q[j] = Dt[j] * w->q[j];
			w->D[j] *= Dt[j];
		}
		for (int i = 0; i < m; i++) {
			for (int j = 0; j < n; j++) w->A[i * n + j] = Et[i] * w->A[i * n + j] * Dt[j];
			w->E[i] *= Et[i];
		}
		double cmean = 0, qn = 0;
		for (int j = 0; j < n; j++) {
			cmean += fabs(w->P[j]);
			if (fabs(w->q[j]) > qn) qn = fabs(w->q[j]);
		}
		cmean /= n;
		qn = limit_scaling(qn);
		double ct = cmean > qn ? cmean : qn;
		ct = 1.0 / limit_scaling(ct);
		if (pow2) {
			int e;
			frexp(ct, &e);
			ct = ldexp(1.0, e - 1);
		}
		for (int j = 0; j < n; j++) {
			w->P[j] *= ct;
			w->q[j] *= ct;
		}
		w->c *= ct;
	}
	for (int j = 0; j < n; j++) w->Dinv[j] = 1.0 / w->D[j];
	for (int i = 0; i < m; i++) {
		w->Einv[i] = 1.0 / w->E[i];
		w->l[i] *= w->E[i];
		w->u[i] *= w->E[i];
	}
	w->cinv = 1.0 / w->c;
}

static void set_rho_vec(ws_t *w)
{
	for (int i = 0; i < w->m; i++) {
		if (w->l[i] < -OSQP_INFTY * MIN_SCALING && w->u[i] > OSQP_INFTY * MIN_SCALING) {
			w->ctype[i] = -1;
			w->rho_vec[i] = RHO_MIN;
		} else if (w->u[i] - w->l[i] < RHO_TOL) {
			w->ctype[i] = 1;
			w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->rho;
		} else {
			w->ctype[i] = 0;
			w->rho_vec[i] = w->rho;
		}
		w->rho_inv[i] = 1.0 / w->rho_vec[i];
	}
}

/* Dense LDL' without pivoting (what QDLDL computes on the quasi-definite KKT; symmetric
 * quasi-definite matrices are strongly factorisable). K is built in place in L. */
static int factor(ws_t *w, double sigma, int reduced)
{
	const int n = w->n, m = w->m;
	const int dim = reduced ? n : n + m;
	w->dim = dim;
	double *K = w->L;
	memset(K, 0, sizeof(double) * dim * dim);
	if (reduced) {
		for (int a = 0; a < n; a++) {
			for (int b = 0; b <= a; b++) {
				double s = 0;
				for (int i = 0; i < m; i++) s += w->rho_vec[i] * w->A[i * n + a] * w->A[i * n + b];
				K[a * dim + b] = s;
			}
			K[a * dim + a] += w->P[a] + sigma;
		}
	} else {
		for (int a = 0; a < n; a++) K[a * dim + a] = w->P[a] + sigma;
		for (int i = 0; i < m; i++) {
			for (int j = 0; j < n; j++) K[(n + i) * dim + j] = w->A[i * n + j];
			K[(n + i) * dim + (n + i)] = -w->rho_inv[i];
		}
	}
	/* in-place LDL' on the lower triangle */
	for (int j = 0; j < dim; j++) {
		double d = K[j * dim + j];
		for (int k = 0; k < j; k++) d -= K[j * dim + k] * K[j * dim + k] * w->Dg[k];
		if (d == 0.0) return -1;
		w->Dg[j] = d;
		for (int i = j + 1; i < dim; i++) {
			double s = K[i * dim + j];
			for (int k = 0; k < j; k++) s -= K[i * dim + k] * K[j * dim + k] * w->Dg[k];
			K[i * dim + j] = s / d;
		}
	}
	return 0;
}

static void ldl_solve(const ws_t *w, double *v)
{
	const int dim = w->dim;
	const double *L = w->L;
	for (int i = 0; i < dim; i++) {
		double s = v[i];
		for (int k = 0; k < i; k++) s -= L[i * dim + k] * v[k];
		v[i] = s;
	}
	for (int i = 0; i < dim; i++) v[i] /= w->Dg[i];
	for (int i = dim - 1; i >= 0; i--) {
		double s = v[i];
		for (int k = i + 1; k < dim; k++) s -= L[k * dim + i] * v[k];
		v[i] = s;
	}
}

static double norm_inf(const double *v, int n)
{
	double r = 0;
	for (int i = 0; i < n; i++)
		if (fabs(v[i]) > r) r = fabs(v[i]);
	return r;
}
static double scaled_norm_inf(const double *s, const double *v, int n)
{
	double r = 0;
	for (int i = 0; i < n; i++)
		if (fabs(s[i] * v[i]) > r) r = fabs(s[i] * v[i]);
	return r;
}

/* ---- Active-set finish, tried at a termination check (device algorithm study; the reference
 * configuration has OSQP's polish off).  Three pieces, all in the scaled space and in the primal
 * Schur form of the KKT system:
 *   as_solve   : equality-constrained QP on a working set by the regularised system + refinement
 *                (OSQP polish.c idea), then classification of every row at that point;
 *   as_farkas  : least-squares point of the working rows; if every residual has the sign of a
 *                violation the residual vector is an exact Farkas certificate of infeasibility;
 *   try_polish : working set guessed from (z,y), then a few primal-dual active-set corrections
 *                (add violated rows, drop wrongly signed multipliers).  A result is ACCEPTED ONLY if
 *                it satisfies the KKT conditions of the full problem (returns 1) or carries a valid
 *                certificate (returns 2); otherwise ADMM simply continues (returns 0). */
static int small_ldl(int n, double *M, double *Dg)
{
	for (int j = 0; j < n; j++) {
		double d = M[j * n + j];
		for (int k = 0; k < j; k++) d -= M[j * n + k] * M[j * n + k] * Dg[k];
		if (!(d > 0)) return 0;
		Dg[j] = d;
		for (int i = j + 1; i < n; i++) {
			double sacc = M[i * n + j];
			for (int k = 0; k < j; k++) sacc -= M[i * n + k] * M[j * n + k] * Dg[k];
			M[i * n + j] = sacc / d;
		}
	}
	return 1;
}
static void small_ldl_solve(int n, const double *M, const double *Dg, double *v)
{
	for (int i = 0; i < n; i++) {
		double sacc = v[i];
		for (int k = 0; k < i; k++) sacc -= M[i * n + k] * v[k];
		v[i] = sacc;
	}
	for (int i = 0; i < n; i++) v[i] /= Dg[i];
	for (int i = n - 1; i >= 0; i--) {
		double sacc = v[i];
		for (int k = i + 1; k < n; k++) sacc -= M[k * n + i] * v[k];
		v[i] = sacc;
	}
}

static int g_as_refine = 3; /* refinement steps of the regularised solve (device default) */

/* returns 1 valid KKT point; 0 otherwise with viol[i] (-1 below l, +1 above u) and wrong[i] filled */
static __thread double g_as_delta = 1e-6; /* 1/penalty of the working-set solve; try_polish lowers it on stalls */
static int as_solve(const ws_t *w, const int *act, const double *r, double *x, double *nu, int *viol, int *wrong,
                    double *vmag, double ktol)
{
	const int n = w->n, m = w->m;
	const double delta = g_as_delta;
	double M[MAXN * MAXN], Dg[MAXN], e2[MAXM], rhs[MAXN];
	for (int a = 0; a < n; a++)
		for (int b = 0; b <= a; b++) {
			double sacc = 0;
			for (int i = 0; i < m; i++)
				if (act[i]) sacc += w->A[i * n + a] * w->A[i * n + b];
			/* primal regularisation only where the cost has no curvature of its own */
			M[a * n + b] = sacc / delta + (a == b ? (w->P[a] > 0 ? w->P[a] : 1e-6) : 0.0);
		}
	for (int i = 0; i < m; i++) { nu[i] = 0; viol[i] = 0; wrong[i] = 0; vmag[i] = 0; }
	for (int j = 0; j < n; j++) x[j] = 0;
	if (!small_ldl(n, M, Dg)) return 0;
	for (int it = 0; it < g_as_refine; it++) {
		for (int j = 0; j < n; j++) rhs[j] = -w->q[j] - w->P[j] * x[j];
		for (int i = 0; i < m; i++) {
			if (!act[i]) continue;
			double ax = 0;
			for (int j = 0; j < n; j++) ax += w->A[i * n + j] * x[j];
			e2[i] = r[i] - ax;
			for (int j = 0; j < n; j++) rhs[j] += w->A[i * n + j] * (e2[i] / delta - nu[i]);
		}
		small_ldl_solve(n, M, Dg, rhs);
		for (int i = 0; i < m; i++) {
			if (!act[i]) continue;
			double adx = 0;
			for (int j = 0; j < n; j++) adx += w->A[i * n + j] * rhs[j];
			nu[i] += (adx - e2[i]) / delta;
		}
		for (int j = 0; j < n; j++) x[j] += rhs[j];
	}
	int bad = 0;
	for (int i = 0; i < m; i++) {
		double ax = 0;
		for (int j = 0; j < n; j++) ax += w->A[i * n + j] * x[j];
		const double tol = ktol * (1 + fabs(ax));
		if (ax < w->l[i] - tol) { viol[i] = -1; bad = 1; vmag[i] = w->l[i] - ax; }
		else if (ax > w->u[i] + tol) { viol[i] = 1; bad = 1; vmag[i] = ax - w->u[i]; }
		const double nt = ktol * (1 + fabs(nu[i]));
		if ((act[i] == -1 && nu[i] > nt) || (act[i] == 1 && nu[i] < -nt)) { wrong[i] = 1; bad = 1; }
	}
	/* stationarity, relative to the size of its own terms (a column can be scaled very small) */
	for (int j = 0; j < n; j++) {
		double g = w->P[j] * x[j] + w->q[j];
		double mag = fabs(w->P[j] * x[j]) + fabs(w->q[j]);
		for (int i = 0; i < m; i++)
			if (act[i]) {
				g += w->A[i * n + j] * nu[i];
				mag += fabs(w->A[i * n + j] * nu[i]);
			}
		if (fabs(g) > 1e-10 * mag + 1e-300) bad = 1;
	}
	return !bad;
}

/* Infeasibility test = phase 1 of the problem: minimise the one-sided least squares
 * sum_i max(0, violation_i(x))^2 by working-set iterations (least-squares point of the rows in S,
 * then S := rows violated there, equalities always in).  At a stationary S every residual has the
 * sign of a violation, A_S' v = 0 by construction and sum r_i v_i = |v|^2 > 0: v is a Farkas
 * certificate as soon as it is not (numerically) zero.  `act0` seeds S. */
static int g_farkas_iters = 2;
static int as_farkas(const ws_t *w, const int *act0, const double *r0)
{
	const int n = w->n, m = w->m;
	int act[MAXM];
	double r[MAXM], MS[MAXN * MAXN], DS[MAXN], wv[MAXN], v[MAXM];
	for (int i = 0; i < m; i++) { act[i] = act0[i]; r[i] = r0[i]; }
	for (int iter = 0;; iter++) {
		double dmax = 0;
		for (int a = 0; a < n; a++)
			for (int b = 0; b <= a; b++) {
				double sacc = 0;
				for (int i = 0; i < m; i++)
					if (act[i]) sacc += w->A[i * n + a] * w->A[i * n + b];
				MS[a * n + b] = sacc;
				if (a == b && sacc > dmax) dmax = sacc;
			}
		for (int a = 0; a < n; a++) MS[a * n + a] += 1e-14 * dmax + 1e-300;
		if (!small_ldl(n, MS, DS)) return 0;
		for (int j = 0; j < n; j++) {
			double sacc = 0;
			for (int i = 0; i < m; i++)
				if (act[i]) sacc += w->A[i * n + j] * r[i];
			wv[j] = sacc;
		}
		small_ldl_solve(n, MS, DS, wv);
		{ /* one refinement step removes the footprint of the 1e-14 regularisation from A_S' v */
			double g2[MAXN];
			for (int j = 0; j < n; j++) g2[j] = 0;
			for (int i = 0; i < m; i++) {
				if (!act[i]) continue;
				double ax = 0;
				for (int j = 0; j < n; j++) ax += w->A[i * n + j] * wv[j];
				for (int j = 0; j < n; j++) g2[j] += w->A[i * n + j] * (r[i] - ax);
			}
			small_ldl_solve(n, MS, DS, g2);
			for (int j = 0; j < n; j++) wv[j] += g2[j];
		}
		/* residuals of the working rows at the least-squares point; rows met with slack leave
		 * (removal only, so the iteration is monotone and ends) */
		int signs_ok = 1;
		double vmax = 0;
		for (int i = 0; i < m; i++) {
			v[i] = 0.0;
			if (!act[i]) continue;
			double ax = 0;
			for (int j = 0; j < n; j++) ax += w->A[i * n + j] * wv[j];
			v[i] = r[i] - ax;
			if ((act[i] == -1 && v[i] < -1e-12) || (act[i] == 1 && v[i] > 1e-12)) {
				signs_ok = 0;
				act[i] = 0;
				continue;
			}
			if (fabs(v[i]) > vmax) vmax = fabs(v[i]);
		}
		if (signs_ok) {
			if (!(vmax > 1e-7)) return 0;
			for (int j = 0; j < n; j++) {
				double g = 0;
				for (int i = 0; i < m; i++) g += w->A[i * n + j] * v[i];
				if (fabs(g) > 1e-9 * vmax) return 0;
			}
			return 1;
		}
		if (iter >= g_farkas_iters) return 0;
	}
}

static int g_as_rounds = 12; /* active-set corrections per attempt (device default) */
void or_set_as_rounds(int k) { g_as_rounds = k; }
void or_set_as_refine(int k) { g_as_refine = k; }

static int try_polish(const ws_t *w, const double *z, const double *y, double *xpol, double ktol)
{
	const int n = w->n, m = w->m;
	int act[MAXM], viol[MAXM], wrong[MAXM];
	double r[MAXM], nu[MAXM], x[MAXN], vmag[MAXM];
	for (int i = 0; i < m; i++) {
		act[i] = 0;
		r[i] = 0;
		if (w->u[i] - w->l[i] < RHO_TOL) { act[i] = 2; r[i] = w->l[i]; } /* equality: always active, free sign */
		else if (z[i] - w->l[i] < -y[i]) { act[i] = -1; r[i] = w->l[i]; }
		else if (w->u[i] - z[i] < y[i]) { act[i] = 1; r[i] = w->u[i]; }
	}
	for (int round = 0;; round++) {
		if (as_solve(w, act, r, x, nu, viol, wrong, vmag, ktol)) {
			for (int j = 0; j < n; j++) xpol[j] = x[j];
			return 1;
		}
		int viol_active = 0;
		for (int i = 0; i < m; i++)
			if (viol[i] && act[i]) viol_active = 1;
		if (viol_active && as_farkas(w, act, r)) return 2;
		if (round >= g_as_rounds) return 0;
		int changed = 0;
		for (int i = 0; i < m; i++) {
			if (act[i] == 2) continue;
			if (wrong[i]) { act[i] = 0; changed = 1; }
		}
		/* of the rows found violated only the most violated one(s) join: the working set then stays
		 * consistent on feasible problems and an inconsistency points at a genuine conflict */
		double worst = 0;
		for (int i = 0; i < m; i++)
			if (viol[i] && !act[i] && vmag[i] > worst) worst = vmag[i];
		for (int i = 0; i < m; i++)
			if (viol[i] && !act[i] && worst > 0 && vmag[i] == worst) {
				act[i] = viol[i];
				r[i] = viol[i] < 0 ? w->l[i] : w->u[i];
				changed = 1;
			}
		if (!changed) return 0;
	}
}

int or_qp_admm(const or_qp *qp, const or_admm_settings *s, double *xout, or_admm_info *info)
{
	static __thread ws_t W;
	ws_t *w = &W;
	const int n = qp->nv, m = qp->nc + qp->nv;
	if (n > MAXN || m > MAXM) return -100;
	w->n = n;
	w->m = m;
	/* ---- form translation, src/qpwrapper_osqp.cpp: P=2H (:271), q=c (:311-317),
	 * A=[A;I] (:319-343), l=[b;lb] (:346-368), u=[OSQP_INFTY or b where be; ub] (:81-84,364-376) */
	for (int j = 0; j < n; j++) {
		w->P[j] = 2.0 * qp->Hd[j];
		w->q[j] = qp->c[j];
	}
	for (int i = 0; i < qp->nc; i++) {
		for (int j = 0; j < n; j++) w->A[i * n + j] = qp->A[i + j * qp->nc];
		w->l[i] = qp->b[i];
		w->u[i] = (qp->be && qp->be[i]) ? qp->b[i] : OSQP_INFTY;
	}
	for (int j = 0; j < n; j++) {
		const int i = qp->nc + j;
		for (int k = 0; k < n; k++) w->A[i * n + k] = (k == j) ? 1.0 : 0.0;
		w->l[i] = qp->lb[j];
		w->u[i] = qp->ub[j];
	}
	if (s->scaling > 0) scale_data(w, s->scaling, s->scaling_pow2);
	else {
		for (int j = 0; j < n; j++) w->D[j] = w->Dinv[j] = 1.0;
		for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1.0;
		w->c = w->cinv = 1.0;
	}
	w->rho = s->rho;
	set_rho_vec(w);
	if (factor(w, s->sigma, s->reduced_kkt)) return -100;

	double x[MAXN] = {0}, z[MAXM] = {0}, y[MAXM] = {0};
	double xp[MAXN], zp[MAXM], xt[MAXN], zt[MAXM], dx[MAXN], dy[MAXM];
	double rhs[MAXN + MAXM], Ax[MAXM], Px[MAXN], Aty[MAXN], tmpn[MAXN], tmpm[MAXM];
	int status = 0, iter = 0, rho_updates = 0;
	double polish_delta = 1e-9;
	double pri_res = 0, dua_res = 0;
	const int ct = s->check_termination;

	/* polish == 2 (device default): one attempt of the active-set finish from the empty working set before the
	 * first iteration; z0 = the point of [l,u] nearest to 0 with y0 = 0 marks nothing but the equalities active */
	if (s->polish == 2) {
		double z0[MAXM], y0[MAXM], xpol[MAXN];
		for (int i = 0; i < m; i++) {
			z0[i] = w->l[i] > 0 ? w->l[i] : (w->u[i] < 0 ? w->u[i] : 0.0);
			y0[i] = 0.0;
		}
		g_as_delta = polish_delta;
		const int pr = try_polish(w, z0, y0, xpol, 1e-9);
		if (pr == 1) {
			memcpy(x, xpol, sizeof(double) * n);
			status = OR_OSQP_SOLVED;
			rho_updates += 1000;
		} else if (pr == 2) {
			status = OR_OSQP_PRIMAL_INFEASIBLE;
			rho_updates += 2000;
		}
	}
	const int decided_before = status != 0;
	for (iter = decided_before ? s->max_iter + 1 : 1; iter <= s->max_iter; iter++) {
		memcpy(xp, x, sizeof(double) * n);
		memcpy(zp, z, sizeof(double) * m);
		/* x~, z~ */
		if (s->reduced_kkt) {
			for (int j = 0; j < n; j++) rhs[j] = s->sigma * xp[j] - w->q[j];
			for (int i = 0; i < m; i++) {
				const double t = w->rho_vec[i] * zp[i] - y[i];
				for (int j = 0; j < n; j++) rhs[j] += w->A[i * n + j] * t;
			}
			ldl_solve(w, rhs);
			for (int j = 0; j < n; j++) xt[j] = rhs[j];
			for (int i = 0; i < m; i++) {
				double a = 0;
				for (int j = 0; j < n; j++) a += w->A[i * n + j] * xt[j];
				zt[i] = a;
			}
		} else {
			for (int j = 0; j < n; j++) rhs[j] = s->sigma * xp[j] - w->q[j];
			for (int i = 0; i < m; i++) rhs[n + i] = zp[i] - w->rho_inv[i] * y[i];
			ldl_solve(w, rhs);
			for (int j = 0; j < n; j++) xt[j] = rhs[j];
			for (int i = 0; i < m; i++) zt[i] = zp[i] + w->rho_inv[i] * (rhs[n + i] - y[i]);
		}
		/* x, z, y */
		for (int j = 0; j < n; j++) {
			x[j] = s->alpha * xt[j] + (1.0 - s->alpha) * xp[j];
			dx[j] = x[j] - xp[j];
		}
		for (int i = 0; i < m; i++) {
			double v = s->alpha * zt[i] + (1.0 - s->alpha) * zp[i] + w->rho_inv[i] * y[i];
			if (v < w->l[i]) v = w->l[i];
			if (v > w->u[i]) v = w->u[i];
			z[i] = v;
		}
		for (int i = 0; i < m; i++) {
			dy[i] = w->rho_vec[i] * (s->alpha * zt[i] + (1.0 - s->alpha) * zp[i] - z[i]);
			y[i] += dy[i];
		}
		const int do_check = (ct && iter % ct == 0);
		const int do_rho = (s->adaptive_rho && s->adaptive_rho_interval && iter % s->adaptive_rho_interval == 0);
		const int last = (iter == s->max_iter);
		if (!(do_check || do_rho || last)) continue;

		/* residuals (update_info / compute_pri_res / compute_dua_res) */
		for (int i = 0; i < m; i++) {
			double a = 0;
			for (int j = 0; j < n; j++) a += w->A[i * n + j] * x[j];
			Ax[i] = a;
			tmpm[i] = a - z[i];
		}
		for (int j = 0; j < n; j++) {
			Px[j] = w->P[j] * x[j];
			double a = 0;
			for (int i = 0; i < m; i++) a += w->A[i * n + j] * y[i];
			Aty[j] = a;
			tmpn[j] = Px[j] + w->q[j] + a;
		}
		pri_res = scaled_norm_inf(w->Einv, tmpm, m);
		dua_res = w->cinv * scaled_norm_inf(w->Dinv, tmpn, n);
		/* the method of multipliers inside contracts by about delta / (delta + s^2) per step (s: smallest singular
		 * value of the working rows): an undecided attempt is repeated at the next check with 1000x the penalty */
		if (s->polish && (do_check || last)) {
			double xpol[MAXN];
			g_as_delta = polish_delta;
			const int pr = try_polish(w, z, y, xpol, 1e-9);
			if (pr == 0 && polish_delta > 1e-14) polish_delta *= 1e-3;
			if (pr == 1) {
				memcpy(x, xpol, sizeof(double) * n);
				status = OR_OSQP_SOLVED;
				rho_updates += 1000; /* marks "ended by polish" in info */
				break;
			}
			if (pr == 2) {
				status = OR_OSQP_PRIMAL_INFEASIBLE;
				rho_updates += 2000; /* marks "ended by the Farkas attempt" */
				break;
			}
		}
		for (int approx = 0; approx <= (last ? 1 : 0); approx++) {
			if (!(do_check || last)) break;
			const double k = approx ? 10.0 : 1.0;
			const double ea = k * s->eps_abs, er = k * s->eps_rel;
			const double epi = k * s->eps_prim_inf, edi = k * s->eps_dual_inf;
			double np1 = scaled_norm_inf(w->Einv, z, m), np2 = scaled_norm_inf(w->Einv, Ax, m);
			const double eps_pri = ea + er * (np1 > np2 ? np1 : np2);
			double nd = scaled_norm_inf(w->Dinv, w->q, n);
			double nd2 = scaled_norm_inf(w->Dinv, Aty, n), nd3 = scaled_norm_inf(w->Dinv, Px, n);
			if (nd2 > nd) nd = nd2;
			if (nd3 > nd) nd = nd3;
			const double eps_dua = ea + er * w->cinv * nd;
			int prim_ok = pri_res < eps_pri, dual_ok = dua_res < eps_dua;
			int prim_inf = 0, dual_inf = 0;
			if (!prim_ok) { /* is_primal_infeasible */
				double pdy[MAXM];
				for (int i = 0; i < m; i++) {
					double v = dy[i];
					if (w->u[i] > OSQP_INFTY * MIN_SCALING) {
						if (w->l[i] < -OSQP_INFTY * MIN_SCALING) v = 0.0;
						else v = v < 0 ? v : 0.0;
					} else if (w->l[i] < -OSQP_INFTY * MIN_SCALING) v = v > 0 ? v : 0.0;
					pdy[i] = v;
				}
				const double ndy = scaled_norm_inf(w->E, pdy, m);
				if (ndy > epi) {
					double lhs = 0;
					for (int i = 0; i < m; i++)
						lhs += w->u[i] * (pdy[i] > 0 ? pdy[i] : 0.0) + w->l[i] * (pdy[i] < 0 ? pdy[i] : 0.0);
					if (lhs < -epi * ndy) {
						double t[MAXN];
						for (int j = 0; j < n; j++) {
							double a = 0;
							for (int i = 0; i < m; i++) a += w->A[i * n + j] * pdy[i];
							t[j] = w->Dinv[j] * a;
						}
						prim_inf = norm_inf(t, n) < epi * ndy;
					}
				}
			}
			if (!dual_ok) { /* is_dual_infeasible */
				const double ndx = scaled_norm_inf(w->D, dx, n);
				if (ndx > edi) {
					double qdx = 0;
					for (int j = 0; j < n; j++) qdx += w->q[j] * dx[j];
					if (qdx < -w->c * edi * ndx) {
						double t[MAXN];
						for (int j = 0; j < n; j++) t[j] = w->Dinv[j] * w->P[j] * dx[j];
						if (norm_inf(t, n) < w->c * edi * ndx) {
							dual_inf = 1;
							for (int i = 0; i < m && dual_inf; i++) {
								double a = 0;
								for (int j = 0; j < n; j++) a += w->A[i * n + j] * dx[j];
								a *= w->Einv[i];
								if ((w->u[i] < OSQP_INFTY * MIN_SCALING && a > edi * ndx) ||
								    (w->l[i] > -OSQP_INFTY * MIN_SCALING && a < -edi * ndx))
									dual_inf = 0;
							}
						}
					}
				}
			}
			if (prim_ok && dual_ok) status = approx ? OR_OSQP_SOLVED_INACCURATE : OR_OSQP_SOLVED;
			else if (prim_inf) status = approx ? OR_OSQP_PRIMAL_INFEASIBLE_INACCURATE : OR_OSQP_PRIMAL_INFEASIBLE;
			else if (dual_inf) status = approx ? OR_OSQP_DUAL_INFEASIBLE_INACCURATE : OR_OSQP_DUAL_INFEASIBLE;
			if (status) break;
		}
		if (status) break;
		if (do_rho) { /* adapt_rho / compute_rho_estimate, on scaled quantities */
			double pr = norm_inf(tmpm, m), dr = norm_inf(tmpn, n);
			double pn = norm_inf(z, m), pn2 = norm_inf(Ax, m);
			if (pn2 > pn) pn = pn2;
			double dn = norm_inf(w->q, n), dn2 = norm_inf(Aty, n), dn3 = norm_inf(Px, n);
			if (dn2 > dn) dn = dn2;
			if (dn3 > dn) dn = dn3;
			pr /= (pn + 1e-10);
			dr /= (dn + 1e-10);
			double rn = w->rho * sqrt(pr / (dr + 1e-10));
			if (rn < RHO_MIN) rn = RHO_MIN;
			if (rn > RHO_MAX) rn = RHO_MAX;
			if (rn > w->rho * s->adaptive_rho_tolerance || rn < w->rho / s->adaptive_rho_tolerance) {
				w->rho = rn;
				set_rho_vec(w);
				if (factor(w, s->sigma, s->reduced_kkt)) return -100;
				rho_updates++;
			}
		}
	}
	if (decided_before) iter = 0;
	if (!status) {
		status = OR_OSQP_MAX_ITER_REACHED;
		iter = s->max_iter;
	}
	for (int j = 0; j < n; j++) xout[j] = w->D[j] * x[j];
	if (info) {
		info->status = status;
		info->iters = iter;
		info->rho_updates = rho_updates;
		info->pri_res = pri_res;
		info->dua_res = dua_res;
	}
	/* QPWrapperOsqp::solve, src/qpwrapper_osqp.cpp:225-238 */
	if (status == OR_OSQP_SOLVED || status == OR_OSQP_SOLVED_INACCURATE) return 1;
	return status;
}
