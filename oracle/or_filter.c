/*
 * or_filter.c -- filter() of the four variants on top of assembly + a QP solver, batch loops,
 * and the seeded synthetic workloads.  TEST INFRASTRUCTURE (see or_oracle.h).
 */
#include "or_internal.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAXNV 40
#define MAXNC 64

/* src/asif.cpp:343-352 (identical in every variant) */
static void saturate_hard(const or_options *o, int nu, double *u)
{
	for (int i = 0; i < nu; i++) {
		if (u[i] > o->ub[i]) u[i] = o->ub[i];
		else if (u[i] < o->lb[i]) u[i] = o->lb[i];
	}
}

/* The robust QP's multipliers can be eliminated exactly when nu == 1: for fixed (u, delta) the
 * best lambda of row group s gives  h_s*delta + min(lo_g*u, hi_g*u) + lo_f >= 0, i.e. two plain
 * rows.  (u*, delta*) of the 18-variable problem the reference assembles equals the optimum of
 * this 2-variable one; used only to obtain the EXACT reference optimum by enumeration. */
static int robust_exact(const or_dims *d, const double *A, const double *Hd, const double *c,
                        const double *lb, const double *ub, double *sol)
{
	const int N = d->npSS, nc = d->nc, nu = d->nu;
	double A2[2 * OR_MAX_NPSS * 2], b2[2 * OR_MAX_NPSS];
	const int nr = 2 * N;
	for (int s = 0; s < N; s++) {
		const int iRow = s * (nu + 2), iCol = nu + 1 + s * 2 * (nu + 1);
		const double h = A[iRow + nu * nc];
		const double lo_g = A[iRow + iCol * nc], hi_g = -A[iRow + (iCol + nu + 1) * nc];
		const double lo_f = A[iRow + (iCol + nu) * nc];
		A2[(2 * s) + 0 * nr] = lo_g;
		A2[(2 * s) + 1 * nr] = h;
		b2[2 * s] = -lo_f;
		A2[(2 * s + 1) + 0 * nr] = hi_g;
		A2[(2 * s + 1) + 1 * nr] = h;
		b2[2 * s + 1] = -lo_f;
	}
	or_qp q = {2, nr, Hd, c, A2, b2, lb, ub, 0};
	double x2[2];
	int r = or_qp_exact_small(&q, x2);
	if (r != 1) return r;
	for (int i = 0; i < d->nv; i++) sol[i] = NAN;
	sol[0] = x2[0];
	sol[1] = x2[1];
	return 1;
}

static int solve_qp(int variant, const or_dims *d, const or_qp *qp, int solver, const or_admm_settings *s, double *sol)
{
	if (solver == OR_SOLVER_ADMM) {
		or_admm_settings def;
		if (!s) {
			or_admm_default_settings(&def);
			s = &def;
		}
		return or_qp_admm(qp, s, sol, 0);
	}
	int r;
	if (variant == OR_VARIANT_ROBUST) r = robust_exact(d, qp->A, qp->Hd, qp->c, qp->lb, qp->ub, sol);
	else r = or_qp_exact_small(qp, sol);
	if (r == 1) return 1;
	if (r == 0) return OR_OSQP_PRIMAL_INFEASIBLE;
	return r;
}

int or_filter(int model, int variant, const or_options *o, int solver, const or_admm_settings *s,
              const double *x, const double *uDes, double *uAct, double *relax, double *sol_full)
{
	const or_model *m = or_model_get(model);
	or_dims d;
	if (!m || or_get_dims(model, variant, o, &d)) return -100;
	double A[MAXNC * MAXNV], b[MAXNC], Hd[MAXNV], c[MAXNV], lb[MAXNV], ub[MAXNV], sol[MAXNV], diag[8];
	uint8_t be[MAXNC];
	if (d.nv > MAXNV || d.nc > MAXNC) return -100;
	or_qp_static(model, variant, o, uDes, Hd, c, lb, ub, be);
	const int code = or_assemble(model, variant, o, x, A, b, diag);
	if (code == -100) return -100;
	or_qp qp = {d.nv, d.nc, Hd, c, A, b, lb, ub, be};
	const int nu = d.nu;
	if (variant == OR_VARIANT_IMPLICIT_TB && code == -3) {
		/* src/asif_implicit_tb.cpp:354-361 */
		double Du[OR_MAX_NU * OR_MAX_NX];
		m->controller(o, x, uAct, Du);
		saturate_hard(o, nu, uAct);
		return -3;
	}
	const int rt = solve_qp(variant, &d, &qp, solver, s, sol);
	if (sol_full && rt == 1) memcpy(sol_full, sol, sizeof(double) * d.nv);
	if (rt == 1) {
		for (int i = 0; i < nu; i++) uAct[i] = sol[i];
		saturate_hard(o, nu, uAct);
		for (int i = 0; i < d.nrelax; i++) relax[i] = sol[nu + i];
		if (variant == OR_VARIANT_IMPLICIT_TB && code == 2) return 2; /* :298-307 */
		return 1;
	}
	switch (variant) {
	case OR_VARIANT_EXPLICIT: /* src/asif.cpp:208-209: uAct, relax untouched */
	case OR_VARIANT_ROBUST:   /* src/asif_robust.cpp:250-251 */
		return -1;
	case OR_VARIANT_IMPLICIT_RB: /* src/asif_implicit_robust.cpp:427-433: identical */
	case OR_VARIANT_IMPLICIT: { /* src/asif_implicit.cpp:348-355 */
		double Du[OR_MAX_NU * OR_MAX_NX];
		m->controller(o, x, uAct, Du);
		saturate_hard(o, nu, uAct);
		return -1;
	}
	case OR_VARIANT_IMPLICIT_TB: {
		double Du[OR_MAX_NU * OR_MAX_NX];
		m->controller(o, x, uAct, Du);
		saturate_hard(o, nu, uAct);
		return code == 2 ? -1 : rt; /* :309-315 vs :345-352 (raw solver code leaks) */
	}
	}
	return -100;
}

/* ------------------------------------------------------------------- batch */
typedef struct {
	int model, variant, solver;
	const or_options *o;
	const or_admm_settings *s;
	int64_t lo, hi;
	const double *x, *uDes;
	double *uAct, *relax;
	int32_t *rc;
	or_dims d;
} job_t;

static void *job_run(void *p)
{
	job_t *j = (job_t *)p;
	const or_dims *d = &j->d;
	for (int64_t i = j->lo; i < j->hi; i++)
		j->rc[i] = or_filter(j->model, j->variant, j->o, j->solver, j->s, j->x + i * d->nx, j->uDes + i * d->nu,
		                     j->uAct + i * d->nu, j->relax + i * d->nrelax, 0);
	return 0;
}

int64_t or_filter_batch(int model, int variant, const or_options *o, int solver, const or_admm_settings *s,
                        int64_t B, const double *x, const double *uDes, double *uAct, double *relax,
                        int32_t *rc, int nthreads)
{
	or_dims d;
	if (or_get_dims(model, variant, o, &d)) return -1;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	job_t jobs[256];
	pthread_t th[256];
	for (int t = 0; t < nthreads; t++) {
		job_t j = {model, variant, solver, o, s, B * t / nthreads, B * (t + 1) / nthreads, x, uDes, uAct, relax, rc, d};
		jobs[t] = j;
	}
	if (nthreads == 1) {
		job_run(&jobs[0]);
		return B;
	}
	for (int t = 0; t < nthreads; t++) pthread_create(&th[t], 0, job_run, &jobs[t]);
	for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
	return B;
}

int64_t or_assemble_batch(int model, int variant, const or_options *o, int64_t B, const double *x,
                          double *A, double *b, int32_t *code, double *diag8)
{
	or_dims d;
	if (or_get_dims(model, variant, o, &d)) return -1;
	for (int64_t i = 0; i < B; i++) {
		double dg[8] = {0};
		code[i] = or_assemble(model, variant, o, x + i * d.nx, A + i * d.nc * d.nv, b + i * d.nc, dg);
		if (diag8) memcpy(diag8 + i * 8, dg, sizeof(dg));
	}
	return B;
}

int64_t or_qp_solve_batch(int nv, int nc, int solver, const or_admm_settings *s, int64_t B,
                          const double *Hd, const double *c, const double *A, const double *b,
                          const double *lb, const double *ub, const uint8_t *be,
                          double *sol, int32_t *status, int32_t *iters)
{
	or_admm_settings def;
	if (!s) {
		or_admm_default_settings(&def);
		s = &def;
	}
	for (int64_t i = 0; i < B; i++) {
		or_qp q = {nv, nc, Hd + i * nv, c + i * nv, A + i * nc * nv, b + i * nc, lb + i * nv, ub + i * nv, be};
		or_admm_info info = {0, 0, 0, 0, 0};
		int r;
		if (solver == OR_SOLVER_ADMM) r = or_qp_admm(&q, s, sol + i * nv, &info);
		else {
			r = or_qp_exact_small(&q, sol + i * nv);
			if (r == 0) r = OR_OSQP_PRIMAL_INFEASIBLE;
		}
		status[i] = r;
		if (iters) iters[i] = info.iters;
	}
	return B;
}

/* ----------------------------------------------------------------- workloads */
/* splitmix64 of (seed*2^32 + k), k = i*16 + j, r = (z >> 11) * 2^-53 */
double or_rng_uniform(uint64_t seed, uint64_t i, uint64_t j)
{
	uint64_t z = (seed << 32) + (i * 16 + j);
	z += 0x9e3779b97f4a7c15ULL;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	z = z ^ (z >> 31);
	return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

void or_make_batch(int cfg, int64_t B, int64_t first, double *x, double *uDes)
{
	for (int64_t k = 0; k < B; k++) {
		const uint64_t i = (uint64_t)(first + k);
		switch (cfg) {
		case 2: /* DI explicit, seed 1 */
			x[2 * k + 0] = -1.2 + 2.4 * or_rng_uniform(1, i, 0);
			x[2 * k + 1] = -1.2 + 2.4 * or_rng_uniform(1, i, 1);
			uDes[k] = -1.5 + 3.0 * or_rng_uniform(1, i, 2);
			break;
		case 3: /* pendulum implicit, seed 2 */
			x[2 * k + 0] = -1.5 + 3.0 * or_rng_uniform(2, i, 0);
			x[2 * k + 1] = -1.5 + 3.0 * or_rng_uniform(2, i, 1);
			uDes[k] = -1.5 + 3.0 * or_rng_uniform(2, i, 2);
			break;
		case 4: { /* segway TB, seed 3: x_j = 0.05*xBound_j*(2r-1) */
			const double xb[4] = {3.0, 3.0, M_PI / 6, M_PI};
			for (int j = 0; j < 4; j++) x[4 * k + j] = 0.05 * xb[j] * (2.0 * or_rng_uniform(3, i, j) - 1.0);
			uDes[k] = -5.0 + 10.0 * or_rng_uniform(3, i, 4);
			break;
		}
		case 9: /* double integrator implicit (examples/DoubleIntegrator_implicit.cpp), seed 9 */
			/* |x| <= 0.4: beyond that the 2 s backup trajectory rarely reaches the small backup set (Pv = 0.002)
			 * and nearly every QP is infeasible (rc -1 + backup controller); this mixes both outcomes about evenly */
			x[2 * k + 0] = -0.4 + 0.8 * or_rng_uniform(9, i, 0);
			x[2 * k + 1] = -0.4 + 0.8 * or_rng_uniform(9, i, 1);
			uDes[k] = -1.5 + 3.0 * or_rng_uniform(9, i, 2);
			break;
		case 8: /* pendulum TB (examples/InvertedPendulum_ImplicitTB.cpp), seed 8: around and inside the backup set */
			x[2 * k + 0] = -1.4 + 3.0 * or_rng_uniform(8, i, 0);
			x[2 * k + 1] = -1.4 + 2.8 * or_rng_uniform(8, i, 1);
			uDes[k] = -1.5 + 3.0 * or_rng_uniform(8, i, 2);
			break;
		case 10: /* pendulum under ASIFimplicitRB (no example in the reference; C3's distribution), seed 10 */
			x[2 * k + 0] = -1.5 + 3.0 * or_rng_uniform(10, i, 0);
			x[2 * k + 1] = -1.5 + 3.0 * or_rng_uniform(10, i, 1);
			uDes[k] = -1.5 + 3.0 * or_rng_uniform(10, i, 2);
			break;
		case 11: /* synthetic two-input model under class ASIF, seed 11: part of the batch starts outside the set */
			x[2 * k + 0] = -1.6 + 3.2 * or_rng_uniform(11, i, 0);
			x[2 * k + 1] = -1.6 + 3.2 * or_rng_uniform(11, i, 1);
			uDes[2 * k + 0] = -1.5 + 3.0 * or_rng_uniform(11, i, 2);
			uDes[2 * k + 1] = -1.5 + 3.0 * or_rng_uniform(11, i, 3);
			break;
		case 12: /* double integrator TB (examples/DoubleIntegrator_implicit_tb.cpp), seed 12 */
			/* the backup set is the disc of radius 0.01 and the closed loop's slow pole is -0.51/s: within the 2.1 s
			 * horizon only states within ~0.03 of the origin reach it; +-0.04 mixes inside / hit / never-hit */
			x[2 * k + 0] = -0.04 + 0.08 * or_rng_uniform(12, i, 0);
			x[2 * k + 1] = -0.04 + 0.08 * or_rng_uniform(12, i, 1);
			uDes[k] = -1.5 + 3.0 * or_rng_uniform(12, i, 2);
			break;
		case 5: /* robust pendulum, seed 4 */
			x[2 * k + 0] = -3.0 + 6.0 * or_rng_uniform(4, i, 0);
			x[2 * k + 1] = -3.0 + 6.0 * or_rng_uniform(4, i, 1);
			uDes[k] = -1.5 + 3.0 * or_rng_uniform(4, i, 2);
			break;
		}
	}
}
