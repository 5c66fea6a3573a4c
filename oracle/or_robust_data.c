/*
 * or_robust_data.c -- CPU restatement of ASIFrobust (src/asif_robust.cpp) on the model and data the
 * reference ships and builds by default: examples/DoubleIntegrator_Robust.cpp (interval mass, gain and
 * friction; safety set = the half-planes of include/KernelData_70-135kg.h; npSSmax = 5 of npSS = 100).
 * TEST INFRASTRUCTURE (see or_oracle.h).  nx == 2, nu == 1.
 *
 * Differences from the C5 path in or_assembly.c (assemble_robust): the half-planes are a data array of any
 * length, and npSSmax < npSS, so the npSSmax smallest h are selected first (src/asif_robust.cpp:296-315;
 * std::sort leaves the order of equal keys unspecified, lowest index first here).
 *
 * Pinning: the interval Lie derivatives are PINNED against the reference's libaffa
 * (tests/golden/affa_di_robust_lie.json, oracle/gen_robust_golden.py through oracle/_ref).  Row selection,
 * row placement and everything at the OSQP boundary: PARITY UNPINNED (src/asif_robust.cpp needs <osqp.h>).
 * u*, delta* := exact optimum of the assembled QP through the multiplier elimination of or_filter.c.
 */
#include "or_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct or_rb {
	or_rb_desc d;
	double *hp;
	or_af mInt, KInt, FInt;
	or_af_ctx cx;
	int nv, nc;
};

/* examples/DoubleIntegrator_Robust.cpp:20-37,85-90 */
void or_rb_default(or_rb_desc *d)
{
	memset(d, 0, sizeof(*d));
	d->npSSmax = 5;
	d->relaxCost = 50.0;
	d->relaxLb = 5.0;
	d->inf = 1e20;
	d->lb[0] = -20;
	d->ub[0] = 20;
	d->mMin = 70.;
	d->mMax = 135.;
	d->Klo = 5.7 - 0.1;
	d->Khi = 5.7 + 0.1;
	d->Flo = 23 - 2; /* FInt = interval(F-DF, F-DF), :37 */
	d->Fhi = 23 - 2;
}

or_rb *or_rb_create(const or_rb_desc *d)
{
	if (d->N < 1 || !d->halfPlanes) return 0;
	or_rb *z = (or_rb *)calloc(1, sizeof(or_rb));
	z->d = *d;
	z->hp = (double *)malloc(sizeof(double) * 2 * d->N);
	memcpy(z->hp, d->halfPlanes, sizeof(double) * 2 * d->N);
	z->d.halfPlanes = z->hp;
	if (z->d.npSSmax > d->N || z->d.npSSmax < 0) z->d.npSSmax = d->N; /* src/asif_robust.cpp:20 */
	z->nv = 1 + 1 + z->d.npSSmax * 2 * 2;                                /* :21 */
	z->nc = z->d.npSSmax * 3;                                            /* :22 */
	/* globals of the example, constructed before main(): symbols of m, K, F come first (:28-37) */
	or_af_interval(&z->cx, &z->mInt, d->mMin, d->mMax);
	or_af_interval(&z->cx, &z->KInt, d->Klo, d->Khi);
	or_af_interval(&z->cx, &z->FInt, d->Flo, d->Fhi);
	return z;
}

void or_rb_destroy(or_rb *z)
{
	if (!z) return;
	free(z->hp);
	free(z);
}

void or_rb_dims(const or_rb *z, int *nv, int *nc, int *npSSmax)
{
	*nv = z->nv;
	*nc = z->nc;
	*npSSmax = z->d.npSSmax;
}

/* updateConstraints, src/asif_robust.cpp:275-367.  sel (optional): the npSSmax selected half-planes. */
int or_rb_assemble(const or_rb *z, const double *x, double *A, double *b, int32_t *sel)
{
	const int nx = 2, nu = 1, N = z->d.N, M = z->d.npSSmax, nv = z->nv, nc = z->nc;
	/* fixed structure of initialize(), :103-133 */
	for (int i = 0; i < nc * nv; i++) A[i] = 0.0;
	for (int i = 0; i < nc; i++) b[i] = 0.0;
	int iCol = nu + 1;
	for (int iRow = 0; iRow < nc; iRow += nu + 2) {
		A[(iRow + 1) + 0 * nc] = -1.0;
		for (int i = 0; i < nu + 1; i++) {
			A[(iRow + 1 + i) + (iCol + i) * nc] = 1.0;
			A[(iRow + 1 + i) + (iCol + nu + 1 + i) * nc] = -1.0;
		}
		b[iRow + nu + 1] = 1.0;
		iCol += 2 * (nu + 1);
	}
	or_af_ctx cx = z->cx;
	or_af xI[2], f[2], g[2], t, u;
	for (int i = 0; i < nx; i++) or_af_interval(&cx, &xI[i], x[i], x[i]); /* :282-284 */
	double *hFull = (double *)malloc(sizeof(double) * N);
	for (int i = 0; i < N; i++) /* examples/DoubleIntegrator_Robust.cpp:41-49 */
		hFull[i] = 1. - z->hp[2 * i] * x[0] - z->hp[2 * i + 1] * x[1];
	/* dynamics, examples/DoubleIntegrator_Robust.cpp:51-58 */
	f[0] = xI[1];
	or_af_neg(&z->FInt, &t);
	or_af_mul(&cx, &t, &xI[1], &u);
	or_af_div(&cx, &u, &z->mInt, &f[1]);
	or_af_const(&g[0], 0.);
	or_af_div(&cx, &z->KInt, &z->mInt, &g[1]);
	/* the npSSmax smallest h, :296-315 */
	int pick[64];
	if (M > 64) {
		free(hFull);
		return -100;
	}
	if (M < N) {
		for (int k = 0; k < M; k++) {
			int best = -1;
			for (int i = 0; i < N; i++) {
				int used = 0;
				for (int q = 0; q < k; q++) used |= pick[q] == i;
				if (!used && (best < 0 || hFull[i] < hFull[best])) best = i;
			}
			pick[k] = best;
		}
	} else
		for (int k = 0; k < M; k++) pick[k] = k;
	or_af DhI[64 * 2], Lfh, Lgh;
	for (int e = 0; e < M * nx; e++) { /* :323-325, column-major order of creation */
		const int i = e % M, j = e / M;
		const double v = -z->hp[2 * pick[i] + j];
		or_af_interval(&cx, &DhI[e], v, v);
	}
	iCol = nu + 1;
	for (int s = 0; s < M; s++) {
		/* include/asif_utils.h:46-62 / :22-44 instantiated on AAF */
		or_af_const(&Lfh, 0.0);
		for (int k = 0; k < nx; k++) {
			or_af_mul(&cx, &DhI[s + k * M], &f[k], &t);
			or_af_add(&Lfh, &t, &Lfh);
		}
		or_af_const(&Lgh, 0.0);
		for (int k = 0; k < nx; k++) {
			or_af_mul(&cx, &DhI[s + k * M], &g[k], &t);
			or_af_add(&Lgh, &t, &Lgh);
		}
		const int iRow = s * (nu + 2);
		double lo, hi;
		A[iRow + nu * nc] = hFull[pick[s]];
		or_af_convert(&Lgh, &lo, &hi);
		A[iRow + (iCol + 0) * nc] = lo;
		A[iRow + (iCol + (nu + 1) + 0) * nc] = -hi;
		or_af_convert(&Lfh, &lo, &hi);
		A[iRow + (iCol + nu) * nc] = lo;
		A[iRow + (iCol + (nu + 1) + nu) * nc] = -hi;
		iCol += 2 * (nu + 1);
		if (sel) sel[s] = pick[s];
	}
	free(hFull);
	return cx.overflow ? -100 : 1;
}

/* note on symbol order: the reference computes all Lfh first, then all Lgh (two matmul calls); products
 * create symbols, but a symbol created for row s never appears in another row's form, so evaluating row by
 * row as above yields the same centres, coefficients and radii. */

/* initialize()/updateCost(), src/asif_robust.cpp:84-101,140-150,369-380 */
void or_rb_qp_static(const or_rb *z, const double *uDes, double *Hd, double *c, double *lb, double *ub, uint8_t *be)
{
	const int nv = z->nv, nc = z->nc;
	for (int i = 0; i < nv; i++) {
		Hd[i] = 0.0;
		c[i] = 0.0;
		lb[i] = 0.0;
		ub[i] = or_no_bound(z->d.inf);
	}
	Hd[0] = 1.0;
	Hd[1] = z->d.relaxCost;
	c[0] = -2.0 * uDes[0];
	c[1] = -2.0 * z->d.relaxCost * z->d.relaxLb;
	lb[0] = z->d.lb[0];
	ub[0] = z->d.ub[0];
	lb[1] = z->d.relaxLb;
	for (int i = 0; i < nc; i++) be[i] = (i % 3) != 0;
}

static int rb_exact(const or_rb *z, const double *A, const double *uDes, double *sol)
{
	const int nc = z->nc, M = z->d.npSSmax, nr = 2 * M;
	double A2[2 * 2 * 64], b2[2 * 64];
	for (int s = 0; s < M; s++) {
		const int iRow = 3 * s, iCol = 2 + 4 * s;
		const double h = A[iRow + 1 * nc], lo_g = A[iRow + iCol * nc], hi_g = -A[iRow + (iCol + 2) * nc];
		const double lo_f = A[iRow + (iCol + 1) * nc];
		A2[2 * s] = lo_g;
		A2[2 * s + nr] = h;
		b2[2 * s] = -lo_f;
		A2[2 * s + 1] = hi_g;
		A2[2 * s + 1 + nr] = h;
		b2[2 * s + 1] = -lo_f;
	}
	const double Hd[2] = {1.0, z->d.relaxCost}, c[2] = {-2.0 * uDes[0], -2.0 * z->d.relaxCost * z->d.relaxLb};
	const double lb[2] = {z->d.lb[0], z->d.relaxLb}, ub[2] = {z->d.ub[0], or_no_bound(z->d.inf)};
	or_qp q = {2, nr, Hd, c, A2, b2, lb, ub, 0};
	double x2[2];
	const int r = or_qp_exact_small(&q, x2);
	if (r != 1) return r;
	for (int i = 0; i < z->nv; i++) sol[i] = NAN;
	sol[0] = x2[0];
	sol[1] = x2[1];
	return 1;
}

/* filter(x, uDes, uAct, relax), src/asif_robust.cpp:218-252 */
int or_rb_filter(const or_rb *z, int solver, const or_admm_settings *s, const double *x, const double *uDes,
                 double *uAct, double *relax)
{
	const int nv = z->nv, nc = z->nc;
	double *A = (double *)malloc(sizeof(double) * nc * nv), *b = (double *)malloc(sizeof(double) * nc);
	double *w = (double *)malloc(sizeof(double) * 5 * nv);
	uint8_t *be = (uint8_t *)malloc(nc);
	double *Hd = w, *c = w + nv, *lb = w + 2 * nv, *ub = w + 3 * nv, *sol = w + 4 * nv;
	int rc = -100;
	if (or_rb_assemble(z, x, A, b, 0) == 1) {
		or_rb_qp_static(z, uDes, Hd, c, lb, ub, be);
		int rt;
		if (solver == OR_SOLVER_ADMM) {
			or_admm_settings def;
			if (!s) {
				or_admm_default_settings(&def);
				s = &def;
			}
			or_qp q = {nv, nc, Hd, c, A, b, lb, ub, be};
			rt = or_qp_admm(&q, s, sol, 0);
		} else rt = rb_exact(z, A, uDes, sol);
		if (rt == 1) {
			double u = sol[0];
			if (u > z->d.ub[0]) u = z->d.ub[0];
			else if (u < z->d.lb[0]) u = z->d.lb[0];
			uAct[0] = u;
			relax[0] = sol[1];
			rc = 1;
		} else rc = -1; /* :250-251 */
	}
	free(A);
	free(b);
	free(w);
	free(be);
	return rc;
}

int64_t or_rb_filter_batch(const or_rb *z, int solver, const or_admm_settings *s, int64_t B, const double *x,
                           const double *uDes, double *uAct, double *relax, int32_t *rc)
{
	for (int64_t i = 0; i < B; i++) rc[i] = or_rb_filter(z, solver, s, x + 2 * i, uDes + i, uAct + i, relax + i);
	return B;
}

int64_t or_rb_assemble_batch(const or_rb *z, int64_t B, const double *x, double *A, double *b, int32_t *code,
                             int32_t *sel)
{
	for (int64_t i = 0; i < B; i++)
		code[i] = or_rb_assemble(z, x + 2 * i, A + i * z->nc * z->nv, b + i * z->nc, sel ? sel + i * z->d.npSSmax : 0);
	return B;
}
