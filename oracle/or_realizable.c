/*
 * or_realizable.c -- CPU restatement of ASIFrealizable (src/asif_realizable.cpp) on the sampled
 * double integrator of examples/DoubleIntegrator_RealizableSampled.cpp.  TEST INFRASTRUCTURE (see
 * or_oracle.h).  nx == 2, nu == 1 (the only shapes the reference ships kernel data and a driver for).
 *
 * Pinning: the interval Lie derivatives over facets are PINNED against the reference's libaffa
 * (tests/golden/affa_rz_facet_lie.json, made by oracle/gen_golden.py through oracle/_ref).  The rest of
 * the assembly and everything at the OSQP boundary is PARITY UNPINNED: src/asif_realizable.cpp needs
 * <osqp.h> and cannot be built here, and the reference has no tests of its own.
 *
 * Resolutions of what the reference leaves to OSQP's 1e-3 tolerances:
 *  - facetSolver_ (:381-441) is a pure feasibility question ("is some point of the facet inside the
 *    uncertainty box around x"); it is answered exactly (segment/box intersection in long double).
 *  - filter(): u*, delta* := exact optimum of the QP the reference assembles.  With nu == 1 the
 *    multipliers of every row group can be eliminated exactly (same argument as ASIFrobust, see
 *    or_filter.c): group s is satisfiable for a given u iff  lo(Lgh_s) u + lo(Lfh_s) >= 0  and
 *    hi(Lgh_s) u + lo(Lfh_s) >= 0.  relax[0] = solutionFull[nu] (:346) is the multiplier l+_0 of group 0,
 *    which the QP does not determine uniquely (H is zero on it); the oracle reports its smallest
 *    feasible value max(u*, 0).
 */
#include "or_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct or_rz {
	or_rz_desc d;
	double *vertices, *normals, *bbox, *table;
	int32_t *fverts, *factive;
	or_af *xFace; /* [nFacets][nx], built at initialize() (:137-157) */
	or_af mInt, KInt, FInt;
	or_af_ctx cx;
	int npSS, nv, nc;
};

/* examples/DoubleIntegrator_RealizableSampled.cpp:19-43,88-92 and include/asif_realizable.h:14-20 */
void or_rz_default(or_rz_desc *d)
{
	memset(d, 0, sizeof(*d));
	d->nx = 2;
	d->nu = 1;
	d->npSSmax = 2;
	d->uncertaintyBounds[0] = 0.031;
	d->uncertaintyBounds[1] = 0.028;
	d->relaxDes = 10.0;
	d->relaxOffset = 0.0;
	d->relaxCost = 100.0;
	d->inf = 1e20;
	d->lb[0] = -20.;
	d->ub[0] = 20.;
	d->mMin = 70.;
	d->mMax = 75.;
	d->Klo = 5.7 - 0.1;
	d->Khi = 5.7 + 0.1;
	d->Flo = 23 - 2; /* FInt = interval(F-DF, F-DF), :43 -- both ends F-DF */
	d->Fhi = 23 - 2;
}

/* examples/DoubleIntegrator_RealizableSampled.cpp:47-54 */
static void dis_dynamics_af(const or_rz *z, or_af_ctx *cx, const or_af *x, or_af *f, or_af *g)
{
	or_af t, u;
	f[0] = x[1];
	or_af_neg(&z->FInt, &t);
	or_af_mul(cx, &t, &x[1], &u);
	or_af_div(cx, &u, &z->mInt, &f[1]);
	or_af_const(&g[0], 0.);
	or_af_div(cx, &z->KInt, &z->mInt, &g[1]);
}

/* one (critical facet, active constraint) pair of :465-500; returns lo/hi of Lgh and Lfh */
static void facet_lie(or_rz *z, int iFacet, int jActive, double *out4)
{
	const int nx = 2;
	const int src = z->factive[iFacet * z->d.maxActiveConstraints + jActive];
	or_af DhInt[2], f[2], g[2], Lfh, Lgh, t;
	for (int k = 0; k < nx; k++) {
		const double Dh = -z->normals[src * nx + k]; /* :474-477 */
		or_af_interval(&z->cx, &DhInt[k], Dh, Dh);   /* :483-487 */
	}
	dis_dynamics_af(z, &z->cx, &z->xFace[iFacet * nx], f, g);
	or_af_const(&Lfh, 0.);
	for (int k = 0; k < nx; k++) { /* :497-499 */
		or_af_mul(&z->cx, &f[k], &DhInt[k], &t);
		or_af_add(&Lfh, &t, &Lfh);
	}
	or_af_const(&Lgh, 0.);
	for (int k = 0; k < nx; k++) { /* :501-506 */
		or_af_mul(&z->cx, &g[k], &DhInt[k], &t);
		or_af_add(&Lgh, &t, &Lgh);
	}
	or_af_convert(&Lgh, &out4[0], &out4[1]);
	or_af_convert(&Lfh, &out4[2], &out4[3]);
}

or_rz *or_rz_create(const or_rz_desc *d)
{
	if (d->nx != 2 || d->nu != 1 || d->nFacets < 1 || d->nVertices < 2) return 0;
	or_rz *z = (or_rz *)calloc(1, sizeof(or_rz));
	z->d = *d;
	const int nx = 2, nF = d->nFacets, nA = d->maxActiveConstraints;
	z->vertices = (double *)malloc(sizeof(double) * d->nVertices * nx);
	z->normals = (double *)malloc(sizeof(double) * nF * nx);
	z->fverts = (int32_t *)malloc(sizeof(int32_t) * nF * nx);
	z->factive = (int32_t *)malloc(sizeof(int32_t) * nF * nA);
	z->bbox = (double *)malloc(sizeof(double) * nF * nx * 2);
	z->table = (double *)malloc(sizeof(double) * nF * nA * 4);
	z->xFace = (or_af *)malloc(sizeof(or_af) * nF * nx);
	memcpy(z->vertices, d->vertices, sizeof(double) * d->nVertices * nx);
	memcpy(z->normals, d->facetNormals, sizeof(double) * nF * nx);
	memcpy(z->fverts, d->facetVertices, sizeof(int32_t) * nF * nx);
	memcpy(z->factive, d->facetActive, sizeof(int32_t) * nF * nA);
	z->d.vertices = z->vertices;
	z->d.facetNormals = z->normals;
	z->d.facetVertices = z->fverts;
	z->d.facetActive = z->factive;
	/* src/asif_realizable.cpp:19-22 */
	z->npSS = d->maxCriticalFacets * nA;
	if (z->d.npSSmax > nF) z->d.npSSmax = nF;
	z->nv = (z->d.npSSmax > 0) ? (1 + z->npSS * 2 * 2 + 1) : (1 + z->npSS * 2 * 2);
	z->nc = z->npSS * 3 + z->d.npSSmax;
	/* globals of the example are constructed before main(): symbols of m, K, F come first (:33-43) */
	z->cx.last = 0;
	z->cx.overflow = 0;
	or_af_interval(&z->cx, &z->mInt, d->mMin, d->mMax);
	or_af_interval(&z->cx, &z->KInt, d->Klo, d->Khi);
	or_af_interval(&z->cx, &z->FInt, d->Flo, d->Fhi);
	/* xFaceInt, :137-157: x = v0; for each further vertex  x = lam*x + (1-lam)*v  with lam = [0,1] */
	for (int i = 0; i < nF; i++) {
		or_af *xf = &z->xFace[i * nx];
		for (int k = 0; k < nx; k++) or_af_const(&xf[k], z->vertices[z->fverts[i * nx + 0] * nx + k]);
		for (int j = 1; j <= nx - 1; j++) {
			or_af lam, one, oml, a, b2;
			or_af_interval(&z->cx, &lam, 0., 1.);
			const double *vtx = &z->vertices[z->fverts[i * nx + j] * nx];
			for (int k = 0; k < nx; k++) {
				or_af_mul(&z->cx, &lam, &xf[k], &a);
				or_af_const(&one, 1.);
				or_af_sub(&one, &lam, &oml);
				or_af_scale(&oml, vtx[k], &b2);
				or_af_add(&a, &b2, &xf[k]);
			}
		}
		/* bounding box, :160-175 (the first nx vertices of the facet) */
		for (int k = 0; k < nx; k++) {
			double lo = HUGE_VAL, hi = -HUGE_VAL;
			for (int j = 0; j < nx; j++) {
				const double v = z->vertices[z->fverts[i * nx + j] * nx + k];
				if (v < lo) lo = v;
				if (v > hi) hi = v;
			}
			z->bbox[(i * nx + k) * 2 + 0] = lo;
			z->bbox[(i * nx + k) * 2 + 1] = hi;
		}
	}
	/* dynamics_(xFaceInt) and the Lie derivatives do not depend on x (:489-506): evaluate once */
	for (int i = 0; i < nF; i++)
		for (int j = 0; j < nA; j++) facet_lie(z, i, j, &z->table[(i * nA + j) * 4]);
	if (z->cx.overflow) {
		or_rz_destroy(z);
		return 0;
	}
	return z;
}

void or_rz_destroy(or_rz *z)
{
	if (!z) return;
	free(z->vertices);
	free(z->normals);
	free(z->fverts);
	free(z->factive);
	free(z->bbox);
	free(z->table);
	free(z->xFace);
	free(z);
}

void or_rz_dims(const or_rz *z, int *nv, int *nc, int *npSS, int *npSSmax)
{
	*nv = z->nv;
	*nc = z->nc;
	*npSS = z->npSS;
	*npSSmax = z->d.npSSmax;
}

void or_rz_table(const or_rz *z, double *table, double *bbox)
{
	if (table) memcpy(table, z->table, sizeof(double) * z->d.nFacets * z->d.maxActiveConstraints * 4);
	if (bbox) memcpy(bbox, z->bbox, sizeof(double) * z->d.nFacets * 2 * 2);
}

/* facetSolver_ (:381-384,411-427): exists lam in [0,1], v = lam v0 + (1-lam) v1 with |v - x| <= unc ? */
static int facet_touches_box(const or_rz *z, int iFacet, const double *x)
{
	const double *v0 = &z->vertices[z->fverts[iFacet * 2 + 0] * 2], *v1 = &z->vertices[z->fverts[iFacet * 2 + 1] * 2];
	long double tlo = 0.0L, thi = 1.0L;
	for (int k = 0; k < 2; k++) {
		const long double d = (long double)v0[k] - v1[k];
		const long double lo = (long double)x[k] - z->d.uncertaintyBounds[k] - v1[k];
		const long double hi = (long double)x[k] + z->d.uncertaintyBounds[k] - v1[k];
		if (d > 0) {
			if (lo / d > tlo) tlo = lo / d;
			if (hi / d < thi) thi = hi / d;
		} else if (d < 0) {
			if (hi / d > tlo) tlo = hi / d;
			if (lo / d < thi) thi = lo / d;
		} else if (lo > 0 || hi < 0) return 0;
	}
	return tlo <= thi;
}

/* updateConstraints, :375-610.  A nc x nv column-major, b[nc]; info (optional, 4 + maxCriticalFacets
 * + npSSmax ints): nCriticalFacets, then the critical facets, then the barrier facets. */
int or_rz_assemble(const or_rz *z, const double *x, double *A, double *b, int32_t *info)
{
	const int nx = 2, nu = 1, nF = z->d.nFacets, nA = z->d.maxActiveConstraints;
	const int npSS = z->npSS, npSSmax = z->d.npSSmax, nv = z->nv, nc = z->nc;
	/* fixed structure of initialize(), :207-246 */
	for (int i = 0; i < nc * nv; i++) A[i] = 0.0;
	for (int i = 0; i < nc; i++) b[i] = 0.0;
	int iCol = nu;
	for (int iRow = 0; iRow < npSS * (nu + 2); iRow += nu + 2) {
		A[(iRow + 1) + 0 * nc] = -1.0;
		for (int i = 0; i < nu + 1; i++) {
			A[(iRow + 1 + i) + (iCol + i) * nc] = 1.0;
			A[(iRow + 1 + i) + (iCol + nu + 1 + i) * nc] = -1.0;
		}
		b[iRow + nu + 1] = 1.0;
		iCol += 2 * (nu + 1);
	}
	double *hFull = (double *)malloc(sizeof(double) * nF);
	int anyNeg = 0;
	for (int i = 0; i < nF; i++) { /* :386-393 */
		double h = 1.;
		for (int j = 0; j < nx; j++) h -= z->normals[i * nx + j] * x[j];
		hFull[i] = h;
		if (h < 0.) anyNeg = 1;
	}
	int crit[64], nCrit = 0;
	for (int i = 0; i < nF && nCrit < z->d.maxCriticalFacets; i++) { /* :396-442 */
		int potential = 1;
		for (int j = 0; j < nx; j++)
			if (x[j] < z->bbox[(i * nx + j) * 2] - z->d.uncertaintyBounds[j] ||
			    x[j] > z->bbox[(i * nx + j) * 2 + 1] + z->d.uncertaintyBounds[j]) {
				potential = 0;
				break;
			}
		if (potential && facet_touches_box(z, i, x)) crit[nCrit++] = i;
	}
	/* :445-527: rows of the active constraints of every critical facet, remaining groups zero */
	iCol = nu;
	int s = 0;
	for (int c = 0; c < nCrit; c++)
		for (int j = 0; j < nA; j++, s++) {
			const double *t = &z->table[(crit[c] * nA + j) * 4];
			const int iRow = s * (nu + 2), col = nu + s * 2 * (nu + 1);
			A[iRow + (col + 0) * nc] = t[0];
			A[iRow + (col + (nu + 1) + 0) * nc] = -t[1];
			A[iRow + (col + nu) * nc] = t[2];
			A[iRow + (col + (nu + 1) + nu) * nc] = -t[3];
		}
	for (; s < npSS; s++) { /* interval(0.).convert() = [0,0]; -right() = -0.0 */
		const int iRow = s * (nu + 2), col = nu + s * 2 * (nu + 1);
		A[iRow + (col + (nu + 1) + 0) * nc] = -0.0;
		A[iRow + (col + (nu + 1) + nu) * nc] = -0.0;
	}
	int barrier[64];
	if (npSSmax > 0) { /* :530-600 */
		or_af_ctx cx = z->cx;
		or_af xInt[2], fI[2], gI[2];
		for (int i = 0; i < nx; i++) or_af_interval(&cx, &xInt[i], x[i], x[i]);
		dis_dynamics_af(z, &cx, xInt, fI, gI);
		double f[2], g[2], lo, hi;
		for (int i = 0; i < nx; i++) { /* :548-553, interval::mid() = lo*0.5 + hi*0.5 */
			or_af_convert(&fI[i], &lo, &hi);
			f[i] = lo * 0.5 + hi * 0.5;
			or_af_convert(&gI[i], &lo, &hi);
			g[i] = lo * 0.5 + hi * 0.5;
		}
		/* npSSmax smallest h (std::sort on h, :562-571; ties are unspecified there, lowest index here) */
		for (int k = 0; k < npSSmax; k++) {
			int best = -1;
			for (int i = 0; i < nF; i++) {
				int used = 0;
				for (int q = 0; q < k; q++) used |= barrier[q] == i;
				if (!used && (best < 0 || hFull[i] < hFull[best])) best = i;
			}
			barrier[k] = best;
		}
		if (npSSmax == nF)
			for (int k = 0; k < nF; k++) barrier[k] = k; /* :574-579: no sorting */
		for (int i = 0; i < npSSmax; i++) { /* :581-599 */
			const int fi = barrier[i];
			double Lfh = 0.0, Lgh = 0.0;
			for (int k = 0; k < nx; k++) Lfh += -z->normals[fi * nx + k] * f[k];
			for (int k = 0; k < nx; k++) Lgh += -z->normals[fi * nx + k] * g[k];
			A[(npSS * (nu + 2)) + i + 0 * nc] = Lgh;
			A[(npSS * (nu + 2)) + i + (nv - 1) * nc] = 1.0;
			b[(npSS * (nu + 2)) + i] = -Lfh - z->d.relaxDes * (hFull[fi] - z->d.relaxOffset);
		}
	}
	if (info) {
		info[0] = nCrit;
		for (int c = 0; c < z->d.maxCriticalFacets; c++) info[1 + c] = c < nCrit ? crit[c] : -1;
		for (int i = 0; i < npSSmax; i++) info[1 + z->d.maxCriticalFacets + i] = barrier[i];
	}
	free(hFull);
	return (nCrit == 0 && anyNeg) ? -1 : 1; /* :602-605 */
}

/* initialize()/updateCost(): H diag, c, bounds, equality flags (:177-204,243-246,686-697) */
void or_rz_qp_static(const or_rz *z, const double *uDes, double *Hd, double *c, double *lb, double *ub, uint8_t *be)
{
	const int nv = z->nv, nc = z->nc, npSS = z->npSS;
	for (int i = 0; i < nv; i++) {
		Hd[i] = 0.0;
		c[i] = 0.0;
		lb[i] = 0.0;
		ub[i] = or_no_bound(z->d.inf);
	}
	Hd[0] = 1.0;
	if (z->d.npSSmax > 0) Hd[nv - 1] = z->d.relaxCost;
	c[0] = -2.0 * uDes[0];
	lb[0] = z->d.lb[0];
	ub[0] = z->d.ub[0];
	for (int i = 0; i < nc; i++) be[i] = (i < npSS * 3) && (i % 3 != 0);
}

/* exact optimum through the multiplier elimination described at the top of the file */
static int rz_exact(const or_rz *z, const double *A, const double *b, const double *uDes, double *sol)
{
	const int nc = z->nc, nv = z->nv, npSS = z->npSS, npSSmax = z->d.npSSmax;
	const int nr = 2 * npSS + npSSmax;
	double *A2 = (double *)calloc((size_t)nr * 2, sizeof(double)), *b2 = (double *)calloc(nr, sizeof(double));
	for (int s = 0; s < npSS; s++) {
		const int iRow = 3 * s, col = 1 + 4 * s;
		const double lo_g = A[iRow + col * nc], hi_g = -A[iRow + (col + 2) * nc], lo_f = A[iRow + (col + 1) * nc];
		A2[2 * s] = lo_g;
		A2[2 * s + 1] = hi_g;
		b2[2 * s] = -lo_f;
		b2[2 * s + 1] = -lo_f;
	}
	for (int i = 0; i < npSSmax; i++) {
		A2[2 * npSS + i] = A[3 * npSS + i];
		A2[2 * npSS + i + nr] = 1.0;
		b2[2 * npSS + i] = b[3 * npSS + i];
	}
	int r;
	double x2[2];
	if (npSSmax > 0) {
		const double Hd[2] = {1.0, z->d.relaxCost}, c[2] = {-2.0 * uDes[0], 0.0};
		const double lb[2] = {z->d.lb[0], 0.0}, ub[2] = {z->d.ub[0], or_no_bound(z->d.inf)};
		or_qp q = {2, nr, Hd, c, A2, b2, lb, ub, 0};
		r = or_qp_exact_small(&q, x2);
	} else {
		const double Hd[1] = {1.0}, c[1] = {-2.0 * uDes[0]};
		or_qp q = {1, nr, Hd, c, A2, b2, z->d.lb, z->d.ub, 0};
		r = or_qp_exact_small(&q, x2);
		x2[1] = 0.0;
	}
	free(A2);
	free(b2);
	if (r != 1) return r;
	for (int i = 0; i < nv; i++) sol[i] = NAN;
	sol[0] = x2[0];
	sol[1] = x2[0] > 0 ? x2[0] : 0.0; /* smallest feasible l+_0 of group 0 */
	if (npSSmax > 0) sol[nv - 1] = x2[1];
	return 1;
}

/* filter(x, uDes, uAct, relax[2]), :284-352 */
int or_rz_filter(const or_rz *z, int solver, const or_admm_settings *s, const double *x, const double *uDes,
                 double *uAct, double *relax, double *sol_full)
{
	const int nv = z->nv, nc = z->nc;
	double *A = (double *)malloc(sizeof(double) * nc * nv), *b = (double *)malloc(sizeof(double) * nc);
	double *w = (double *)malloc(sizeof(double) * 5 * nv);
	uint8_t *be = (uint8_t *)malloc(nc);
	double *Hd = w, *c = w + nv, *lb = w + 2 * nv, *ub = w + 3 * nv, *sol = w + 4 * nv;
	int rc;
	if (or_rz_assemble(z, x, A, b, 0) < 0) rc = -2; /* :324-326 */
	else {
		or_rz_qp_static(z, uDes, Hd, c, lb, ub, be);
		int rt;
		if (solver == OR_SOLVER_ADMM) {
			or_admm_settings def;
			if (!s) {
				or_admm_default_settings(&def);
				s = &def;
			}
			or_qp q = {nv, nc, Hd, c, A, b, lb, ub, be};
			rt = or_qp_admm(&q, s, sol, 0);
		} else rt = rz_exact(z, A, b, uDes, sol);
		if (rt == 1) { /* :340-349 */
			double u = sol[0];
			if (u > z->d.ub[0]) u = z->d.ub[0];
			else if (u < z->d.lb[0]) u = z->d.lb[0];
			uAct[0] = u;
			relax[0] = sol[1];
			relax[1] = sol[nv - 1];
			if (sol_full) memcpy(sol_full, sol, sizeof(double) * nv);
			rc = 1;
		} else rc = -1;
	}
	free(A);
	free(b);
	free(w);
	free(be);
	return rc;
}

int64_t or_rz_filter_batch(const or_rz *z, int solver, const or_admm_settings *s, int64_t B, const double *x,
                           const double *uDes, double *uAct, double *relax, int32_t *rc)
{
	for (int64_t i = 0; i < B; i++) rc[i] = or_rz_filter(z, solver, s, x + 2 * i, uDes + i, uAct + i, relax + 2 * i, 0);
	return B;
}

int64_t or_rz_assemble_batch(const or_rz *z, int64_t B, const double *x, double *A, double *b, int32_t *code,
                             int32_t *info, int info_stride)
{
	for (int64_t i = 0; i < B; i++)
		code[i] = or_rz_assemble(z, x + 2 * i, A + i * z->nc * z->nv, b + i * z->nc, info ? info + i * info_stride : 0);
	return B;
}
