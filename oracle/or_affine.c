/*
 * or_affine.c -- fixed-capacity affine forms.  TEST INFRASTRUCTURE (see or_oracle.h).
 *
 * Restates the subset of libaffa the robust filter uses, arithmetic for arithmetic
 * (round-to-nearest everywhere; the reference's directed rounding is commented out,
 * lib/libaffa/src/aa_interval.cpp:85-93).  Differences in representation only:
 * no heap, a per-evaluation symbol counter instead of the global AAF::last
 * (aa_aafcommon.cpp:32) -- only the relative order of symbols matters.
 * Pinned against the real libaffa via oracle/_ref (tests/test_oracle_affine.py).
 */
#include "or_oracle.h"
#include <math.h>
#include <string.h>

enum { AF_AFFINE = 1, AF_INFINITE = 2, AF_NAN = 4 };

/* aa_util.h:31-44 */
static int af_binary_special(int a, int b)
{
	if (a == AF_AFFINE && b == AF_AFFINE) return AF_AFFINE;
	if (a == AF_NAN || b == AF_NAN) return AF_NAN;
	if (a == (AF_NAN | AF_AFFINE) || b == (AF_NAN | AF_AFFINE)) return AF_NAN | AF_AFFINE;
	return a | b;
}

/* aa_aaf.h:149-153  AAF(double): no noise symbol */
void or_af_const(or_af *r, double v0)
{
	r->c = v0;
	r->n = 0;
	r->special = AF_AFFINE;
}

/* aa_aafcommon.cpp:81-100  AAF(interval): always allocates a fresh symbol, even for a point */
void or_af_interval(or_af_ctx *cx, or_af *r, double lo, double hi)
{
	unsigned en = ++cx->last;
	r->n = 1;
	r->idx[0] = en;
	if (hi - lo == HUGE_VAL) {
		r->c = 0;
		r->v[0] = HUGE_VAL;
		r->special = AF_INFINITE;
	} else {
		r->c = (hi + lo) / 2;
		r->v[0] = (hi - lo) / 2;
		r->special = AF_AFFINE;
	}
}

/* aa_aafcommon.cpp:232-245 */
double or_af_rad(const or_af *a)
{
	double sum = 0;
	for (int i = 0; i < a->n; i++) {
		if (a->v[i] >= 0.0) sum += a->v[i];
		else sum += -a->v[i];
	}
	return sum;
}

static int af_is_infinite(const or_af *a)
{
	if (a->special & AF_INFINITE) return 1;
	return or_af_rad(a) == HUGE_VAL;
}

static int af_is_indeterminate(const or_af *a)
{
	if (a->special & AF_INFINITE) return 1;
	if (a->special & AF_NAN) return 1;
	return or_af_rad(a) == HUGE_VAL;
}

/* aa_aafcommon.cpp:217-226 */
void or_af_convert(const or_af *a, double *lo, double *hi)
{
	if (af_is_indeterminate(a)) {
		*lo = -HUGE_VAL;
		*hi = HUGE_VAL;
		return;
	}
	double r = or_af_rad(a);
	*lo = a->c - r;
	*hi = a->c + r;
}

/* Sorted-union walk shared by + - * (aa_aafarithm.cpp:35-98,103-167; aa_aafapprox.cpp:34-101).
 * mode 0: a+b, 1: a-b, 2: a.c*vb + b.c*va */
static int af_merge(const or_af *a, const or_af *b, or_af *r, int mode)
{
	int ia = 0, ib = 0, k = 0;
	while (ia < a->n || ib < b->n) {
		if (k >= OR_AF_CAP) return -1;
		int takeA = 0, takeB = 0;
		if (ia == a->n) takeB = 1;
		else if (ib == b->n) takeA = 1;
		else if (a->idx[ia] < b->idx[ib]) takeA = 1;
		else if (b->idx[ib] < a->idx[ia]) takeB = 1;
		else takeA = takeB = 1;
		double va = takeA ? a->v[ia] : 0.0, vb = takeB ? b->v[ib] : 0.0;
		r->idx[k] = takeA ? a->idx[ia] : b->idx[ib];
		if (mode == 0) r->v[k] = (takeA && takeB) ? va + vb : (takeA ? va : vb);
		else if (mode == 1) r->v[k] = (takeA && takeB) ? va - vb : (takeA ? va : -vb);
		else r->v[k] = (takeA && takeB) ? a->c * vb + b->c * va : (takeA ? b->c * va : a->c * vb);
		ia += takeA;
		ib += takeB;
		k++;
	}
	return k;
}

void or_af_add(const or_af *a, const or_af *b, or_af *r)
{
	or_af t;
	t.c = a->c + b->c;
	t.special = af_binary_special(a->special, b->special);
	t.n = af_merge(a, b, &t, 0);
	if (t.n < 0) { t.n = 0; t.special = AF_NAN; }
	*r = t;
}

void or_af_sub(const or_af *a, const or_af *b, or_af *r)
{
	or_af t;
	t.c = a->c - b->c;
	t.special = af_binary_special(a->special, b->special);
	t.n = af_merge(a, b, &t, 1);
	if (t.n < 0) { t.n = 0; t.special = AF_NAN; }
	*r = t;
}

/* aa_aafarithm.cpp:172-183 */
void or_af_neg(const or_af *a, or_af *r)
{
	or_af t = *a;
	t.c = -t.c;
	for (int i = 0; i < t.n; i++) t.v[i] = -t.v[i];
	*r = t;
}

/* aa_aafarithm.cpp:189-200 */
void or_af_scale(const or_af *a, double k, or_af *r)
{
	or_af t = *a;
	t.c = k * a->c;
	for (int i = 0; i < t.n; i++) t.v[i] = k * t.v[i];
	*r = t;
}

/* aa_aafapprox.cpp:34-101: product; the quadratic term becomes ONE new independent
 * symbol with coefficient rad(a)*rad(b) (kept even when it is zero) */
void or_af_mul(or_af_ctx *cx, const or_af *a, const or_af *b, or_af *r)
{
	or_af t;
	t.c = a->c * b->c;
	int k = af_merge(a, b, &t, 2);
	if (k < 0 || k >= OR_AF_CAP) {
		cx->overflow = 1;
		t.n = 0;
		t.special = AF_NAN;
		*r = t;
		return;
	}
	t.idx[k] = ++cx->last;
	t.v[k] = or_af_rad(a) * or_af_rad(b);
	t.n = k + 1;
	t.special = af_binary_special(a->special, b->special);
	*r = t;
}

/* aa_aafarithm.cpp:233-261: z = alpha*P + dzeta, plus a new symbol delta */
static void af_affine_ctor(or_af_ctx *cx, const or_af *p, double alpha, double dzeta, double delta, int type, or_af *r)
{
	or_af t;
	if (p->n >= OR_AF_CAP) {
		cx->overflow = 1;
		t.n = 0;
		t.c = 0;
		t.special = AF_NAN;
		*r = t;
		return;
	}
	t.c = alpha * p->c + dzeta;
	t.n = p->n + 1;
	t.special = type;
	for (int i = 0; i < p->n; i++) {
		t.idx[i] = p->idx[i];
		t.v[i] = alpha * p->v[i];
	}
	t.idx[p->n] = ++cx->last;
	t.v[p->n] = delta;
	*r = t;
}

/* aa_aafapprox.cpp:155-179 (mini-range 1/x); interval::mid/radius aa_interval.cpp:79-118 */
void or_af_inv(or_af_ctx *cx, const or_af *p, or_af *r)
{
	if (p->special == AF_NAN) { or_af t; t.c = 0; t.n = 0; t.special = AF_NAN; *r = t; return; }
	if (p->special == AF_INFINITE) { or_af_interval(cx, r, -HUGE_VAL, HUGE_VAL); return; }
	double a, b, lo0;
	or_af_convert(p, &a, &b);
	lo0 = a;
	if (af_is_infinite(p) || ((a <= 0) && (b >= 0))) {
		or_af_interval(cx, r, -HUGE_VAL, HUGE_VAL);
		return;
	}
	const double t1 = fabs(a), t2 = fabs(b);
	a = t1 < t2 ? t1 : t2; /* std::min(t1,t2) */
	b = t1 < t2 ? t2 : t1; /* std::max(t1,t2) */
	const double alpha = -1 / (b * b);
	const double ilo = (1 / a) - alpha * a, ihi = 2 / b;
	const double mid = ilo * 0.5 + ihi * 0.5;
	const double r0 = mid - ilo, r1 = ihi - mid;
	const double radius = (r0 >= r1 ? r0 : r1);
	double dzeta = mid;
	if (lo0 < 0) dzeta = -dzeta;
	af_affine_ctor(cx, p, alpha, dzeta, radius, p->special, r);
}

/* aa_aafapprox.cpp:108-110 */
void or_af_div(or_af_ctx *cx, const or_af *a, const or_af *b, or_af *r)
{
	or_af ib;
	or_af_inv(cx, b, &ib);
	or_af_mul(cx, a, &ib, r);
}

/* aa_aaftrigo.cpp:42-135: 8-point least-squares line + max residual; width < 1e-10
 * collapses to a point interval (:67-71, a patch by the reference's author) */
void or_af_sin(or_af_ctx *cx, const or_af *p, or_af *r)
{
	enum { NPTS = 8 };
	const double PI = 4 * atan(1.0);
	if (af_is_infinite(p)) { or_af_interval(cx, r, -1, 1); return; }
	double a, b;
	or_af_convert(p, &a, &b);
	const double w = b - a;
	double alpha, dzeta, delta;
	if (w >= 2 * PI) {
		or_af_interval(cx, r, -1, 1);
		return;
	} else if (w < 1e-10) {
		const double tmp = sin(a * 0.5 + b * 0.5);
		or_af_interval(cx, r, tmp, tmp);
		return;
	} else {
		double x[NPTS], y[NPTS], res[NPTS];
		x[0] = a;
		y[0] = sin(a);
		x[NPTS - 1] = b;
		y[NPTS - 1] = sin(b);
		const double pas = w / (NPTS - 1);
		for (unsigned i = 1; i < NPTS - 1; i++) {
			x[i] = x[i - 1] + pas;
			y[i] = sin(x[i]);
		}
		double xm = 0, ym = 0;
		for (unsigned i = 0; i < NPTS; i++) {
			xm = xm + x[i];
			ym = ym + y[i];
		}
		xm = xm / NPTS;
		ym = ym / NPTS;
		double temp2 = 0;
		alpha = 0;
		for (unsigned i = 0; i < NPTS; i++) {
			const double temp1 = x[i] - xm;
			alpha += y[i] * temp1;
			temp2 += temp1 * temp1;
		}
		alpha = alpha / temp2;
		dzeta = ym - alpha * xm;
		for (unsigned i = 0; i < NPTS; i++) res[i] = fabs(y[i] - (dzeta + alpha * x[i]));
		delta = res[0];
		for (unsigned i = 1; i < NPTS; i++)
			if (delta < res[i]) delta = res[i]; /* std::max_element: first maximum */
	}
	af_affine_ctor(cx, p, alpha, dzeta, delta, p->special, r);
}

/* Register-program runner with the same instruction encoding as ref_affa_shim.cpp, so one
 * sequence can be replayed on this restatement and on the reference's libaffa. */
int or_af_run(const or_af_instr *prog, int nprog, int nreg, int cap, double *center, int *n, double *lo,
              double *hi, unsigned *idx, double *coef)
{
	enum { OP_CONST = 0, OP_INTERVAL, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_INV, OP_NEG, OP_SCALE, OP_SIN, OP_COPY };
	if (nreg > 64) return -1;
	static __thread or_af R[64];
	or_af_ctx cx = {0, 0};
	for (int r = 0; r < nreg; r++) or_af_const(&R[r], 0);
	for (int p = 0; p < nprog; p++) {
		const or_af_instr *I = &prog[p];
		switch (I->op) {
		case OP_CONST: or_af_const(&R[I->dst], I->imm0); break;
		case OP_INTERVAL: or_af_interval(&cx, &R[I->dst], I->imm0, I->imm1); break;
		case OP_ADD: or_af_add(&R[I->a], &R[I->b], &R[I->dst]); break;
		case OP_SUB: or_af_sub(&R[I->a], &R[I->b], &R[I->dst]); break;
		case OP_MUL: or_af_mul(&cx, &R[I->a], &R[I->b], &R[I->dst]); break;
		case OP_DIV: or_af_div(&cx, &R[I->a], &R[I->b], &R[I->dst]); break;
		case OP_INV: or_af_inv(&cx, &R[I->a], &R[I->dst]); break;
		case OP_NEG: or_af_neg(&R[I->a], &R[I->dst]); break;
		case OP_SCALE: or_af_scale(&R[I->a], I->imm0, &R[I->dst]); break;
		case OP_SIN: or_af_sin(&cx, &R[I->a], &R[I->dst]); break;
		case OP_COPY: R[I->dst] = R[I->a]; break;
		default: return -1;
		}
	}
	if (cx.overflow) return -3;
	for (int r = 0; r < nreg; r++) {
		center[r] = R[r].c;
		n[r] = R[r].n;
		or_af_convert(&R[r], &lo[r], &hi[r]);
		if (n[r] > cap) return -2;
		for (int k = 0; k < n[r]; k++) {
			idx[r * cap + k] = R[r].idx[k];
			coef[r * cap + k] = R[r].v[k];
		}
	}
	return 0;
}
