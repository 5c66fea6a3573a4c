// affine_dev.hpp -- fixed-capacity affine forms on the device, for the robust filter's interval
// Lie derivatives (src/asif_robust.cpp:282-337).
//
// Arithmetic of the subset of libaffa the reference reaches (lib/libaffa/src of the reference tree):
//   AAF(double)      aa_aaf.h:149-153           no noise symbol
//   AAF(interval)    aa_aafcommon.cpp:81-100    ALWAYS a fresh symbol, also for a point interval
//   + -              aa_aafarithm.cpp:35-167    sorted-index merge
//   * (AAF,AAF)      aa_aafapprox.cpp:34-101    one new symbol with coefficient rad(a)*rad(b), kept even if 0
//   unary -, *double aa_aafarithm.cpp:172-200
//   inv, /           aa_aafapprox.cpp:108-179   mini-range 1/x
//   sin              aa_aaftrigo.cpp:42-135     8-point least squares + max residual; width<1e-10 -> point
//   convert, rad     aa_aafcommon.cpp:217-245
// Round-to-nearest throughout, like the reference (its directed rounding is commented out,
// aa_interval.cpp:85-93).  No heap and no global state: the symbol counter (AAF::last in the
// reference, aa_aafcommon.cpp:32) is a per-lane register; only the relative order of symbols matters.
// Forms live in per-lane private memory; this path is assembly-only work of a few hundred flops.
// Every product and sum is rounded separately (no FMA contraction inside this header), as in the
// reference built for baseline x86-64: the results are bit-identical to libaffa's wherever libm agrees.
#pragma once
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)

namespace asif {

constexpr int kAfCap = 16;

struct Af {
	double c;
	int n;
	unsigned idx[kAfCap];
	double v[kAfCap];
};

struct AfCtx {
	unsigned last;
	bool overflow;
};

__device__ inline void af_const(Af &r, double v0)
{
	r.c = v0;
	r.n = 0;
}

__device__ inline void af_interval(AfCtx &cx, Af &r, double lo, double hi)
{
	r.n = 1;
	r.idx[0] = ++cx.last;
	r.c = (hi + lo) / 2;
	r.v[0] = (hi - lo) / 2;
	if (hi - lo == __builtin_huge_val()) { // unbounded interval (inv of a form straddling zero), aa_aafcommon.cpp:81-100
		r.c = 0;
		r.v[0] = __builtin_huge_val();
	}
}

__device__ inline double af_rad(const Af &a)
{
	double s = 0;
	for (int i = 0; i < a.n; i++) s += fabs(a.v[i]);
	return s;
}

__device__ inline void af_convert(const Af &a, double &lo, double &hi)
{
	const double r = af_rad(a);
	lo = a.c - r;
	hi = a.c + r;
	// indeterminate form (unbounded, or NaN coefficients, which only descendants of an unbounded form carry --
	// libaffa tracks those with its AAF_TYPE flags): the whole line, aa_aafcommon.cpp:217-226
	if (!(r < __builtin_huge_val())) {
		lo = -__builtin_huge_val();
		hi = __builtin_huge_val();
	}
}

// mode 0: a+b, 1: a-b, 2: a.c*vb + b.c*va.  Returns the merged length or -1 on overflow.
__device__ inline int af_merge(const Af &a, const Af &b, Af &r, int mode)
{
	int ia = 0, ib = 0, k = 0;
	while (ia < a.n || ib < b.n) {
		if (k >= kAfCap) return -1;
		bool ta = false, tb = false;
		if (ia == a.n) tb = true;
		else if (ib == b.n) ta = true;
		else if (a.idx[ia] < b.idx[ib]) ta = true;
		else if (b.idx[ib] < a.idx[ia]) tb = true;
		else ta = tb = true;
		const double va = ta ? a.v[ia] : 0.0, vb = tb ? b.v[ib] : 0.0;
		r.idx[k] = ta ? a.idx[ia] : b.idx[ib];
		if (mode == 0) r.v[k] = (ta && tb) ? va + vb : (ta ? va : vb);
		else if (mode == 1) r.v[k] = (ta && tb) ? va - vb : (ta ? va : -vb);
		else r.v[k] = (ta && tb) ? a.c * vb + b.c * va : (ta ? b.c * va : a.c * vb);
		ia += ta ? 1 : 0;
		ib += tb ? 1 : 0;
		k++;
	}
	return k;
}

__device__ inline void af_add(AfCtx &cx, const Af &a, const Af &b, Af &r)
{
	Af t;
	t.c = a.c + b.c;
	t.n = af_merge(a, b, t, 0);
	if (t.n < 0) { t.n = 0; cx.overflow = true; }
	r = t;
}

__device__ inline void af_sub(AfCtx &cx, const Af &a, const Af &b, Af &r)
{
	Af t;
	t.c = a.c - b.c;
	t.n = af_merge(a, b, t, 1);
	if (t.n < 0) { t.n = 0; cx.overflow = true; }
	r = t;
}

__device__ inline void af_neg(const Af &a, Af &r)
{
	Af t = a;
	t.c = -t.c;
	for (int i = 0; i < t.n; i++) t.v[i] = -t.v[i];
	r = t;
}

__device__ inline void af_scale(const Af &a, double k, Af &r)
{
	Af t = a;
	t.c = k * a.c;
	for (int i = 0; i < t.n; i++) t.v[i] = k * t.v[i];
	r = t;
}

__device__ inline void af_mul(AfCtx &cx, const Af &a, const Af &b, Af &r)
{
	Af t;
	t.c = a.c * b.c;
	const int k = af_merge(a, b, t, 2);
	if (k < 0 || k >= kAfCap) {
		cx.overflow = true;
		t.n = 0;
		r = t;
		return;
	}
	t.idx[k] = ++cx.last;
	t.v[k] = af_rad(a) * af_rad(b);
	t.n = k + 1;
	r = t;
}

// z = alpha*P + dzeta with a new symbol delta (aa_aafarithm.cpp:233-261)
__device__ inline void af_affine(AfCtx &cx, const Af &p, double alpha, double dzeta, double delta, Af &r)
{
	Af t;
	if (p.n >= kAfCap) {
		cx.overflow = true;
		t.c = 0;
		t.n = 0;
		r = t;
		return;
	}
	t.c = alpha * p.c + dzeta;
	t.n = p.n + 1;
	for (int i = 0; i < p.n; i++) {
		t.idx[i] = p.idx[i];
		t.v[i] = alpha * p.v[i];
	}
	t.idx[p.n] = ++cx.last;
	t.v[p.n] = delta;
	r = t;
}

__device__ inline void af_inv(AfCtx &cx, const Af &p, Af &r)
{
	double a, b;
	af_convert(p, a, b);
	const double lo0 = a;
	if ((a <= 0) && (b >= 0)) { // straddles zero: the reference returns (-inf, inf)
		af_interval(cx, r, -__builtin_huge_val(), __builtin_huge_val());
		return;
	}
	const double t1 = fabs(a), t2 = fabs(b);
	a = t1 < t2 ? t1 : t2;
	b = t1 < t2 ? t2 : t1;
	const double alpha = -1 / (b * b);
	const double ilo = (1 / a) - alpha * a, ihi = 2 / b;
	const double mid = ilo * 0.5 + ihi * 0.5;
	const double r0 = mid - ilo, r1 = ihi - mid;
	const double radius = (r0 >= r1 ? r0 : r1);
	af_affine(cx, p, alpha, lo0 < 0 ? -mid : mid, radius, r);
}

__device__ inline void af_div(AfCtx &cx, const Af &a, const Af &b, Af &r)
{
	Af ib;
	af_inv(cx, b, ib);
	af_mul(cx, a, ib, r);
}

__device__ inline void af_sin(AfCtx &cx, const Af &p, Af &r)
{
	constexpr int NPTS = 8;
	const double PI2 = 2 * 3.14159265358979323846;
	double a, b;
	af_convert(p, a, b);
	const double w = b - a;
	if (w >= PI2) {
		af_interval(cx, r, -1, 1);
		return;
	}
	if (w < 1e-10) {
		const double tmp = sin(a * 0.5 + b * 0.5);
		af_interval(cx, r, tmp, tmp);
		return;
	}
	double x[NPTS], y[NPTS];
	x[0] = a;
	y[0] = sin(a);
	x[NPTS - 1] = b;
	y[NPTS - 1] = sin(b);
	const double pas = w / (NPTS - 1);
	for (int i = 1; i < NPTS - 1; i++) {
		x[i] = x[i - 1] + pas;
		y[i] = sin(x[i]);
	}
	double xm = 0, ym = 0;
	for (int i = 0; i < NPTS; i++) {
		xm = xm + x[i];
		ym = ym + y[i];
	}
	xm = xm / NPTS;
	ym = ym / NPTS;
	double temp2 = 0, alpha = 0;
	for (int i = 0; i < NPTS; i++) {
		const double temp1 = x[i] - xm;
		alpha += y[i] * temp1;
		temp2 += temp1 * temp1;
	}
	alpha = alpha / temp2;
	const double dzeta = ym - alpha * xm;
	double delta = 0;
	for (int i = 0; i < NPTS; i++) delta = fmax(delta, fabs(y[i] - (dzeta + alpha * x[i])));
	af_affine(cx, p, alpha, dzeta, delta, r);
}

} // namespace asif
#pragma clang fp contract(fast)
