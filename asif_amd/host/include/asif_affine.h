// asif_affine.h -- host affine arithmetic for the robust filter's model callbacks: the subset of
// libaffa's `AAF` / `interval` interface that the reference's robust examples use
// (examples/InvertedPendulum_Robust.cpp:62-69, examples/DoubleIntegrator_Robust.cpp:45-52):
// AAF(double), AAF(interval), + - * / between forms and with doubles, unary -, sin, convert(), rad().
// Same arithmetic as lib/libaffa/src (see asif_amd/csrc/affine_dev.hpp for the line-by-line citations):
// round-to-nearest, a fresh noise symbol per interval and per non-affine operation, zero coefficients
// kept.  Fixed capacity, no heap; the symbol counter is per thread instead of one process-wide static.
// Inside the reference tree, include its own "aa.h" instead -- the filter class only needs the names.
#pragma once
#include <cmath>

class interval {
public:
	interval() : lo(0), hi(0) {}
	interval(double m) : lo(m), hi(m) {}
	interval(double l, double h) : lo(l), hi(h) {}
	double left() const { return lo; }
	double right() const { return hi; }
	double width() const { return hi - lo; }
	double mid() const { return lo * 0.5 + hi * 0.5; }

private:
	double lo, hi;
};

class AAF {
public:
	static constexpr int kCap = 48;
	AAF(double v0 = 0) : c_(v0), n_(0), overflow_(false), special_(kAffine) {}
	AAF(interval iv) : c_((iv.right() + iv.left()) / 2), n_(1), overflow_(false), special_(kAffine)
	{
		idx_[0] = ++last();
		v_[0] = (iv.right() - iv.left()) / 2;
		if (iv.right() - iv.left() == HUGE_VAL) { // aa_aafcommon.cpp:81-100: an unbounded interval
			c_ = 0;
			v_[0] = HUGE_VAL;
			special_ = kInfinite;
		}
	}
	AAF(double lo, double hi) : AAF(interval(lo, hi)) {}

	double get_center() const { return c_; }
	unsigned get_length() const { return (unsigned)n_; }
	double get_coeff(unsigned k) const { return v_[k]; }     // as libaffa's AAF::get_coeff / get_index
	unsigned get_index(unsigned k) const { return idx_[k]; }
	bool overflowed() const { return overflow_; }
	double rad() const
	{
		double s = 0;
		for (int i = 0; i < n_; i++) s += std::fabs(v_[i]);
		return s;
	}
	interval convert() const // aa_aafcommon.cpp:217-226: indeterminate forms convert to the whole line
	{
		const double r = rad();
		if ((special_ & (kInfinite | kNan)) || r == HUGE_VAL) return interval(-HUGE_VAL, HUGE_VAL);
		return interval(c_ - r, c_ + r);
	}
	static void set_default(unsigned v = 0) { last() = v; }

	AAF operator+(const AAF &p) const { return merged(p, 0); }
	AAF operator-(const AAF &p) const { return merged(p, 1); }
	AAF operator*(const AAF &p) const
	{
		AAF t = merged(p, 2);
		t.c_ = c_ * p.c_;
		t.push(rad() * p.rad());
		return t;
	}
	AAF operator/(const AAF &p) const { return (*this) * inv(p); }
	AAF operator-() const
	{
		AAF t(*this);
		t.c_ = -t.c_;
		for (int i = 0; i < t.n_; i++) t.v_[i] = -t.v_[i];
		return t;
	}
	AAF operator*(double k) const
	{
		AAF t(*this);
		t.c_ = k * c_;
		for (int i = 0; i < t.n_; i++) t.v_[i] = k * t.v_[i];
		return t;
	}
	friend AAF inv(const AAF &p)
	{
		if (p.special_ == kNan) return nanForm(); // aa_aafapprox.cpp:155-179
		if (p.special_ == kInfinite) return AAF(interval(-HUGE_VAL, HUGE_VAL));
		const interval iv = p.convert();
		double a = iv.left(), b = iv.right();
		if (a <= 0 && b >= 0) return AAF(interval(-HUGE_VAL, HUGE_VAL));
		const double t1 = std::fabs(a), t2 = std::fabs(b);
		a = t1 < t2 ? t1 : t2;
		b = t1 < t2 ? t2 : t1;
		const double alpha = -1 / (b * b);
		const double ilo = (1 / a) - alpha * a, ihi = 2 / b, mid = ilo * 0.5 + ihi * 0.5;
		const double r0 = mid - ilo, r1 = ihi - mid;
		return affine(p, alpha, iv.left() < 0 ? -mid : mid, r0 >= r1 ? r0 : r1);
	}
	friend AAF sin(const AAF &p)
	{
		const int NPTS = 8;
		if (p.special_ == kNan) return nanForm(); // aa_aaftrigo.cpp:42-135
		if (p.special_ == kInfinite) return AAF(interval(-1, 1));
		const interval iv = p.convert();
		const double a = iv.left(), b = iv.right(), w = b - a;
		if (w >= 2 * (4 * std::atan(1.0))) return AAF(interval(-1, 1));
		if (w < 1e-10) {
			const double t = std::sin(iv.mid());
			return AAF(interval(t, t));
		}
		double x[NPTS], y[NPTS];
		x[0] = a; y[0] = std::sin(a);
		x[NPTS - 1] = b; y[NPTS - 1] = std::sin(b);
		const double pas = w / (NPTS - 1);
		for (int i = 1; i < NPTS - 1; i++) { x[i] = x[i - 1] + pas; y[i] = std::sin(x[i]); }
		double xm = 0, ym = 0;
		for (int i = 0; i < NPTS; i++) { xm = xm + x[i]; ym = ym + y[i]; }
		xm = xm / NPTS;
		ym = ym / NPTS;
		double t2 = 0, alpha = 0;
		for (int i = 0; i < NPTS; i++) {
			const double t1 = x[i] - xm;
			alpha += y[i] * t1;
			t2 += t1 * t1;
		}
		alpha = alpha / t2;
		const double dzeta = ym - alpha * xm;
		double delta = 0;
		for (int i = 0; i < NPTS; i++) delta = std::fmax(delta, std::fabs(y[i] - (dzeta + alpha * x[i])));
		return affine(p, alpha, dzeta, delta);
	}

private:
	enum { kAffine = 1, kInfinite = 2, kNan = 4 }; // AAF_TYPE, aa_util.h
	static int binarySpecial(int a, int b)         // aa_util.h:31-44
	{
		if (a == kAffine && b == kAffine) return kAffine;
		if (a == kNan || b == kNan) return kNan;
		if (a == (kNan | kAffine) || b == (kNan | kAffine)) return kNan | kAffine;
		return a | b;
	}
	static AAF nanForm()
	{
		AAF t(0.0);
		t.special_ = kNan;
		return t;
	}
	double c_;
	int n_;
	bool overflow_;
	int special_;
	unsigned idx_[kCap];
	double v_[kCap];
	static unsigned &last()
	{
		static thread_local unsigned counter = 0;
		return counter;
	}
	void push(double coeff)
	{
		if (n_ >= kCap) { overflow_ = true; return; }
		idx_[n_] = ++last();
		v_[n_++] = coeff;
	}
	static AAF affine(const AAF &p, double alpha, double dzeta, double delta)
	{
		AAF t(p);
		t.c_ = alpha * p.c_ + dzeta;
		for (int i = 0; i < t.n_; i++) t.v_[i] = alpha * p.v_[i];
		t.push(delta);
		return t;
	}
	// sorted-index merge; mode 0: a+b, 1: a-b, 2: a.c*vb + b.c*va (centre set by the caller for 2)
	AAF merged(const AAF &p, int mode) const
	{
		AAF t(mode == 0 ? c_ + p.c_ : (mode == 1 ? c_ - p.c_ : 0.0));
		t.overflow_ = overflow_ || p.overflow_;
		t.special_ = binarySpecial(special_, p.special_);
		int ia = 0, ib = 0;
		while (ia < n_ || ib < p.n_) {
			if (t.n_ >= kCap) { t.overflow_ = true; break; }
			bool ta = false, tb = false;
			if (ia == n_) tb = true;
			else if (ib == p.n_) ta = true;
			else if (idx_[ia] < p.idx_[ib]) ta = true;
			else if (p.idx_[ib] < idx_[ia]) tb = true;
			else ta = tb = true;
			const double va = ta ? v_[ia] : 0.0, vb = tb ? p.v_[ib] : 0.0;
			t.idx_[t.n_] = ta ? idx_[ia] : p.idx_[ib];
			if (mode == 0) t.v_[t.n_] = (ta && tb) ? va + vb : (ta ? va : vb);
			else if (mode == 1) t.v_[t.n_] = (ta && tb) ? va - vb : (ta ? va : -vb);
			else t.v_[t.n_] = (ta && tb) ? c_ * vb + p.c_ * va : (ta ? p.c_ * va : c_ * vb);
			t.n_++;
			ia += ta;
			ib += tb;
		}
		return t;
	}
};

inline AAF operator*(double k, const AAF &p) { return p * k; }
inline AAF operator+(double k, const AAF &p) { return AAF(k) + p; }
inline AAF operator-(double k, const AAF &p) { return AAF(k) - p; }

typedef AAF interval_t; // include/asif_robust.h:7
