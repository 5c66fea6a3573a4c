/*
 * or_models.c -- the four example models the configs run.  TEST INFRASTRUCTURE (see or_oracle.h).
 *
 * The numeric constants ARE the workload: they are restated from the
 * reference's example files, and every expression keeps the reference's operand
 * order so each intermediate rounds the same way.
 * Layouts: Dh col-major npSS x nx (Dh[i + j*npSS]); g col-major nx x nu; Df col-major
 * nx x nx; Dg index i + k*nx + j*nx*nu; Du col-major nu x nx.
 */
#include "or_internal.h"
#include <math.h>
#include <string.h>

/* include/asif_utils.h:46-62, restated: Ab[i] = 0; Ab[i] = Ab[i] + A[i+k*nl]*b[k] */
void or_matvec(const double *A, int nl, int ncol, const double *b, double *Ab)
{
	for (int i = 0; i < nl; i++) {
		Ab[i] = 0.0;
		for (int k = 0; k < ncol; k++) Ab[i] = Ab[i] + A[i + k * nl] * b[k];
	}
}

/* include/asif_utils.h:22-44 */
void or_matmul(const double *A, int nlA, int ncA, const double *B, int ncB, double *AB)
{
	for (int i = 0; i < nlA; i++)
		for (int j = 0; j < ncB; j++) {
			int id = i + j * nlA;
			AB[id] = 0.0;
			for (int k = 0; k < ncA; k++) AB[id] = AB[id] + A[i + k * nlA] * B[k + j * ncA];
		}
}

/* include/asif_utils.h:64-73 */
double or_vecnorm(const double *v, int len)
{
	double tmp = 0;
	for (int i = 0; i < len; i++) tmp += v[i] * v[i];
	return sqrt(tmp);
}

/* ----------------------------------------------------------------------------
 * Double integrator -- examples/DoubleIntegrator.cpp:12-61
 * x = (position, velocity); safe box +-1 with braking-distance parabolas. */
static const double di_xb[2] = {-1.0, 1.0};
static const double di_vb[2] = {-1.0, 1.0};

static void di_safety(const void *ud, const double *x, double *h, double *Dh)
{
	(void)ud;
	/* :24-38 -- which side carries the v^2/2 term depends on the sign of v */
	if (x[1] > 0) {
		h[0] = di_xb[1] - x[0] - (x[1] * x[1]) / 2.0;
		Dh[0] = -1.0;
		Dh[4] = -x[1];
		h[1] = x[0] - di_xb[0];
		Dh[1] = 1.0;
		Dh[5] = 0.0;
	} else {
		h[0] = -x[0] + di_xb[1];
		Dh[0] = -1.0;
		Dh[4] = 0.0;
		h[1] = x[0] - di_xb[0] - (x[1] * x[1]) / 2.0;
		Dh[1] = 1.0;
		Dh[5] = -x[1];
	}
	h[2] = x[1] - di_vb[0];
	Dh[2] = 0.0;
	Dh[6] = 1.0;
	h[3] = -x[1] + di_vb[1];
	Dh[3] = 0.0;
	Dh[7] = -1.0;
}

static void di_dynamics(const void *ud, const double *x, double *f, double *g)
{
	(void)ud;
	/* :40-61 -- f = A x accumulated from 0.0, A = [0 1; 0 0] col-major, g = (0,1) */
	static const double Amat[4] = {0.0, 0.0, 1.0, 0.0};
	static const double bvec[2] = {0.0, 1.0};
	for (int i = 0; i < 2; i++) {
		f[i] = 0.0;
		for (int j = 0; j < 2; j++) f[i] += Amat[i + j * 2] * x[j];
	}
	g[0] = bvec[0];
	g[1] = bvec[1];
}

/* ----------------------------------------------------------------------------
 * Inverted pendulum -- examples/InvertedPendulum_Implicit.cpp:13-80 */
static const double ip_K[2] = {-3.0, -3.0};
static const double ip_P[4] = {1.25, 0.25, 0.25, 0.25};
static const double ip_mPpPt[4] = {-2.5, -0.5, -0.5, -0.5};
static const double ip_Pv = 0.05;

static void ip_safety(const void *ud, const double *x, double *h, double *Dh)
{
	(void)ud;
	const double lo = -M_PI, hi = M_PI; /* xBound = vBound = {-pi, pi}, :21-22 */
	h[0] = -x[0] + hi; Dh[0] = -1.0; Dh[4] = 0.0;
	h[1] = x[0] - lo;  Dh[1] = 1.0;  Dh[5] = 0.0;
	h[2] = x[1] - lo;  Dh[2] = 0.0;  Dh[6] = 1.0;
	h[3] = -x[1] + hi; Dh[3] = 0.0;  Dh[7] = -1.0;
}

/* the box safety set above with interval_t operands: -x + hi is unary minus then "+ double",
 * x - lo is "- double" (lib/libaffa/src/aa_aafarithm.cpp: constants move the central value only) */
static void box_safety_af(const or_af *x, double lo, double hi, or_af *h)
{
	or_af cl, ch, t;
	or_af_const(&cl, lo);
	or_af_const(&ch, hi);
	or_af_neg(&x[0], &t);
	or_af_add(&t, &ch, &h[0]);
	or_af_sub(&x[0], &cl, &h[1]);
	or_af_sub(&x[1], &cl, &h[2]);
	or_af_neg(&x[1], &t);
	or_af_add(&t, &ch, &h[3]);
}

static void ip_safety_af(const void *ud, or_af_ctx *cx, const or_af *x, or_af *h)
{
	(void)ud;
	(void)cx;
	box_safety_af(x, -M_PI, M_PI, h);
}

static void ip_backup(const void *ud, const double *x, double *h, double *Dh, double *DDh)
{
	(void)ud;
	(void)DDh;
	/* :39-52 -- h = Pv - x'Px accumulated i outer / j inner; Dh = -(P+P')x */
	h[0] = ip_Pv;
	for (int i = 0; i < 2; i++)
		for (int j = 0; j < 2; j++) h[0] -= ip_P[i + j * 2] * x[i] * x[j];
	or_matvec(ip_mPpPt, 2, 2, x, Dh);
}

static void ip_dynamics(const void *ud, const double *x, double *f, double *g)
{
	(void)ud;
	f[0] = x[1];
	f[1] = sin(x[0]);
	g[0] = 0.;
	g[1] = 1.;
}

static void ip_ctrl(const void *ud, const double *x, double *u, double *Du)
{
	(void)ud;
	or_matvec(ip_K, 1, 2, x, u); /* :63-71 */
	Du[0] = ip_K[0];
	Du[1] = ip_K[1];
}

static void ip_grad(const void *ud, const double *x, double *Df, double *Dg)
{
	(void)ud;
	Df[0] = 0.;        Df[2] = 1.;
	Df[1] = cos(x[0]); Df[3] = 0.;
	for (int i = 0; i < 4; i++) Dg[i] = 0.0;
}

/* ----------------------------------------------------------------------------
 * Segway -- examples/segway_implicit_tb.cpp:13-212 (MATLAB-generated dynamics).
 * The friction factor in f is multiplied by 0.0 in the reference (:78), so every
 * term carrying it is a signed zero; they are kept as literal 0.0*... products only
 * where dropping them could change a rounding, i.e. nowhere: x + (+-0) == x. The
 * surviving terms keep their left-to-right association. */
static const double sg_xb[4] = {3.0, 3.0, M_PI / 6, M_PI};
static const double sg_K[4] = {44.7214, 44.6528, 150.1612, 37.6492};
static const double sg_Pv = 0.05;

static void sg_safety(const void *ud, const double *x, double *h, double *Dh)
{
	(void)ud;
	for (int i = 0; i < 16; i++) Dh[i] = 0.0;
	for (int i = 0; i < 4; i++) {
		h[i] = (sg_xb[i] * sg_xb[i]) - (x[i] * x[i]);
		Dh[i * 5] = -2.0 * x[i];
	}
}

static void sg_backup(const void *ud, const double *x, double *h, double *Dh, double *DDh)
{
	(void)ud;
	for (int i = 0; i < 16; i++) DDh[i] = 0.0;
	h[0] = sg_Pv * sg_Pv;
	for (int i = 0; i < 4; i++) {
		h[0] -= (x[i] / sg_xb[i]) * (x[i] / sg_xb[i]);
		Dh[i] = -2.0 * x[i] / (sg_xb[i] * sg_xb[i]);
		DDh[i * 5] = -2.0 / (sg_xb[i] * sg_xb[i]);
	}
}

static void sg_ctrl(const void *ud, const double *x, double *u, double *Du)
{
	(void)ud;
	double xt[4] = {0., 0., -0.1383244254, 0.}; /* :57-68 equilibrium offset */
	for (int i = 0; i < 4; i++) xt[i] += x[i];
	or_matvec(sg_K, 1, 4, xt, u);
	for (int i = 0; i < 4; i++) Du[i] = sg_K[i];
}

static void sg_dynamics(const void *ud, const double *X, double *f, double *g)
{
	(void)ud;
	/* :70-111 */
	const double Fric = 0.0 * 2.595498 * tanh(X[1] / 0.001);
	const double w2 = X[3] * X[3];
	const double s1 = sin(X[2]);
	const double s2 = sin(2.0 * X[2]);
	const double c2 = cos(2.0 * X[2]);
	const double c1 = cos(X[2]);
	const double iden = 1.0 / ((14.553176960783997 + -2.0831375273848773 * c2) + -0.59146430898882 * s2);
	f[0] = X[1];
	f[1] = 0.0975 * ((((((((((-23.195670626755415 * Fric + -0.0043160179477503974 * Fric * 44.798) +
	                          -0.22270033964034344 * Fric * 44.798) +
	                         44.798 * (((-1.3347669149041519 * Fric + -0.2693850964936445 * w2) +
	                                    -0.0022454764220255392 * w2) + -0.11586336477125109 * w2) * 0.195 * c1) +
	                        59.510408935182809 * c2) + -0.185817500742 * Fric * 44.798 * 0.195 * s1) +
	                      86.686408318784913 * w2 * 0.195 * s1) + 0.72258001100852454 * w2 * 0.195 * s1) +
	                    37.284092841364554 * w2 * 0.195 * s1) + 4.1423245261005457 * s2) +
	                  -213.73800805067131 * s2) * iden;
	f[2] = X[3];
	f[3] = iden * ((((((((((8.0 * Fric * 0.055936595310797 + 4.0 * Fric * 44.798 * 0.038025) +
	                        8.0 * Fric * 2.485 * 0.038025) +
	                       89.596 * (0.333691728726038 * Fric * 0.195 + -0.45669752988922296) * c1) +
	                      15.554616935932147 * w2 * 0.038025 * c2) + 16.405863695295427 * s1) +
	                    0.092908750371 * Fric * 44.798 * 0.195 * s1) + 249.80488266222164 * s1) +
	                  27.713966400983114 * s1) + 1.0827059060875992 * w2 * 0.038025 * s2) +
	                -55.866072832711595 * w2 * 0.038025 * s2);
	g[0] = 0.0;
	const double gc = 1.4575004011882324 * c1;
	const double gs = 0.20290365220710288 * s1;
	g[1] = 0.551244194154502 * ((4.1706936767483551 + gc) + gs) *
	       (1.0 / (((8.3593271361634187 + -2.1243074194638587 * (c1 * c1)) + -0.04116989207898096 * (s1 * s1)) +
	               -0.29573215449441 * s2));
	g[2] = 0.0;
	g[3] = -5.65378660671284 * ((2.0043013906215941 + gc) + gs) * iden;
}

static void sg_grad(const void *ud, const double *x, double *Df, double *Dg)
{
	(void)ud;
	/* :113-212.  Note the Jacobian was generated WITH the tanh friction term although
	 * f above has it zeroed; reproduce the arithmetic, not the intent. */
	const double c1 = cos(x[2]);
	const double s1 = sin(x[2]);
	const double a2 = x[2] * 2.0;
	const double w2 = x[3] * x[3];
	const double c2 = cos(a2);
	const double s2 = sin(a2);
	const double th = tanh(x[1] * 1000.0);
	const double th2 = th * th;
	const double t25 = th * 15.13175750513302 - 40.918271887954823;
	const double t26 = w2 * 3.3849959169972448 + th * 30.26351501026604;
	const double t23 = 1.0 / ((c2 * 2.0831375273848769 + s2 * 0.59146430898882) - 14.553176960784);
	Df[0] = 0.0;
	Df[1] = 0.0;
	Df[2] = 0.0;
	Df[3] = 0.0;
	Df[4] = 1.0;
	const double e1 = s1 * (th2 * 1000.0 - 1000.0);
	Df[5] = -t23 * (((th2 * 8443.5211353581435 + e1 * 0.41077609832706019) +
	                 c1 * (th2 * 30263.515010266041 - 30263.515010266041) * 0.0975) - 8443.5211353581435);
	Df[6] = 0.0;
	Df[7] = t23 * (((th2 * 20808.641003022261 + e1 * 2.1065440939849238) +
	                c1 * (th2 * 15131.75750513302 - 15131.75750513302)) - 20808.641003022261);
	Df[8] = 0.0;
	const double cth = c1 * th;
	const double sth = s1 * th;
	const double e3 = (c2 * 1.18292861797764 + -(s2 * 4.1662750547697547)) * (t23 * t23);
	Df[9] = t23 * ((((c2 * 40.8711582872913 + s2 * 11.604529742360651) - c1 * w2 * 2.3707272057666411) +
	                cth * 0.41077609832706019) - s1 * t26 * 0.0975) -
	        e3 * (((((c2 * -5.8022648711803244 + s2 * 20.435579143645651) + th * 8.443521135358143) -
	                s1 * w2 * 2.3707272057666411) + sth * 0.41077609832706019) + c1 * t26 * 0.0975);
	Df[10] = 0.0;
	const double wc = w2 * c2;
	const double ws = w2 * s2;
	Df[11] = t23 * ((((c1 * -293.92471275850022 - cth * 2.1065440939849238) + wc * 4.1662750547697547) +
	                 ws * 1.18292861797764) + s1 * t25) +
	         e3 * (((((s1 * 293.92471275850022 + th * 20.808641003022259) + wc * 0.59146430898881985) +
	                 sth * 2.1065440939849238) - ws * 2.0831375273848769) + c1 * t25);
	Df[12] = 0.0;
	Df[13] = t23 * (c1 * x[3] * 0.6600742038144628 - s1 * x[3] * 4.7414544115332831);
	Df[14] = 1.0;
	Df[15] = -t23 * (c2 * x[3] * 1.18292861797764 - s2 * x[3] * 4.1662750547697547);

	const double d4 = (c2 * 2.0831375273848769 + s2 * 0.59146430898882) - 14.553176960784;
	const double d26 = ((c1 * c1 * 2.1243074194638591 + s2 * 0.29573215449441) + s1 * s1 * 0.04116989207898096) -
	                   8.3593271361634187;
	for (int i = 0; i < 16; i++) Dg[i] = 0.0;
	Dg[9] = -(c1 * 0.1118494602519098 - s1 * 0.80343863413287053) / d26 +
	        1.0 / (d26 * d26) * (c2 * 0.59146430898882 - c1 * s1 * 4.1662750547697547) *
	            ((c1 * 0.80343863413287053 + s1 * 0.1118494602519098) + 2.2990706749044238);
	Dg[11] = (c1 * 1.1471739513016379 - s1 * 8.24039624751662) / d4 -
	         1.0 / (d4 * d4) * (c2 * 1.18292861797764 - s2 * 4.1662750547697547) *
	             ((c1 * 8.24039624751662 + s1 * 1.1471739513016379) + 11.33189235811229);
}

/* ----------------------------------------------------------------------------
 * Robust inverted pendulum -- examples/InvertedPendulum_Robust.cpp:20-79.
 * Half-plane safety set 1 - a.x >= 0 (data supplied through or_options because the
 * shipped SafetySetData vector is empty, :51); g[1] = [pMin,pMax]. */
static void ipr_safety(const void *ud, const double *x, double *h, double *Dh)
{
	const or_options *o = (const or_options *)ud;
	const int N = o->nHalfPlanes;
	for (int i = 0; i < N; i++) {
		h[i] = 1. - o->halfPlanes[2 * i] * x[0] - o->halfPlanes[2 * i + 1] * x[1];
		Dh[i] = -o->halfPlanes[2 * i];
		Dh[i + N] = -o->halfPlanes[2 * i + 1];
	}
}

/* :62-69 -- f[0] = x[1] (copy incl. its symbol), f[1] = sin(x[0]), g[0] = 0. (no symbol),
 * g[1] = interval(pMin,pMax) (new symbol).  Symbol creation order is the statement order. */
static void ipr_dynamics_af(const void *ud, or_af_ctx *cx, const or_af *x, or_af *f, or_af *g)
{
	const or_options *o = (const or_options *)ud;
	f[0] = x[1];
	or_af_sin(cx, &x[0], &f[1]);
	or_af_const(&g[0], 0.);
	or_af_interval(cx, &g[1], o->pMin, o->pMax);
}

/* ----------------------------------------------------------------------------
 * Inverted pendulum for the time-to-backup-set filter -- examples/InvertedPendulum_ImplicitTB.cpp:14-99:
 * asymmetric box, half-space backup set x0 >= pi/2 - 0.1, velocity-tracking backup controller. */
static void ipt_safety(const void *ud, const double *x, double *h, double *Dh)
{
	(void)ud;
	const double xlo = -M_PI / 2., xhi = M_PI, vlo = -M_PI / 2., vhi = M_PI / 2.; /* :23-24 */
	h[0] = -x[0] + xhi; Dh[0] = -1.0; Dh[4] = 0.0;
	h[1] = x[0] - xlo;  Dh[1] = 1.0;  Dh[5] = 0.0;
	h[2] = x[1] - vlo;  Dh[2] = 0.0;  Dh[6] = 1.0;
	h[3] = -x[1] + vhi; Dh[3] = 0.0;  Dh[7] = -1.0;
}

static void ipt_backup(const void *ud, const double *x, double *h, double *Dh, double *DDh)
{
	(void)ud;
	const double x0 = M_PI / 2.; /* :36-65 */
	h[0] = x[0] - x0 + 0.1;
	Dh[0] = 1.;
	Dh[1] = 0.;
	if (DDh)
		for (int i = 0; i < 4; i++) DDh[i] = 0.;
}

static void ipt_ctrl(const void *ud, const double *x, double *u, double *Du)
{
	(void)ud;
	const double vDes = (M_PI / 10.), K = 10.; /* :76-85 */
	u[0] = K * (vDes - x[1]);
	Du[0] = 0.;
	Du[1] = -K;
}

/* ----------------------------------------------------------------------------
 * Double integrator with an LQR-like backup controller -- examples/DoubleIntegrator_implicit.cpp:13-90
 * (plain box, unlike examples/DoubleIntegrator.cpp; ellipsoidal backup set). */
static const double dii_K[2] = {-10.0, -20};
static const double dii_P[4] = {0.500000000000000, 0.288675134594813, 0.288675134594813, 0.5};
/* shipped as {-1, -0.577.., -0.577.., +1}: the last entry is not -(P+P')(1,1) = -1 (:30); reproduced as is */
static const double dii_mPpPt[4] = {-1.0, -0.577350269189626, -0.577350269189626, 1.0};
static const double dii_Pv = 0.002;

static void dii_safety(const void *ud, const double *x, double *h, double *Dh)
{
	(void)ud;
	const double lo = -1.0, hi = 1.0; /* xBound = vBound = {-1, 1}, :21-22 */
	h[0] = -x[0] + hi; Dh[0] = -1.0; Dh[4] = 0.0;
	h[1] = x[0] - lo;  Dh[1] = 1.0;  Dh[5] = 0.0;
	h[2] = x[1] - lo;  Dh[2] = 0.0;  Dh[6] = 1.0;
	h[3] = -x[1] + hi; Dh[3] = 0.0;  Dh[7] = -1.0;
}

static void dii_safety_af(const void *ud, or_af_ctx *cx, const or_af *x, or_af *h)
{
	(void)ud;
	(void)cx;
	box_safety_af(x, -1.0, 1.0, h);
}

static void dii_backup(const void *ud, const double *x, double *h, double *Dh, double *DDh)
{
	(void)ud;
	(void)DDh;
	h[0] = dii_Pv; /* :42-55 */
	for (int i = 0; i < 2; i++)
		for (int j = 0; j < 2; j++) h[0] -= dii_P[i + j * 2] * x[i] * x[j];
	or_matvec(dii_mPpPt, 2, 2, x, Dh);
}

static void dii_ctrl(const void *ud, const double *x, double *u, double *Du)
{
	(void)ud;
	or_matvec(dii_K, 1, 2, x, u); /* :65-73 */
	Du[0] = dii_K[0];
	Du[1] = dii_K[1];
}

static void dii_grad(const void *ud, const double *x, double *Df, double *Dg)
{
	(void)ud;
	(void)x;
	Df[0] = 0.0; Df[1] = 0.0; Df[2] = 1.0; Df[3] = 0.0; /* A, :75-80 */
	for (int i = 0; i < 4; i++) Dg[i] = 0.0;
}

/* ----------------------------------------------------------------------------
 * Double integrator under the time-to-backup-set filter -- examples/DoubleIntegrator_implicit_tb.cpp:13-103:
 * the box, controller and dynamics of the implicit example above, a circular backup set x'x <= Pv^2.
 * The example's backupSet writes DDh[i] = mPpPt[i] for i < nx only (:49), i.e. two of the four entries of the
 * Hessian, and the class reads all four from an uninitialised stack array (src/asif_implicit_tb.cpp:494,623):
 * the last two are indeterminate in the reference.  Restated with the whole of mPpPt, the matrix that line
 * indexes (UNPINNED by construction: no other choice is the reference's either). */
static const double dit_P[4] = {1.0, 0.0, 0.0, 1.0};
static const double dit_mPpPt[4] = {-2.0, 0.0, 0.0, -2.0};
static const double dit_Pv = 0.01;

static void dit_backup(const void *ud, const double *x, double *h, double *Dh, double *DDh)
{
	(void)ud;
	h[0] = dit_Pv * dit_Pv; /* :41-57 */
	for (int i = 0; i < 2; i++)
		for (int j = 0; j < 2; j++) h[0] -= dit_P[i + j * 2] * x[i] * x[j];
	or_matvec(dit_mPpPt, 2, 2, x, Dh);
	if (DDh)
		for (int i = 0; i < 4; i++) DDh[i] = dit_mPpPt[i];
}

/* ----------------------------------------------------------------------------
 * Synthetic two-input model for class ASIF (NOT from the reference: none of its examples has nu > 1, but
 * src/asif.cpp is written for any nu): x' = F x + G u with a non-diagonal input matrix, five half-planes. */
static void p2_safety(const void *ud, const double *x, double *h, double *Dh)
{
	(void)ud;
	const double a[5][2] = {{1., 0.}, {-1., 0.}, {0., 1.}, {0., -1.}, {0.6, 0.8}}, r[5] = {1., 1., 1., 1., 1.2};
	for (int i = 0; i < 5; i++) {
		h[i] = r[i] - a[i][0] * x[0] - a[i][1] * x[1];
		Dh[i] = -a[i][0];
		Dh[i + 5] = -a[i][1];
	}
}
static void p2_dynamics(const void *ud, const double *x, double *f, double *g)
{
	(void)ud;
	f[0] = -0.5 * x[0] + 0.2 * x[1];
	f[1] = 0.1 * x[0] + -0.3 * x[1];
	g[0] = 1.0; g[1] = 0.0; /* column 0 */
	g[2] = 0.3; g[3] = 1.0; /* column 1 */
}

static const or_model MODELS[8] = {
    {2, 1, 4, 0, di_safety, 0, di_dynamics, 0, 0, 0, 0, 0},
    {2, 1, 4, 1, ip_safety, ip_backup, ip_dynamics, ip_grad, ip_ctrl, 0, ip_safety_af, 10}, /* examples/InvertedPendulum_Implicit.cpp:17 */
    {4, 1, 4, 1, sg_safety, sg_backup, sg_dynamics, sg_grad, sg_ctrl, 0, 0, 4},
    {2, 1, 0, 0, ipr_safety, 0, 0, 0, 0, ipr_dynamics_af, 0, 0},
    {2, 1, 4, 1, ipt_safety, ipt_backup, ip_dynamics, ip_grad, ipt_ctrl, 0, 0, 4}, /* dynamics :67-74,87-94 = the pendulum's */
    {2, 1, 4, 1, dii_safety, dii_backup, di_dynamics, dii_grad, dii_ctrl, 0, dii_safety_af, 4}, /* dynamics :57-63 = A x, B; npBTSS :17 */
    {2, 2, 5, 0, p2_safety, 0, p2_dynamics, 0, 0, 0, 0, 0},
    {2, 1, 4, 1, dii_safety, dit_backup, di_dynamics, dii_grad, dii_ctrl, 0, 0, 4}, /* box :33-39, A x, B :59-65, K :67-74 as the implicit example; npBTSS :16 */
};

const or_model *or_model_get(int id)
{
	if (id < 0 || id > 7) return 0;
	return &MODELS[id];
}
