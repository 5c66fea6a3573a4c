/*
 * or_assembly.c -- constraint-row assembly of the four filter variants.
 * TEST INFRASTRUCTURE (see or_oracle.h).  Each function names the reference lines it restates.
 */
#include "or_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static __thread int g_last_crit[16];
static __thread int g_last_ncrit;

int or_last_crit_idx(int *idx, int cap)
{
	int n = g_last_ncrit < cap ? g_last_ncrit : cap;
	for (int i = 0; i < n; i++) idx[i] = g_last_crit[i];
	return n;
}

/* Option values exactly as the example main()s set them before initialize():
 *  C2 examples/DoubleIntegrator.cpp:15-16,69-70 (initialize(lb,ub) -> struct defaults, include/asif.h:13-16)
 *  C3 examples/InvertedPendulum_Implicit.cpp:19-20,93-97
 *  C4 examples/segway_implicit_tb.cpp:18-19,223-230
 *  C5 examples/InvertedPendulum_Robust.cpp:23-24,35-38,120-121 (ROBUST build flavour) + SURVEY 8(d) half-planes */
void or_default_options(int model, int variant, or_options *o)
{
	memset(o, 0, sizeof(*o));
	o->relaxCost = 50.0;
	o->relaxLb = 5.0;
	o->relaxReachLb = 5.0;
	o->relaxTTS = 5.0;
	o->relaxMinOrtho = 5.0;
	o->backTrajHorizon = 1.0;
	o->backTrajExtend = 0.05;
	o->backTrajDt = 0.01;
	o->backTrajMinOrtho = 0.01;
	o->satSharpness = (variant == OR_VARIANT_EXPLICIT) ? 5.0 : 0.1;
	o->inf = 1e20;
	o->pMin = o->pMax = 1.0;
	o->backContDt = 0.01; /* include/asif_implicit_robust.h:31 */
	o->n_debug = -1;      /* :26 */
	o->npSSmax = -1;            /* include/asif.h:28 */
	o->backTrajAbsTol = 1.0e-6; /* include/asif_implicit.h:29-30 */
	o->backTrajRelTol = 1.0e-6;
	switch (model) {
	case OR_MODEL_PLANAR_TWO_INPUT: /* synthetic: both inputs in [-1, 1], the class defaults otherwise */
		o->lb[0] = o->lb[1] = -1.0;
		o->ub[0] = o->ub[1] = 1.0;
		break;
	case OR_MODEL_DOUBLE_INTEGRATOR:
		o->lb[0] = -1.0;
		o->ub[0] = 1.0;
		break;
	case OR_MODEL_INVERTED_PENDULUM:
		o->lb[0] = -1.5;
		o->ub[0] = 1.5;
		o->backTrajHorizon = 5.0;
		o->backTrajDt = 0.001;
		o->relaxReachLb = 5.0;
		o->relaxLb = 10.0;
		break;
	case OR_MODEL_SEGWAY:
		o->lb[0] = -20.0;
		o->ub[0] = 20.0;
		o->backTrajHorizon = 3.0;
		o->backTrajDt = 0.01;
		o->relaxCost = 10;
		o->relaxLb = 2.0;
		o->relaxTTS = 30.0;
		o->relaxMinOrtho = 60.0;
		o->backTrajMinOrtho = 0.001;
		break;
	case OR_MODEL_DOUBLE_INTEGRATOR_IMPLICIT: /* examples/DoubleIntegrator_implicit.cpp:19-20,89-93 */
		o->lb[0] = -1.0;
		o->ub[0] = 1.0;
		o->backTrajHorizon = 2.0;
		o->backTrajDt = 0.01;
		o->relaxReachLb = 5.0;
		o->relaxLb = 10.0;
		break;
	case OR_MODEL_INVERTED_PENDULUM_TB: /* examples/InvertedPendulum_ImplicitTB.cpp:19-22,106-114 */
		o->lb[0] = -1.5;
		o->ub[0] = 1.5;
		o->backTrajHorizon = 11.0;
		o->backTrajDt = 0.001;
		o->relaxCost = 10.;
		o->relaxLb = 10.0;
		o->relaxTTS = 30.0;
		o->relaxMinOrtho = 60.0;
		o->backTrajMinOrtho = 0.001;
		break;
	case OR_MODEL_DOUBLE_INTEGRATOR_TB: /* examples/DoubleIntegrator_implicit_tb.cpp:18-19,107-112 */
		o->lb[0] = -1.0;
		o->ub[0] = 1.0;
		o->backTrajHorizon = 2.0;
		o->backTrajDt = 0.001;
		o->relaxLb = 10.0;
		o->relaxTTS = 5.0;
		o->relaxMinOrtho = 5.0;
		break;
	case OR_MODEL_INVERTED_PENDULUM_ROBUST: {
		o->lb[0] = -1.5;
		o->ub[0] = 1.5;
		o->pMin = 0.8;
		o->pMax = 1.2;
		o->nHalfPlanes = 4;
		const double a = 1.0 / M_PI;
		const double hp[8] = {a, 0, -a, 0, 0, a, 0, -a};
		memcpy(o->halfPlanes, hp, sizeof(hp));
		break;
	}
	default:
		break;
	}
}

static int traj_len(int variant, const or_options *o, int npBTSS, double *dt_out)
{
	/* src/asif_implicit.cpp:211-216; src/asif_implicit_tb.cpp:177-182 (with the (1+extend) factor) */
	double T = o->backTrajHorizon;
	if (variant == OR_VARIANT_IMPLICIT_TB) T = o->backTrajHorizon * (1.0 + o->backTrajExtend);
	double dt = o->backTrajDt;
	int npBT = (int)(round(T / dt) + 1);
	if (npBT < npBTSS) {
		npBT = npBTSS;
		dt = T / (double)(npBT - 1);
	}
	if (dt_out) *dt_out = dt;
	return npBT;
}

int or_get_dims(int model, int variant, const or_options *o, or_dims *d)
{
	const or_model *m = or_model_get(model);
	if (!m) return -1;
	memset(d, 0, sizeof(*d));
	d->nx = m->nx;
	d->nu = m->nu;
	d->npSS = m->npSS;
	d->npBS = m->npBS;
	switch (variant) {
	case OR_VARIANT_EXPLICIT: /* src/asif.cpp:17-22 */
		if (!m->dynamics || model == OR_MODEL_INVERTED_PENDULUM_ROBUST) return -1;
		d->nv = m->nu + 1;
		d->nc = (o->npSSmax > 0 && o->npSSmax < m->npSS) ? o->npSSmax : m->npSS; /* src/asif.cpp:21 */
		d->nrelax = 1;
		break;
	case OR_VARIANT_IMPLICIT_RB: /* src/asif_implicit_robust.cpp:206-214: same shape as ASIFimplicit */
		if (!m->safety_af) return -1;
		/* fall through */
	case OR_VARIANT_IMPLICIT: /* src/asif_implicit.cpp:121-129 */
		if (!m->controller) return -1;
		d->npBTSS = m->npBTSS; /* examples/InvertedPendulum_Implicit.cpp:17, examples/DoubleIntegrator_implicit.cpp:17 */
		d->nv = m->nu + 2;
		d->nc = d->npBTSS * m->npSS + m->npBS;
		d->nrelax = 2;
		d->npBT = traj_len(variant, o, d->npBTSS, 0);
		break;
	case OR_VARIANT_IMPLICIT_TB: /* src/asif_implicit_tb.cpp:118-125 */
		if (!m->controller) return -1;
		d->npBTSS = 4; /* examples/segway_implicit_tb.cpp:16 */
		d->nv = m->nu + 1;
		d->nc = d->npBTSS * m->npSS + 2;
		d->nrelax = 1;
		d->npBT = traj_len(variant, o, d->npBTSS, 0);
		break;
	case OR_VARIANT_ROBUST: /* src/asif_robust.cpp:17-22, npSSmax = npSS */
		if (!m->dynamics_af) return -1;
		d->npSS = o->nHalfPlanes;
		d->nv = m->nu + 1 + d->npSS * 2 * (m->nu + 1);
		d->nc = d->npSS * (1 + (m->nu + 1));
		d->nrelax = 1;
		break;
	default:
		return -1;
	}
	return 0;
}

/* ---------------------------------------------------------------- explicit */
/* src/asif.cpp:233-312.  npSSmax < npSS: the rows of the npSSmax smallest h, in ascending order of h (:250-268;
 * std::sort leaves ties unspecified there, lowest index first here).  LfhIn / LghIn (may be NULL): the caller-
 * supplied Lie derivatives of filter(x, uDes, uAct, Lfh, Lgh) (:287-292), indexed by ROW. */
static __thread int g_kept[OR_MAX_NPSS], g_nkept;
int or_last_kept_rows(int *idx, int cap)
{
	const int n = g_nkept < cap ? g_nkept : cap;
	for (int i = 0; i < n; i++) idx[i] = g_kept[i];
	return g_nkept;
}
static int assemble_explicit_lie(const or_model *m, const or_options *o, const double *x, const double *LfhIn,
                                 const double *LghIn, double *A, double *b)
{
	const int nx = m->nx, nu = m->nu, npFull = m->npSS;
	const int np = (o->npSSmax > 0 && o->npSSmax < npFull) ? o->npSSmax : npFull, nc = np;
	double hFull[OR_MAX_NPSS], DhFull[OR_MAX_NPSS * OR_MAX_NX], f[OR_MAX_NX], g[OR_MAX_NX * OR_MAX_NU];
	double h[OR_MAX_NPSS], Dh[OR_MAX_NPSS * OR_MAX_NX], Lfh[OR_MAX_NPSS], Lgh[OR_MAX_NPSS * OR_MAX_NU];
	m->safety(o, x, hFull, DhFull);
	m->dynamics(o, x, f, g);
	int order[OR_MAX_NPSS];
	for (int i = 0; i < npFull; i++) order[i] = i;
	if (np < npFull)
		for (int i = 1; i < npFull; i++) { /* stable insertion sort by h */
			const int v = order[i];
			int j = i;
			while (j > 0 && hFull[order[j - 1]] > hFull[v]) {
				order[j] = order[j - 1];
				j--;
			}
			order[j] = v;
		}
	g_nkept = np;
	for (int i = 0; i < np; i++) {
		g_kept[i] = order[i];
		h[i] = hFull[order[i]];
		for (int j = 0; j < nx; j++) Dh[i + j * np] = DhFull[order[i] + j * npFull];
	}
	or_matvec(Dh, np, nx, f, Lfh);
	or_matmul(Dh, np, nx, g, nu, Lgh);
	if (LfhIn && LghIn) {
		for (int i = 0; i < np; i++) Lfh[i] = LfhIn[i];
		for (int i = 0; i < np * nu; i++) Lgh[i] = LghIn[i];
	}
	for (int i = 0; i < np; i++) {
		for (int j = 0; j < nu; j++) A[i + j * nc] = Lgh[i + j * np];
		A[i + nu * nc] = h[i];
		b[i] = -Lfh[i];
	}
	return 1;
}
static int assemble_explicit(const or_model *m, const or_options *o, const double *x, double *A, double *b)
{
	return assemble_explicit_lie(m, o, x, 0, 0, A, b);
}

int or_filter_explicit_lie(int model, const or_options *o, const double *x, const double *uDes, const double *Lfh,
                           const double *Lgh, double *uAct, double *relax)
{
	const or_model *m = or_model_get(model);
	or_dims d;
	if (!m || or_get_dims(model, OR_VARIANT_EXPLICIT, o, &d)) return -100;
	double A[OR_MAX_NPSS * (OR_MAX_NU + 1)], b[OR_MAX_NPSS], Hd[OR_MAX_NU + 1], c[OR_MAX_NU + 1], lb[OR_MAX_NU + 1],
	    ub[OR_MAX_NU + 1], sol[OR_MAX_NU + 1];
	uint8_t be[OR_MAX_NPSS];
	or_qp_static(model, OR_VARIANT_EXPLICIT, o, uDes, Hd, c, lb, ub, be);
	assemble_explicit_lie(m, o, x, Lfh, Lgh, A, b);
	or_qp qp = {d.nv, d.nc, Hd, c, A, b, lb, ub, be};
	if (or_qp_exact_small(&qp, sol) != 1) return -1; /* src/asif.cpp:208-209 */
	for (int i = 0; i < d.nu; i++) uAct[i] = sol[i] > o->ub[i] ? o->ub[i] : (sol[i] < o->lb[i] ? o->lb[i] : sol[i]);
	relax[0] = sol[d.nu];
	return 1;
}

/* ------------------------------------------------- backup closed loop (shared) */
/* src/asif_implicit.cpp:682-737 == src/asif_implicit_tb.cpp:764-819 */
static void saturate_soft(const or_model *m, const or_options *o, const double *u, double *uSat, double *DuSat)
{
	const double r = o->satSharpness;
	const double alpha = M_PI / 8;
	const double beta = M_PI / 4;
	for (int i = 0; i < m->nu; i++) {
		const double mi = o->lb[i], ma = o->ub[i];
		const double range = ma - mi;
		const double middle = (ma + mi) / 2;
		const double uc = 2 * (u[i] - middle) / range;
		const double bevelL = r * tan(alpha);
		const double bevelStart = 1 - cos(beta) * bevelL;
		const double bevelStop = 1 + bevelL;
		const double bevelXc = bevelStop;
		const double bevelYc = 1 - r;
		if (uc >= bevelStop) {
			uSat[i] = ma;
			DuSat[i] = 0;
		} else if (uc <= -bevelStop) {
			uSat[i] = mi;
			DuSat[i] = 0;
		} else if (uc <= bevelStart && uc >= -bevelStart) {
			uSat[i] = u[i];
			DuSat[i] = 1;
		} else if (uc > bevelStart) {
			uSat[i] = sqrt(r * r - (uc - bevelXc) * (uc - bevelXc)) + bevelYc;
			DuSat[i] = (bevelXc - uc) / sqrt(r * r - (uc - bevelXc) * (uc - bevelXc));
			uSat[i] = 0.5 * uSat[i] * range + middle;
		} else if (uc < -bevelStart) {
			uSat[i] = -sqrt(r * r - (uc + bevelXc) * (uc + bevelXc)) - bevelYc;
			DuSat[i] = (bevelXc + uc) / sqrt(r * r - (uc + bevelXc) * (uc + bevelXc));
			uSat[i] = 0.5 * uSat[i] * range + middle;
		} else { /* NaN input */
			DuSat[i] = 1;
			uSat[i] = u[i];
		}
	}
}

/* ASIFimplicitRB holds the backup input over backContDt along the trajectory
 * (src/asif_implicit_robust.cpp:891-903; members t_last_zoh_, u_zoh_, Du_zoh_) */
typedef struct {
	double dt;     /* options_.backTrajDt as initialize() left it */
	double t_last;
	double u[OR_MAX_NU], Du[OR_MAX_NU * OR_MAX_NX];
} zoh_t;

/* src/asif_implicit.cpp:751-815 (separate dynamics + dynamicsGradients branch, :789-807);
 * with zoh != NULL: src/asif_implicit_robust.cpp:878-953, same branch (:923-941) -- it saturates the held
 * input but keeps the FRESH Du in the sensitivity (Du_zoh_ is only read by the fused-gradient branch) */
static void backup_cl(const or_model *m, const or_options *o, const double *x, double *fCL, double *DfCL, zoh_t *zoh,
                      double t)
{
	const int nx = m->nx, nu = m->nu;
	double f[OR_MAX_NX], g[OR_MAX_NX * OR_MAX_NU], u[OR_MAX_NU], Du[OR_MAX_NU * OR_MAX_NX];
	double uSat[OR_MAX_NU], DuSat[OR_MAX_NU];
	double Df[OR_MAX_NX * OR_MAX_NX], Dg[OR_MAX_NX * OR_MAX_NU * OR_MAX_NX];
	m->controller(o, x, u, Du);
	if (zoh) {
		if (t <= zoh->dt) zoh->t_last = -1.;
		if (t >= (zoh->t_last + o->backContDt - 0.0001)) {
			for (int i = 0; i < nu; i++) zoh->u[i] = u[i];
			for (int i = 0; i < nu * nx; i++) zoh->Du[i] = Du[i];
			zoh->t_last = t;
		}
		saturate_soft(m, o, zoh->u, uSat, DuSat);
	} else
		saturate_soft(m, o, u, uSat, DuSat);
	m->dynamics(o, x, f, g);
	m->gradients(o, x, Df, Dg);
	for (int i = 0; i < nx; i++)
		for (int j = 0; j < nx; j++) {
			const int id = i + j * nx;
			DfCL[id] = Df[id];
			for (int k = 0; k < nu; k++)
				DfCL[id] += Dg[i + k * nx + j * nx * nu] * uSat[k] + g[i + k * nx] * DuSat[k] * Du[k + j * nu];
		}
	or_matvec(g, nx, nu, uSat, fCL);
	for (int i = 0; i < nx; i++) fCL[i] += f[i];
}

/* src/asif_implicit.cpp:817-827: z = [x; vec Q], zdot = [fCL; DfCL*Q] */
static void ode_rhs(const or_model *m, const or_options *o, const double *z, double *zdot, zoh_t *zoh, double t)
{
	double DfCL[OR_MAX_NX * OR_MAX_NX];
	backup_cl(m, o, z, zdot, DfCL, zoh, t);
	or_matmul(DfCL, m->nx, m->nx, z + m->nx, m->nx, zdot + m->nx);
}

/* ---- the reference's USE_ODEINT build (src/asif_implicit.cpp:427-460): boost::numeric::odeint
 *   make_dense_output(backTrajAbsTol, backTrajRelTol, runge_kutta_dopri5<state_t>())
 * observed at t_k = k*backTrajDt through an n_step_iterator.  Boost is NOT in /root/reference and NOT in this
 * image: restated from the published method (Dormand & Prince 1980 tableau; Shampine's / Hairer's continuous
 * extension as odeint's runge_kutta_dopri5::calc_state evaluates it) and from memory of odeint's controller
 * (controlled_runge_kutta<..., default_error_checker, default_step_adjuster>, FSAL variant):
 *   err = max_i |xerr_i| / (abs + rel (|x_i| + dt |dxdt_i|));
 *   err > 1: reject, dt *= max(0.9 err^(-1/3), 1/5)             (error order 4 -> exponent -1/(4-1));
 *   else accept, and if err < 0.5: dt *= 0.9 max(5^-5, err)^(-1/5)   (stepper order 5; growth capped at 5x);
 *   the iterator steps while t_cur < t_k - eps and then interpolates; initial dt = backTrajDt.
 * PARITY UNPINNED at this boundary, like OSQP: no golden vector of the reference exists (it ships no tests and its
 * default build has USE_ODEINT off, CMakeLists.txt:20). */
typedef struct {
	int nz;
	double t, t_old, dt;
	double z[OR_MAX_NX + OR_MAX_NX * OR_MAX_NX], z_old[OR_MAX_NX + OR_MAX_NX * OR_MAX_NX];
	double k[7][OR_MAX_NX + OR_MAX_NX * OR_MAX_NX]; /* stage derivatives of the LAST accepted step; k[6] = f(z) (FSAL) */
	double dz[OR_MAX_NX + OR_MAX_NX * OR_MAX_NX];   /* derivative at z */
	int failed; /* odeint throws (500 failed attempts of one step; and a NaN error estimate never recovers): the
	             * reference has no handler, so the process ends.  Here: every later sample is NaN, the rows are
	             * non-finite and filter() fails like a failed solve (rc -1, backup controller). */
} dopri5_t;

static void dopri5_init(dopri5_t *d, const or_model *m, const or_options *o, const double *z0, int nz, double dt)
{
	memset(d, 0, sizeof(*d));
	d->nz = nz;
	d->t = d->t_old = 0.0;
	d->dt = dt;
	memcpy(d->z, z0, sizeof(double) * nz);
	memcpy(d->z_old, z0, sizeof(double) * nz);
	ode_rhs(m, o, d->z, d->dz, 0, 0.0);
}

/* one accepted step (retries inside); sets d->failed instead when odeint would have thrown */
static void dopri5_step(dopri5_t *d, const or_model *m, const or_options *o)
{
	static const double a21 = 1.0 / 5, a31 = 3.0 / 40, a32 = 9.0 / 40, a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9,
	                    a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729,
	                    a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176,
	                    a65 = -5103.0 / 18656, c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192,
	                    c5 = -2187.0 / 6784, c6 = 11.0 / 84;
	static const double dc1 = 35.0 / 384 - 5179.0 / 57600, dc3 = 500.0 / 1113 - 7571.0 / 16695,
	                    dc4 = 125.0 / 192 - 393.0 / 640, dc5 = -2187.0 / 6784 - (-92097.0 / 339200),
	                    dc6 = 11.0 / 84 - 187.0 / 2100, dc7 = -1.0 / 40;
	const int n = d->nz;
	double zt[OR_MAX_NX + OR_MAX_NX * OR_MAX_NX], zn[OR_MAX_NX + OR_MAX_NX * OR_MAX_NX], k[7][OR_MAX_NX + OR_MAX_NX * OR_MAX_NX];
	for (int tries = 0; tries < 500; tries++) {
		const double h = d->dt;
		memcpy(k[0], d->dz, sizeof(double) * n);
		for (int i = 0; i < n; i++) zt[i] = d->z[i] + h * a21 * k[0][i];
		ode_rhs(m, o, zt, k[1], 0, 0.0);
		for (int i = 0; i < n; i++) zt[i] = d->z[i] + h * (a31 * k[0][i] + a32 * k[1][i]);
		ode_rhs(m, o, zt, k[2], 0, 0.0);
		for (int i = 0; i < n; i++) zt[i] = d->z[i] + h * (a41 * k[0][i] + a42 * k[1][i] + a43 * k[2][i]);
		ode_rhs(m, o, zt, k[3], 0, 0.0);
		for (int i = 0; i < n; i++) zt[i] = d->z[i] + h * (a51 * k[0][i] + a52 * k[1][i] + a53 * k[2][i] + a54 * k[3][i]);
		ode_rhs(m, o, zt, k[4], 0, 0.0);
		for (int i = 0; i < n; i++)
			zt[i] = d->z[i] + h * (a61 * k[0][i] + a62 * k[1][i] + a63 * k[2][i] + a64 * k[3][i] + a65 * k[4][i]);
		ode_rhs(m, o, zt, k[5], 0, 0.0);
		for (int i = 0; i < n; i++)
			zn[i] = d->z[i] + h * (c1 * k[0][i] + c3 * k[2][i] + c4 * k[3][i] + c5 * k[4][i] + c6 * k[5][i]);
		ode_rhs(m, o, zn, k[6], 0, 0.0);
		double err = 0.0;
		for (int i = 0; i < n; i++) {
			const double xe = h * (dc1 * k[0][i] + dc3 * k[2][i] + dc4 * k[3][i] + dc5 * k[4][i] + dc6 * k[5][i] + dc7 * k[6][i]);
			const double e = fabs(xe) / (o->backTrajAbsTol + o->backTrajRelTol * (fabs(d->z[i]) + fabs(h) * fabs(d->dz[i])));
			if (e > err) err = e;
			if (e != e) d->failed = 1; /* a NaN error estimate: max() drops it and the step would be accepted */
		}
		if (d->failed) return;
		if (err > 1.0) {
			double fac = 0.9 * pow(err, -1.0 / 3.0);
			if (fac < 0.2) fac = 0.2;
			d->dt = h * fac;
			continue;
		}
		memcpy(d->z_old, d->z, sizeof(double) * n);
		memcpy(d->z, zn, sizeof(double) * n);
		memcpy(d->k, k, sizeof(k));
		memcpy(d->dz, k[6], sizeof(double) * n);
		d->t_old = d->t;
		d->t = d->t + h;
		if (err < 0.5) {
			const double floor5 = pow(5.0, -5.0);
			const double e = err > floor5 ? err : floor5;
			d->dt = h * 0.9 * pow(e, -1.0 / 5.0);
		}
		return;
	}
	d->failed = 1; /* odeint's failed_step_checker: 500 consecutive rejections */
}

/* state at time ts >= the last sample: step while t < ts - eps, then the continuous extension on [t_old, t] */
static void dopri5_sample(dopri5_t *d, const or_model *m, const or_options *o, double ts, double *out)
{
	while (!d->failed && ts - d->t > 2.220446049250313e-16) dopri5_step(d, m, o);
	const int n = d->nz;
	if (d->failed) {
		for (int i = 0; i < n; i++) out[i] = NAN;
		return;
	}
	const double b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84;
	const double h = d->t - d->t_old;
	if (!(h > 0.0)) { /* before any step: ts == 0 */
		memcpy(out, d->z, sizeof(double) * n);
		return;
	}
	const double th = (ts - d->t_old) / h;
	const double X1 = 5.0 * (2558722523.0 - 31403016.0 * th) / 11282082432.0;
	const double X3 = 100.0 * (882725551.0 - 15701508.0 * th) / 32700410799.0;
	const double X4 = 25.0 * (443332067.0 - 31403016.0 * th) / 1880347072.0;
	const double X5 = 32805.0 * (23143187.0 - 3489224.0 * th) / 199316789632.0;
	const double X6 = 55.0 * (29972135.0 - 7076736.0 * th) / 822651844.0;
	const double X7 = 10.0 * (7414447.0 - 829305.0 * th) / 29380423.0;
	const double thm1 = th - 1.0, thsq = th * th;
	const double A = thsq * (3.0 - 2.0 * th), B = thsq * thm1, C = thsq * thm1 * thm1, D = th * thm1 * thm1;
	const double bt1 = A * b1 - C * X1 + D, bt3 = A * b3 + C * X3, bt4 = A * b4 - C * X4, bt5 = A * b5 + C * X5,
	             bt6 = A * b6 - C * X6, bt7 = B + C * X7;
	for (int i = 0; i < n; i++)
		out[i] = d->z_old[i] + h * (bt1 * d->k[0][i] + bt3 * d->k[2][i] + bt4 * d->k[3][i] + bt5 * d->k[4][i] +
		                                           bt6 * d->k[5][i] + bt7 * d->k[6][i]);
}

typedef struct {
	int npBT, nz;
	double dt;
	double *t;     /* [npBT] accumulated sample times */
	double *z;     /* [npBT][nz] */
	double *hFull; /* [npBT][npSS] */
	double *DhFull;/* [npBT][npSS*nx] */
	double *hMin;  /* [npBT] */
	int *order;    /* [npBT] */
} traj_t;

static void traj_free(traj_t *T)
{
	free(T->t); free(T->z); free(T->hFull); free(T->DhFull); free(T->hMin); free(T->order);
}

/* src/asif_implicit.cpp:417-425,461-484 == src/asif_implicit_tb.cpp:421-430,464-488:
 * forward Euler on [x; vec Q], Q(0)=I, time ignored by the rhs; safety set sampled at every point */
static void integrate(const or_model *m, const or_options *o, int variant, int npBTSS, const double *x, traj_t *T)
{
	const int nx = m->nx, np = m->npSS, nz = nx + nx * nx;
	T->npBT = traj_len(variant, o, npBTSS, &T->dt);
	T->nz = nz;
	const int n = T->npBT;
	T->t = (double *)calloc(n, sizeof(double));
	T->z = (double *)calloc((size_t)n * nz, sizeof(double));
	T->hFull = (double *)calloc((size_t)n * np, sizeof(double));
	T->DhFull = (double *)calloc((size_t)n * np * nx, sizeof(double));
	T->hMin = (double *)calloc(n, sizeof(double));
	T->order = (int *)calloc(n, sizeof(int));
	double *z0 = T->z;
	for (int i = 0; i < nx; i++) z0[i] = x[i];
	for (int i = nx; i < nz; i += nx + 1) z0[i] = 1.0;
	double zdot[OR_MAX_NX + OR_MAX_NX * OR_MAX_NX];
	zoh_t zoh;
	memset(&zoh, 0, sizeof(zoh));
	zoh.dt = T->dt;
	const int rb = variant == OR_VARIANT_IMPLICIT_RB;
	const int dopri = o->integrator == 1 && (variant == OR_VARIANT_IMPLICIT || variant == OR_VARIANT_IMPLICIT_TB); /* src/asif_implicit_tb.cpp:431-463 */
	dopri5_t ds;
	if (dopri) dopri5_init(&ds, m, o, z0, nz, T->dt);
	for (int i = 0; i < n; i++) {
		double *zi = T->z + (size_t)i * nz;
		if (i > 0 && dopri) {
			/* src/asif_implicit.cpp:447-460: n_step_iterator over a dense-output dopri5, sample i at t = i*backTrajDt */
			T->t[i] = T->dt * (double)i;
			dopri5_sample(&ds, m, o, T->t[i], zi);
		} else if (i > 0) {
			const double *zp = zi - nz;
			T->t[i] = T->t[i - 1] + T->dt;
			/* src/asif_implicit_robust.cpp:567: the rhs at sample i-1 is stamped t = i*backTrajDt */
			ode_rhs(m, o, zp, zdot, rb ? &zoh : 0, (double)(unsigned)i * T->dt);
			for (int k = 0; k < nz; k++) zi[k] = zdot[k] * T->dt;
			for (int k = 0; k < nz; k++) zi[k] = zi[k] + zp[k];
		}
		double *hi = T->hFull + (size_t)i * np;
		m->safety(o, zi, hi, T->DhFull + (size_t)i * np * nx);
		double mn = hi[0];
		for (int k = 1; k < np; k++)
			if (hi[k] < mn) mn = hi[k];
		T->hMin[i] = mn;
		T->order[i] = i;
	}
}

/* Indexes of the smallest hMin first; ties -> lowest index first.  The reference uses std::sort
 * (src/asif_implicit.cpp:487), whose order among equal keys is implementation-defined
 * (SURVEY App. B 3); "lowest index wins" is this build's fixed rule, on device too.
 * Only the leading npBTSS (<=10) entries are ever consumed, so a partial selection suffices. */
static void sort_by_hmin(traj_t *T, int count)
{
	const int want = count < 16 ? count : 16;
	for (int k = 0; k < want; k++) {
		int best = k;
		for (int i = k + 1; i < count; i++) {
			const int a = T->order[i], c = T->order[best];
			if (T->hMin[a] < T->hMin[c] || (T->hMin[a] == T->hMin[c] && a < c)) best = i;
		}
		const int tmp = T->order[k];
		T->order[k] = T->order[best];
		T->order[best] = tmp;
	}
}

/* ------------------------------------------------- ASIFimplicitRB extras */
static __thread double g_rb_dh_index[OR_MAX_NX];
static __thread double g_rb_lfh_diff, g_rb_lgh_diff[OR_MAX_NU];

void or_rb_last_learning(double *Dh_index, double *Lfh_diff, double *Lgh_diff)
{
	if (Dh_index) memcpy(Dh_index, g_rb_dh_index, sizeof(g_rb_dh_index));
	if (Lfh_diff) *Lfh_diff = g_rb_lfh_diff;
	if (Lgh_diff) memcpy(Lgh_diff, g_rb_lgh_diff, sizeof(g_rb_lgh_diff));
}

/* src/asif_implicit_robust.cpp:635-647: x_int[i] = interval(x[i]-x_unc[i], x[i]+x_unc[i]);
 * safetySet_int_(x_int, h_int, Dh_int); h = h_int.convert().left() */
static int rb_safety_lo(const or_model *m, const or_options *o, const double *x, double *hlo)
{
	or_af_ctx cx = {0, 0};
	or_af xi[OR_MAX_NX], h[OR_MAX_NPSS];
	for (int i = 0; i < m->nx; i++) or_af_interval(&cx, &xi[i], x[i] - o->x_unc[i], x[i] + o->x_unc[i]);
	m->safety_af(o, &cx, xi, h);
	for (int i = 0; i < m->npSS; i++) {
		double hi;
		or_af_convert(&h[i], &hlo[i], &hi);
	}
	return cx.overflow ? -100 : 0;
}

int or_rb_safety_lo(int model, const or_options *o, const double *x, double *hlo)
{
	const or_model *m = or_model_get(model);
	if (!m || !m->safety_af) return -100;
	return rb_safety_lo(m, o, x, hlo);
}

/* include/asif_learning_utils.h:34-76 (driftNN) == :78-121 (actNN): two ReLU layers and a linear one */
static void rb_mlp(const double *w1, const double *b1, const double *w2, const double *b2, const double *w3,
                   const double *b3, int din, int dh1, int dh2, int dout, const double *in, double *out)
{
	double o1[dh1 > 0 ? dh1 : 1], o2[dh2 > 0 ? dh2 : 1], o3[dout > 0 ? dout : 1];
	or_matvec(w1, dh1, din, in, o1);
	for (int i = 0; i < dh1; i++) o1[i] = fmax(0., o1[i] + b1[i]);
	or_matvec(w2, dh2, dh1, o1, o2);
	for (int i = 0; i < dh2; i++) o2[i] = fmax(0., o2[i] + b2[i]);
	or_matvec(w3, dout, dh2, o2, o3);
	for (int i = 0; i < dout; i++) out[i] = o3[i] + b3[i];
}

/* include/asif_learning_utils.h:123-155 (update_weights): inputs [x; Dh[0..nx)] zero padded to d_*_in,
 * Lfh[0] += drift[0], Lgh[i] += act[i] for i < nu -- the first nu entries of the column-major Lgh, i.e. row i
 * of input 0 */
static int rb_update_weights(const or_learning *L, const double *x, int nx, const double *Dh, double *Lfh, double *Lgh,
                             int nu)
{
	if (!L || (int)L->d_drift_in < 2 * nx || (int)L->d_act_in < 2 * nx || L->d_drift_out < 1 || (int)L->d_act_out < nu)
		return -100; /* the reference would run off its stack arrays */
	double din[L->d_drift_in], ain[L->d_act_in], dout[L->d_drift_out], aout[L->d_act_out];
	memset(din, 0, sizeof(din));
	memset(ain, 0, sizeof(ain));
	for (int i = 0; i < nx; i++) {
		din[i] = x[i];
		din[i + nx] = Dh[i];
		ain[i] = x[i];
		ain[i + nx] = Dh[i];
	}
	rb_mlp(L->w_1_drift, L->b_1_drift, L->w_2_drift, L->b_2_drift, L->w_3_drift, L->b_3_drift, L->d_drift_in,
	       L->d_drift_hidden, L->d_drift_hidden_2, L->d_drift_out, din, dout);
	rb_mlp(L->w_1_act, L->b_1_act, L->w_2_act, L->b_2_act, L->w_3_act, L->b_3_act, L->d_act_in, L->d_act_hidden,
	       L->d_act_hidden_2, L->d_act_out, ain, aout);
	g_rb_lfh_diff = dout[0];
	Lfh[0] += dout[0];
	for (int i = 0; i < nu; i++) {
		Lgh[i] += aout[i];
		g_rb_lgh_diff[i] = aout[i];
	}
	return 0;
}

/* ---------------------------------------------------------------- implicit */
/* src/asif_implicit.cpp:403-651; rb: src/asif_implicit_robust.cpp:481-778 (same rows; safe-row margins
 * replaced by interval lower ends, optional learned residual on the first row's Lie derivatives; the
 * interval Lie derivatives Lfh_int / Lgh_int of :698-709 are computed and never used there) */
static int assemble_implicit(const or_model *m, const or_options *o, const double *x, double *A, double *b, int rb)
{
	const int nx = m->nx, nu = m->nu, np = m->npSS, nb = m->npBS, npBTSS = m->npBTSS;
	const int nTC = npBTSS * np + nb, nv = nu + 2;
	double f[OR_MAX_NX], g[OR_MAX_NX * OR_MAX_NU];
	m->dynamics(o, x, f, g);
	traj_t T;
	integrate(m, o, rb ? OR_VARIANT_IMPLICIT_RB : OR_VARIANT_IMPLICIT, npBTSS, x, &T);
	sort_by_hmin(&T, T.npBT);
	/* n_debug / use_learning / learning_data_ exist in BOTH classes (include/asif_implicit.h:23,33,125,
	 * src/asif_implicit.cpp:219-224,500-515,533-537,585-588 == src/asif_implicit_robust.cpp:298-303,590-605,
	 * 627-631,713-715).  initialize(): n_debug outside (-1, npBT-1) is reset to -1 */
	const int learn = o->use_learning != 0;
	const int n_debug = ((rb || learn) && o->n_debug > -1 && o->n_debug < T.npBT - 1) ? o->n_debug : -1;
	int rc = 1;

	double h[64] = {0.0}, Dh[64 * OR_MAX_NX] = {0.0};
	double DhSSDx[OR_MAX_NPSS * OR_MAX_NX], DhBS[OR_MAX_NX], DhBSDx[OR_MAX_NX];
	g_last_ncrit = npBTSS;
	for (int idx = 0; idx < npBTSS; idx++) {
		const int cur = T.order[idx];
		g_last_crit[idx] = cur;
		memcpy(&h[idx * np], &T.hFull[(size_t)cur * np], np * sizeof(double));
		const double *DhSS = &T.DhFull[(size_t)cur * np * nx];
		const double *Q = T.z + (size_t)cur * T.nz + nx;
		or_matmul(DhSS, np, nx, Q, nx, DhSSDx);
		for (int i = 0; i < np; i++)
			for (int j = 0; j < nx; j++) Dh[(idx * np + i) + j * nTC] = DhSSDx[i + j * np];
		/* :624-632: Dh_index_ = the np x nx product of the most critical sample, column-major */
		if ((rb || learn) && idx == 0 && n_debug == -1)
			for (int i = 0; i < nx; i++) g_rb_dh_index[i] = DhSSDx[i];
		/* :635-647 */
		if (rb && rb_safety_lo(m, o, T.z + (size_t)cur * T.nz, &h[idx * np])) rc = -100;
	}
	if (n_debug != -1) { /* :590-605 */
		double DhDbg[OR_MAX_NPSS * OR_MAX_NX];
		or_matmul(&T.DhFull[(size_t)n_debug * np * nx], np, nx, T.z + (size_t)n_debug * T.nz + nx, nx, DhDbg);
		for (int i = 0; i < nx; i++) g_rb_dh_index[i] = DhDbg[i];
	}
	const double *zend = T.z + (size_t)(T.npBT - 1) * T.nz;
	m->backup(o, zend, &h[npBTSS * np], DhBS, 0);
	or_matmul(DhBS, nb, nx, zend + nx, nx, DhBSDx);
	for (int i = 0; i < nb; i++)
		for (int j = 0; j < nx; j++) Dh[(npBTSS * np + i) + j * nTC] = DhBSDx[i + j * nb];

	double Lfh[64], Lgh[64 * OR_MAX_NU];
	or_matvec(Dh, nTC, nx, f, Lfh);
	or_matmul(Dh, nTC, nx, g, nu, Lgh);
	if (learn) { /* :713-715 */
		if (rb_update_weights(o->learning, x, nx, g_rb_dh_index, Lfh, Lgh, nu)) rc = -100;
	}
	/* :591-611 */
	for (int i = 0; i < nTC * nv; i++) A[i] = 0.0;
	for (int i = 0; i < nTC; i++)
		for (int j = 0; j < nu; j++) A[i + j * nTC] = Lgh[i + j * nTC];
	for (int i = 0; i < npBTSS * np; i++) A[i + nu * nTC] = h[i];
	for (int i = npBTSS * np; i < nTC; i++) A[i + (nu + 1) * nTC] = h[i];
	for (int i = 0; i < nTC; i++) b[i] = -Lfh[i];
	traj_free(&T);
	return rc;
}

/* ---------------------------------------------------------------------- TB */
/* src/asif_implicit_tb.cpp:716-733 */
static void assemble_tb_trivial(const or_options *o, int nTC, int nv, double *A, double *b, double *diag)
{
	for (int i = 0; i < nTC * nv; i++) A[i] = 0.0;
	for (int i = 0; i < nTC; i++) b[i] = -or_no_bound(o->inf);
	if (diag) {
		diag[0] = 0.0; /* TTS_ */
		diag[1] = 1.0; /* BTorthoBS_ */
		diag[2] = 0.0;
	}
}

/* src/asif_implicit_tb.cpp:407-714; returns 1, or -1 when the backup set is never hit (:529-536) */
static int assemble_tb(const or_model *m, const or_options *o, const double *x, double *A, double *b, double *diag)
{
	const int nx = m->nx, nu = m->nu, np = m->npSS, npBTSS = 4;
	const int nTC = npBTSS * np + 2;
	double f[OR_MAX_NX], g[OR_MAX_NX * OR_MAX_NU];
	m->dynamics(o, x, f, g);
	traj_t T;
	integrate(m, o, OR_VARIANT_IMPLICIT_TB, npBTSS, x, &T);

	/* time-to-safety scan, :490-536 -- effectively "first hit wins" */
	double hBSnm1 = -1.0, hBS[1] = {0}, DhBS[OR_MAX_NX], DDhBS[OR_MAX_NX * OR_MAX_NX];
	int BSHit = 0, idxHit = 0;
	double cosTilde[1] = {0}, fClBS[OR_MAX_NX], DfClBS[OR_MAX_NX * OR_MAX_NX];
	double den1 = 0, den2 = 0, den = 0, ortho = 0;
	const double *btX = 0;
	for (int i = 1; i < T.npBT; i++) {
		if (hBSnm1 < 0.0) {
			m->backup(o, T.z + (size_t)i * T.nz, hBS, DhBS, DDhBS);
			if (hBS[0] >= 0.0) {
				idxHit = i;
				BSHit = 1;
				btX = T.z + (size_t)i * T.nz;
				backup_cl(m, o, btX, fClBS, DfClBS, 0, 0.0);
				or_matmul(DhBS, 1, nx, fClBS, 1, cosTilde);
				den1 = or_vecnorm(DhBS, nx);
				den2 = or_vecnorm(fClBS, nx);
				den = den1 * den2;
				ortho = cosTilde[0] / den;
				if (ortho > 2.0 * o->backTrajMinOrtho) break;
			}
		}
		hBSnm1 = hBS[0];
	}
	if (!BSHit) {
		if (diag) {
			diag[1] = 0;
		}
		traj_free(&T);
		return -1;
	}
	sort_by_hmin(&T, idxHit + 1);

	double h[32] = {0.0}, Dh[32 * OR_MAX_NX] = {0.0};
	double DhSSDx[OR_MAX_NPSS * OR_MAX_NX], DhBSDx[OR_MAX_NX];
	g_last_ncrit = 0;
	for (int idx = 0; idx < npBTSS; idx++) {
		if (idx > idxHit) { /* fewer trajectory points than rows: inert padding, :556-566 */
			for (int i = 0; i < np; i++) {
				h[idx * np + i] = 1.0;
				for (int j = 0; j < nx; j++) Dh[(idx * np + i) + j * nTC] = 0.0;
			}
		} else {
			const int cur = T.order[idx];
			g_last_crit[g_last_ncrit++] = cur;
			memcpy(&h[idx * np], &T.hFull[(size_t)cur * np], np * sizeof(double));
			const double *DhSS = &T.DhFull[(size_t)cur * np * nx];
			or_matmul(DhSS, np, nx, T.z + (size_t)cur * T.nz + nx, nx, DhSSDx);
			for (int i = 0; i < np; i++)
				for (int j = 0; j < nx; j++) Dh[(idx * np + i) + j * nTC] = DhSSDx[i + j * np];
		}
	}
	/* time-to-safety row, :588-599 */
	const double TTS = T.t[idxHit];
	const double hReach = o->backTrajHorizon - T.t[idxHit];
	const double *btDX = btX + nx;
	or_matmul(DhBS, 1, nx, btDX, nx, DhBSDx);
	h[npBTSS * np] = hReach;
	for (int i = 0; i < nx; i++) Dh[(npBTSS * np) + i * nTC] = DhBSDx[i] / cosTilde[0];
	/* orthogonality row and its gradient, :601-641 */
	const double denSquared = den * den;
	h[npBTSS * np + 1] = ortho - o->backTrajMinOrtho;
	double DxHit[OR_MAX_NX * OR_MAX_NX];
	or_matmul(fClBS, nx, 1, DhBSDx, nx, DxHit);
	for (int i = 0; i < nx * nx; i++) DxHit[i] = btDX[i] - DxHit[i];
	double Dnum[OR_MAX_NX] = {0.0}, Dden1[OR_MAX_NX] = {0.0}, Dden2[OR_MAX_NX] = {0.0}, Dden[OR_MAX_NX];
	for (int i = 0; i < nx; i++)
		for (int k = 0; k < nx; k++) {
			double temp1 = 0.0, temp2 = 0.0;
			for (int l = 0; l < nx; l++) {
				temp1 += DDhBS[k + l * nx] * DxHit[l + i * nx];
				temp2 += DfClBS[k + l * nx] * DxHit[l + i * nx];
			}
			const double temp3 = DhBS[k] * temp2;
			const double temp4 = temp1 * fClBS[k];
			Dden1[i] += temp3;
			Dden2[i] += temp4;
			Dnum[i] += temp3 + temp4;
		}
	for (int i = 0; i < nx; i++) Dden[i] = den2 * Dden1[i] / den1 + den1 * Dden2[i] / den2;
	for (int i = 0; i < nx; i++) {
		DhBSDx[i] = (Dnum[i] * den - cosTilde[0] * Dden[i]) / denSquared;
		Dh[(npBTSS * np + 1) + i * nTC] = DhBSDx[i];
	}
	double Lfh[32], Lgh[32 * OR_MAX_NU];
	or_matvec(Dh, nTC, nx, f, Lfh);
	or_matmul(Dh, nTC, nx, g, nu, Lgh);
	/* :655-674.  Column nu of the two extra rows is never written by the reference after the
	 * constructor / trivial fill zeroed it, so it is 0. */
	for (int i = 0; i < nTC * (nu + 1); i++) A[i] = 0.0;
	for (int i = 0; i < nTC; i++)
		for (int j = 0; j < nu; j++) A[i + j * nTC] = Lgh[i + j * nTC];
	for (int i = 0; i < npBTSS * np; i++) A[i + nu * nTC] = h[i];
	for (int i = 0; i < nTC; i++) b[i] = -Lfh[i];
	b[npBTSS * np] -= o->relaxTTS * h[npBTSS * np];
	b[npBTSS * np + 1] -= o->relaxMinOrtho * (h[npBTSS * np + 1]);
	if (diag) {
		diag[0] = TTS;
		diag[1] = ortho;
		diag[2] = (double)idxHit;
	}
	traj_free(&T);
	return 1;
}

/* ------------------------------------------------------------------ robust */
/* src/asif_robust.cpp:103-133 (fixed structure) + :275-367 (interval rows), npSSmax == npSS */
static int assemble_robust(const or_model *m, const or_options *o, const double *x, double *A, double *b)
{
	const int nx = m->nx, nu = m->nu, N = o->nHalfPlanes;
	const int nv = nu + 1 + N * 2 * (nu + 1), nc = N * (nu + 2);
	for (int i = 0; i < nc * nv; i++) A[i] = 0.0;
	for (int i = 0; i < nc; i++) b[i] = 0.0;
	int iCol = nu + 1;
	for (int iRow = 0; iRow < nc; iRow += nu + 2) {
		for (int i = 0; i < nu; i++)
			for (int j = 0; j < nu; j++) A[(iRow + 1 + i) + j * nc] = -1.0; /* full -1 block, exact for nu==1 only (App. B 8) */
		for (int i = 0; i < nu + 1; i++) {
			A[(iRow + 1 + i) + (iCol + i) * nc] = 1.0;
			A[(iRow + 1 + i) + (iCol + nu + 1 + i) * nc] = -1.0;
		}
		b[iRow + nu + 1] = 1.0;
		iCol += 2 * (nu + 1);
	}
	or_af_ctx cx = {0, 0};
	or_af xI[OR_MAX_NX], f[OR_MAX_NX], g[OR_MAX_NX * OR_MAX_NU];
	for (int i = 0; i < nx; i++) or_af_interval(&cx, &xI[i], x[i], x[i]);
	double h[OR_MAX_NPSS], Dh[OR_MAX_NPSS * OR_MAX_NX];
	m->safety(o, x, h, Dh);
	for (int i = 0; i < nx; i++) or_af_const(&f[i], 0);
	for (int i = 0; i < nx * nu; i++) or_af_const(&g[i], 0);
	m->dynamics_af(o, &cx, xI, f, g);
	or_af DhI[OR_MAX_NPSS * OR_MAX_NX], Lfh[OR_MAX_NPSS], Lgh[OR_MAX_NPSS * OR_MAX_NU], t;
	for (int i = 0; i < N * nx; i++) or_af_interval(&cx, &DhI[i], Dh[i], Dh[i]);
	/* include/asif_utils.h:46-62 instantiated on AAF: Ab[i] = 0.0; Ab[i] = Ab[i] + A*b */
	for (int i = 0; i < N; i++) {
		or_af_const(&Lfh[i], 0.0);
		for (int k = 0; k < nx; k++) {
			or_af_mul(&cx, &DhI[i + k * N], &f[k], &t);
			or_af_add(&Lfh[i], &t, &Lfh[i]);
		}
	}
	for (int i = 0; i < N; i++)
		for (int j = 0; j < nu; j++) {
			or_af *r = &Lgh[i + j * N];
			or_af_const(r, 0.0);
			for (int k = 0; k < nx; k++) {
				or_af_mul(&cx, &DhI[i + k * N], &g[k + j * nx], &t);
				or_af_add(r, &t, r);
			}
		}
	iCol = nu + 1;
	int s = 0;
	for (int iRow = 0; iRow < nc; iRow += nu + 2) {
		double lo, hi;
		A[iRow + nu * nc] = h[s];
		for (int j = 0; j < nu; j++) {
			or_af_convert(&Lgh[s + j * N], &lo, &hi);
			A[iRow + (iCol + j) * nc] = lo;
			A[iRow + (iCol + (nu + 1) + j) * nc] = -hi;
		}
		or_af_convert(&Lfh[s], &lo, &hi);
		A[iRow + (iCol + nu) * nc] = lo;
		A[iRow + (iCol + (nu + 1) + nu) * nc] = -hi;
		iCol += 2 * (nu + 1);
		s++;
	}
	return cx.overflow ? -100 : 1;
}

/* ---------------------------------------------------------------- dispatch */
int or_assemble(int model, int variant, const or_options *o, const double *x, double *A, double *b, double *diag)
{
	const or_model *m = or_model_get(model);
	or_dims d;
	if (!m || or_get_dims(model, variant, o, &d)) return -100;
	switch (variant) {
	case OR_VARIANT_EXPLICIT:
		return assemble_explicit(m, o, x, A, b);
	case OR_VARIANT_IMPLICIT:
		return assemble_implicit(m, o, x, A, b, 0);
	case OR_VARIANT_IMPLICIT_RB:
		return assemble_implicit(m, o, x, A, b, 1);
	case OR_VARIANT_IMPLICIT_TB: {
		/* src/asif_implicit_tb.cpp:278-290: inside the backup set -> trivial rows, filter() returns 2 */
		double hb[1], Dhb[OR_MAX_NX], DDhb[OR_MAX_NX * OR_MAX_NX];
		m->backup(o, x, hb, Dhb, DDhb);
		if (hb[0] >= 0) {
			assemble_tb_trivial(o, d.nc, d.nv, A, b, diag);
			return 2;
		}
		int r = assemble_tb(m, o, x, A, b, diag);
		return r == 1 ? 1 : -3;
	}
	case OR_VARIANT_ROBUST:
		return assemble_robust(m, o, x, A, b);
	}
	return -100;
}

/* Cost / bounds / equality flags each variant hands to QPWrapperAbstract::initialize + updateCost:
 *  explicit src/asif.cpp:84-98,314-325; implicit src/asif_implicit.cpp:237-254,653-664;
 *  TB src/asif_implicit_tb.cpp:198-210; robust src/asif_robust.cpp:89-101,140-148 */
void or_qp_static(int model, int variant, const or_options *o, const double *uDes,
                  double *Hd, double *c, double *lb, double *ub, uint8_t *be)
{
	or_dims d;
	if (or_get_dims(model, variant, o, &d)) return;
	const int nu = d.nu;
	for (int i = 0; i < d.nv; i++) { Hd[i] = 0; c[i] = 0; lb[i] = 0; ub[i] = 0; }
	for (int i = 0; i < d.nc; i++) be[i] = 0;
	for (int i = 0; i < nu; i++) {
		Hd[i] = 1.0;
		c[i] = -2.0 * uDes[i];
		lb[i] = o->lb[i];
		ub[i] = o->ub[i];
	}
	switch (variant) {
	case OR_VARIANT_EXPLICIT:
		Hd[nu] = o->relaxCost;
		c[nu] = -2.0 * o->relaxCost * o->relaxLb;
		lb[nu] = o->relaxLb;
		ub[nu] = o->relaxLb; /* src/asif.cpp:91: the relax variable is pinned */
		break;
	case OR_VARIANT_IMPLICIT:
	case OR_VARIANT_IMPLICIT_RB: /* src/asif_implicit_robust.cpp:305-334,780-791: identical */
		Hd[nu] = Hd[nu + 1] = o->relaxCost;
		c[nu] = -2.0 * o->relaxCost * o->relaxLb;
		c[nu + 1] = -2.0 * o->relaxCost * o->relaxReachLb;
		lb[nu] = o->relaxLb;
		lb[nu + 1] = o->relaxReachLb;
		ub[nu] = ub[nu + 1] = or_no_bound(o->inf);
		break;
	case OR_VARIANT_IMPLICIT_TB:
		Hd[nu] = o->relaxCost;
		c[nu] = -2.0 * o->relaxCost * o->relaxLb;
		lb[nu] = o->relaxLb;
		ub[nu] = or_no_bound(o->inf);
		break;
	case OR_VARIANT_ROBUST:
		Hd[nu] = o->relaxCost;
		c[nu] = -2.0 * o->relaxCost * o->relaxLb;
		lb[nu] = o->relaxLb;
		ub[nu] = or_no_bound(o->inf);
		for (int i = nu + 1; i < d.nv; i++) { lb[i] = 0.0; ub[i] = or_no_bound(o->inf); }
		for (int i = 0; i < d.nc; i++) be[i] = (i % (nu + 2)) != 0;
		break;
	}
}
