/* or_internal.h -- shared declarations inside oracle/.  TEST INFRASTRUCTURE (see or_oracle.h). */
#ifndef OR_INTERNAL_H
#define OR_INTERNAL_H

#include "or_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Model callbacks with the reference's std::function signatures (include/asif.h:22-27,
 * include/asif_implicit.h:48-62, include/asif_implicit_tb.h:46-60, include/asif_robust.h:24-29);
 * ud carries or_options for models that need data. */
typedef struct {
	int nx, nu, npSS, npBS;
	void (*safety)(const void *ud, const double *x, double *h, double *Dh);
	void (*backup)(const void *ud, const double *x, double *h, double *Dh, double *DDh);
	void (*dynamics)(const void *ud, const double *x, double *f, double *g);
	void (*gradients)(const void *ud, const double *x, double *Df, double *Dg);
	void (*controller)(const void *ud, const double *x, double *u, double *Du);
	void (*dynamics_af)(const void *ud, or_af_ctx *cx, const or_af *x, or_af *f, or_af *g);
	/* the safety set on affine forms (ASIFimplicitRB's safetySet_int, include/asif_implicit_robust.h:47-49);
	 * only h is consumed by the reference (src/asif_implicit_robust.cpp:645-647), Dh_int feeds dead code */
	void (*safety_af)(const void *ud, or_af_ctx *cx, const or_af *x, or_af *h);
	int npBTSS; /* critical samples kept by the implicit variant (constructor argument of the example) */
} or_model;

const or_model *or_model_get(int id);

/* Options::inf as the solver reads it: OSQP takes any magnitude from OSQP_INFTY = 1e30 on for "no bound", so an
 * infinite or huge Options::inf (the TB class's inert rows b = -inf, src/asif_implicit_tb.cpp:727, the relaxation upper
 * bounds) is the same problem to the reference as 1e30 -- the value the exact solver, which has no such notion, is handed */
static inline double or_no_bound(double inf) { return inf > 1e30 ? 1e30 : inf; }

void or_matvec(const double *A, int nl, int ncol, const double *b, double *Ab);
void or_matmul(const double *A, int nlA, int ncA, const double *B, int ncB, double *AB);
double or_vecnorm(const double *v, int len);

#endif
