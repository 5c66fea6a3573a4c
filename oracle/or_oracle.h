/*
 * or_oracle.h -- CPU oracle for the batched CBF-QP safety-filter hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call anything in oracle/.
 * The shipped path (asif_amd/csrc -> libasif_hip.so) never links or calls it.
 *
 * What it is: a plain-C99 restatement of the reference's per-control-step
 * arithmetic (DrewSingletary/asif, cited file:line per function, paths relative
 * to the reference root), FP64, column-major, same loop/evaluation order,
 * built with -ffp-contract=off so no FMA contraction changes a rounding.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - affine arithmetic (or_affine.c): PINNED against the reference's own
 *     libaffa sources compiled unmodified into oracle/_ref (tests/golden/affa_*.json).
 *   - explicit / implicit assembly: pinned at the two known-answer points that
 *     SURVEY.md 8(c) recorded from the reference code (tests/golden/survey_known_answers.json).
 *   - TB / robust assembly rows beyond that, and everything at the OSQP boundary:
 *     PARITY UNPINNED.  src/asif*.cpp cannot be built here (needs <osqp.h>, absent
 *     from the image; no stand-in header is written), and OSQP itself is absent.
 *     The reference has no tests or golden vectors of its own.
 */
#ifndef OR_ORACLE_H
#define OR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- models */
enum or_model_id {
	OR_MODEL_DOUBLE_INTEGRATOR = 0,        /* examples/DoubleIntegrator.cpp:12-61          */
	OR_MODEL_INVERTED_PENDULUM = 1,        /* examples/InvertedPendulum_Implicit.cpp:13-80 */
	OR_MODEL_SEGWAY = 2,                   /* examples/segway_implicit_tb.cpp:13-212       */
	OR_MODEL_INVERTED_PENDULUM_ROBUST = 3, /* examples/InvertedPendulum_Robust.cpp:20-79   */
	OR_MODEL_INVERTED_PENDULUM_TB = 4,     /* examples/InvertedPendulum_ImplicitTB.cpp:14-99 */
	OR_MODEL_DOUBLE_INTEGRATOR_IMPLICIT = 5, /* examples/DoubleIntegrator_implicit.cpp:13-90 */
	OR_MODEL_PLANAR_TWO_INPUT = 6, /* NOT an example of the reference: a synthetic nx = 2, nu = 2 model for class ASIF,
	                                * so that the nu > 1 loops of src/asif.cpp:279-303,314-352 are exercised */
	OR_MODEL_DOUBLE_INTEGRATOR_TB = 7 /* examples/DoubleIntegrator_implicit_tb.cpp:13-103 */
};

enum or_variant_id {
	OR_VARIANT_EXPLICIT = 0,     /* src/asif.cpp             */
	OR_VARIANT_IMPLICIT = 1,     /* src/asif_implicit.cpp    */
	OR_VARIANT_IMPLICIT_TB = 2,  /* src/asif_implicit_tb.cpp */
	OR_VARIANT_ROBUST = 3,       /* src/asif_robust.cpp      */
	OR_VARIANT_IMPLICIT_RB = 5   /* src/asif_implicit_robust.cpp (4 = realizable, which has its own entry points) */
};

#define OR_MAX_NX 4
#define OR_MAX_NU 2
#define OR_MAX_NPSS 8

/* LearningData, include/asif_learning_utils.h:8-32 (the two output-side pointers Lfh_diff / Lgh_diff are
 * diagnostics the reference allocates per call, :146-147; not carried).  Weight matrices are dense
 * column-major [rows x cols] as matrixVectorMultiply reads them (include/asif_utils.h:46-62). */
typedef struct {
	uint32_t d_drift_in, d_act_in, d_drift_hidden, d_act_hidden, d_drift_hidden_2, d_act_hidden_2, d_drift_out,
	    d_act_out;
	const double *w_1_drift, *w_2_drift, *w_3_drift, *b_1_drift, *b_2_drift, *b_3_drift;
	const double *w_1_act, *w_2_act, *w_3_act, *b_1_act, *b_2_act, *b_3_act;
} or_learning;

/* Union of the four Options structs (include/asif.h:11-17, asif_implicit.h:20-34,
 * asif_implicit_tb.h:19-33, asif_robust.h:14-19) plus input bounds. */
typedef struct {
	double relaxCost;
	double relaxLb;          /* explicit/robust relaxLb; implicit/TB relaxSafeLb */
	double relaxReachLb;     /* implicit only */
	double relaxTTS;         /* TB only */
	double relaxMinOrtho;    /* TB only */
	double backTrajHorizon;
	double backTrajExtend;   /* TB only */
	double backTrajDt;
	double backTrajMinOrtho; /* TB only */
	double satSharpness;
	double inf;
	double lb[OR_MAX_NU];
	double ub[OR_MAX_NU];
	/* robust pendulum model data (examples/InvertedPendulum_Robust.cpp:35-38,51):
	 * the shipped SafetySetData is empty; half-planes are supplied here. */
	double pMin, pMax;
	int32_t nHalfPlanes;
	double halfPlanes[2 * OR_MAX_NPSS]; /* {a0,a1} pairs of 1 - a.x >= 0 */
	/* ASIFimplicitRB only (include/asif_implicit_robust.h:21-38) */
	double backContDt;          /* zero-order hold of the backup controller along the trajectory */
	double x_unc[OR_MAX_NX];    /* state uncertainty radius (Options::x_unc, nullptr -> zeros) */
	int32_t n_debug;            /* -1: Dh_index_ taken at the most critical sample */
	int32_t use_learning;
	const or_learning *learning;/* ASIFimplicitRB::learning_data_ (public member the caller fills) */
	/* class ASIF: constructor argument npSSmax (include/asif.h:28): <= 0 or >= npSS keeps every row, otherwise the
	 * npSSmax rows with the smallest h are kept per call (src/asif.cpp:250-268) */
	int32_t npSSmax;
	/* backup-trajectory integrator: 0 forward Euler (the reference's default build), 1 the USE_ODEINT build
	 * (dopri5 dense output, src/asif_implicit.cpp:427-460) with Options::backTrajAbsTol / backTrajRelTol */
	int32_t integrator;
	double backTrajAbsTol, backTrajRelTol;
} or_options;

/* Defaults per variant+model exactly as the named example's main() sets them. */
void or_default_options(int model, int variant, or_options *o);

typedef struct {
	int nx, nu, npSS, npBS, npBTSS;
	int nv, nc;      /* QP size at the QPWrapperAbstract boundary */
	int nrelax;      /* number of relax outputs of filter() */
	int npBT;        /* backup-trajectory samples (0 if none) */
} or_dims;

int or_get_dims(int model, int variant, const or_options *o, or_dims *d);

/* -------------------------------------------------------- affine arithmetic */
#define OR_AF_CAP 48
typedef struct {
	double c;                  /* central value */
	int n;                     /* number of noise symbols (zero coefficients kept) */
	int special;               /* 1 affine, 2 infinite, 4 nan (AAF_TYPE) */
	unsigned idx[OR_AF_CAP];   /* ascending symbol indexes */
	double v[OR_AF_CAP];
} or_af;
typedef struct { unsigned last; int overflow; } or_af_ctx; /* reference: global AAF::last */

void or_af_const(or_af *r, double v0);
void or_af_interval(or_af_ctx *cx, or_af *r, double lo, double hi);
void or_af_add(const or_af *a, const or_af *b, or_af *r);
void or_af_sub(const or_af *a, const or_af *b, or_af *r);
void or_af_neg(const or_af *a, or_af *r);
void or_af_scale(const or_af *a, double k, or_af *r);
void or_af_mul(or_af_ctx *cx, const or_af *a, const or_af *b, or_af *r);
void or_af_inv(or_af_ctx *cx, const or_af *a, or_af *r);
void or_af_div(or_af_ctx *cx, const or_af *a, const or_af *b, or_af *r);
void or_af_sin(or_af_ctx *cx, const or_af *a, or_af *r);
double or_af_rad(const or_af *a);
void or_af_convert(const or_af *a, double *lo, double *hi);
/* register-program runner, same encoding as oracle/ref_affa_shim.cpp */
typedef struct { int op, dst, a, b; double imm0, imm1; } or_af_instr;
int or_af_run(const or_af_instr *prog, int nprog, int nreg, int cap, double *center, int *n, double *lo,
              double *hi, unsigned *idx, double *coef);

/* ------------------------------------------------------------------ QP */
/* min x'Hx + c'x  s.t.  A x >= b (== b where be), lb <= x <= ub
 * (include/qpwrapper_abstract.h:11-15).  H diagonal (diagonalCost=true is the
 * only mode any config uses, src/qpwrapper_osqp.cpp:267-272). A col-major nc x nv. */
typedef struct {
	int nv, nc;
	const double *Hd, *c, *A, *b, *lb, *ub;
	const uint8_t *be; /* may be NULL */
} or_qp;

/* Exact optimum by active-set enumeration in long double; nv <= 3, Hd > 0.
 * Returns 1 feasible (x filled), 0 infeasible. */
int or_qp_exact_small(const or_qp *qp, double *x);

typedef struct {
	double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
	int scaling, adaptive_rho, adaptive_rho_interval, check_termination, max_iter;
	double adaptive_rho_tolerance;
	int reduced_kkt;  /* 0: quasi-definite (n+m) KKT like OSQP/QDLDL; 1: n x n Schur form (device layout) */
	int polish;       /* 0 off (reference), 1 on: validated active-set polish at every check */
	int scaling_pow2; /* 0 exact Ruiz factors (reference); 1 factors rounded to powers of two (device) */
} or_admm_settings;

void or_admm_default_settings(or_admm_settings *s); /* OSQP 0.6 defaults + max_iter=2000 (src/qpwrapper_osqp.cpp:68-69) */

typedef struct { int status; int iters; int rho_updates; double pri_res, dua_res; } or_admm_info;

/* OSQP status values (0.6.x, from memory; headers absent): */
#define OR_OSQP_SOLVED 1
#define OR_OSQP_SOLVED_INACCURATE 2
#define OR_OSQP_PRIMAL_INFEASIBLE_INACCURATE 3
#define OR_OSQP_DUAL_INFEASIBLE_INACCURATE 4
#define OR_OSQP_MAX_ITER_REACHED (-2)
#define OR_OSQP_PRIMAL_INFEASIBLE (-3)
#define OR_OSQP_DUAL_INFEASIBLE (-4)

/* OSQP-style ADMM on the translated problem (src/qpwrapper_osqp.cpp:263-376).
 * Returns QPWrapperOsqp::solve()'s value: 1 if SOLVED/SOLVED_INACCURATE else raw status (:225-238). */
int or_qp_admm(const or_qp *qp, const or_admm_settings *s, double *x, or_admm_info *info);

/* ------------------------------------------------------------ assembly */
/* updateConstraints() of the given variant: fills A (nc x nv col-major), b[nc].
 * diag (optional, >= 8 doubles): [0]=TTS_, [1]=BTorthoBS_, [2]=idxHit, [3..]=reserved.
 * Returns the variant's own code: 1 ok; TB: 2 = inside backup set (trivial rows),
 * -3 = backup set never reached (A,b untouched). */
int or_assemble(int model, int variant, const or_options *o, const double *x, double *A, double *b, double *diag);

/* Full QP at the boundary: Hd,c,lb,ub,be as initialize()/updateCost() build them. */
void or_qp_static(int model, int variant, const or_options *o, const double *uDes,
                  double *Hd, double *c, double *lb, double *ub, uint8_t *be);

/* class ASIF with caller-supplied Lie derivatives, filter(x, uDes, uAct, Lfh, Lgh[, relax]) (src/asif.cpp:130-165,
 * 287-292): rows use h of the (selected) safety functions and Lfh[i], Lgh[i + j*nc] as handed in. */
int or_filter_explicit_lie(int model, const or_options *o, const double *x, const double *uDes, const double *Lfh,
                           const double *Lgh, double *uAct, double *relax);
/* indexes of the safety functions kept by the last explicit or_assemble on this thread, in row order */
int or_last_kept_rows(int *idx, int cap);

/* critical sample indexes picked by the last or_assemble on this thread (implicit/TB) */
int or_last_crit_idx(int *idx, int cap);

/* ASIFimplicitRB: lower end of the interval safety set over x +- x_unc, h_int[i].convert().left()
 * (src/asif_implicit_robust.cpp:635-647); the interval safety set is the model's safetySet written on
 * interval_t, as a user of the class would supply it (no example in the reference constructs one). */
int or_rb_safety_lo(int model, const or_options *o, const double *x, double *hlo);
/* Dh_index_ (first nx entries), Lfh_diff, Lgh_diff[nu] of the last RB or_assemble on this thread */
void or_rb_last_learning(double *Dh_index, double *Lfh_diff, double *Lgh_diff);

/* -------------------------------------------------------------- filter */
enum or_solver_kind { OR_SOLVER_EXACT = 0, OR_SOLVER_ADMM = 1 };

/* One filter() call, cold start.  uAct[nu], relax[nrelax], returns rc
 * (1, -1, TB: 2, -3 or raw solver status). sol_full (optional) gets the nv-vector. */
int or_filter(int model, int variant, const or_options *o, int solver, const or_admm_settings *s,
              const double *x, const double *uDes, double *uAct, double *relax, double *sol_full);

/* Batch loop, AoS inputs x[B][nx], uDes[B][nu]; outputs uAct[B][nu], relax[B][nrelax], rc[B].
 * nthreads<=1 -> plain loop on the calling thread. Returns number of instances processed. */
int64_t or_filter_batch(int model, int variant, const or_options *o, int solver, const or_admm_settings *s,
                        int64_t B, const double *x, const double *uDes, double *uAct, double *relax,
                        int32_t *rc, int nthreads);

/* Batch assembly for row-parity tests: A[B][nc*nv], b[B][nc], code[B]. */
int64_t or_assemble_batch(int model, int variant, const or_options *o, int64_t B, const double *x,
                          double *A, double *b, int32_t *code, double *diag8);

/* Generic batch QP solve (AoS per instance: Hd[nv], c[nv], A[nc*nv], b[nc], lb[nv], ub[nv]). be shared. */
int64_t or_qp_solve_batch(int nv, int nc, int solver, const or_admm_settings *s, int64_t B,
                          const double *Hd, const double *c, const double *A, const double *b,
                          const double *lb, const double *ub, const uint8_t *be,
                          double *sol, int32_t *status, int32_t *iters);

/* ------------------------------------------------------------ realizable */
/* ASIFrealizable (src/asif_realizable.cpp) on a polytopic kernel (include/asif_realizable.h:14-36) with
 * the sampled double integrator of examples/DoubleIntegrator_RealizableSampled.cpp; or_realizable.c. */
typedef struct {
	int32_t nx, nu, nVertices, nFacets, maxCriticalFacets, maxActiveConstraints, npSSmax;
	const double *vertices;       /* [nVertices][nx] */
	const int32_t *facetVertices; /* [nFacets][nx]   */
	const double *facetNormals;   /* [nFacets][nx]   */
	const int32_t *facetActive;   /* [nFacets][maxActiveConstraints] */
	double uncertaintyBounds[OR_MAX_NX];
	double relaxDes, relaxOffset, relaxCost, inf; /* include/asif_realizable.h:14-20 */
	double lb[OR_MAX_NU], ub[OR_MAX_NU];
	double mMin, mMax, Klo, Khi, Flo, Fhi;        /* interval parameters of the example's dynamics */
} or_rz_desc;
typedef struct or_rz or_rz;

void or_rz_default(or_rz_desc *d); /* everything but the kernel arrays, as the example's main() sets it */
or_rz *or_rz_create(const or_rz_desc *d);
void or_rz_destroy(or_rz *z);
void or_rz_dims(const or_rz *z, int *nv, int *nc, int *npSS, int *npSSmax);
/* table[nFacets][maxActive][4] = lo(Lgh), hi(Lgh), lo(Lfh), hi(Lfh) over the facet; bbox[nFacets][nx][2] */
void or_rz_table(const or_rz *z, double *table, double *bbox);
/* returns updateConstraints()'s code (1, or -1 = outside the kernel with no critical facet) */
int or_rz_assemble(const or_rz *z, const double *x, double *A, double *b, int32_t *info);
void or_rz_qp_static(const or_rz *z, const double *uDes, double *Hd, double *c, double *lb, double *ub, uint8_t *be);
/* rc 1, -1 (QP failed), -2 (assembly refused); relax[2] = {solutionFull[nu], solutionFull[nv-1]} */
int or_rz_filter(const or_rz *z, int solver, const or_admm_settings *s, const double *x, const double *uDes,
                 double *uAct, double *relax, double *sol_full);
int64_t or_rz_filter_batch(const or_rz *z, int solver, const or_admm_settings *s, int64_t B, const double *x,
                           const double *uDes, double *uAct, double *relax, int32_t *rc);
int64_t or_rz_assemble_batch(const or_rz *z, int64_t B, const double *x, double *A, double *b, int32_t *code,
                             int32_t *info, int info_stride);

/* ------------------------------------------------- robust filter on shipped half-plane data */
/* ASIFrobust on examples/DoubleIntegrator_Robust.cpp + include/KernelData_*.h; or_robust_data.c. */
typedef struct {
	int32_t N, npSSmax;
	const double *halfPlanes; /* [N][2]: 1 - a.x >= 0 (SafetySetData) */
	double relaxCost, relaxLb, inf;
	double lb[OR_MAX_NU], ub[OR_MAX_NU];
	double mMin, mMax, Klo, Khi, Flo, Fhi;
} or_rb_desc;
typedef struct or_rb or_rb;

void or_rb_default(or_rb_desc *d);
or_rb *or_rb_create(const or_rb_desc *d);
void or_rb_destroy(or_rb *z);
void or_rb_dims(const or_rb *z, int *nv, int *nc, int *npSSmax);
int or_rb_assemble(const or_rb *z, const double *x, double *A, double *b, int32_t *sel);
void or_rb_qp_static(const or_rb *z, const double *uDes, double *Hd, double *c, double *lb, double *ub, uint8_t *be);
int or_rb_filter(const or_rb *z, int solver, const or_admm_settings *s, const double *x, const double *uDes,
                 double *uAct, double *relax);
int64_t or_rb_filter_batch(const or_rb *z, int solver, const or_admm_settings *s, int64_t B, const double *x,
                           const double *uDes, double *uAct, double *relax, int32_t *rc);
int64_t or_rb_assemble_batch(const or_rb *z, int64_t B, const double *x, double *A, double *b, int32_t *code,
                             int32_t *sel);

/* SURVEY 8(d) RNG: splitmix64(seed*2^32 + k) -> r=(z>>11)*2^-53, k = i*16+j */
double or_rng_uniform(uint64_t seed, uint64_t i, uint64_t j);
/* Seeded synthetic batch of config cfg (2..5): fills x[B][nx], uDes[B][nu] */
void or_make_batch(int cfg, int64_t B, int64_t first, double *x, double *uDes);

#ifdef __cplusplus
}
#endif
#endif
