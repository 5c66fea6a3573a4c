/*
 * asif_hip.h -- C ABI of libasif_hip.so: the MI355X-native batched CBF-QP safety filter.
 *
 * This is the drop-in boundary for the reference's per-control-step hot path
 * (DrewSingletary/asif; paths below are relative to that tree):
 *
 *   asif_hip_filter_batch      replaces  ASIF::filter / ASIFimplicit::filter / ASIFimplicitTB::filter /
 *                                        ASIFrobust::filter called once per agent
 *                                        (src/asif.cpp:176-210, src/asif_implicit.cpp:305-356,
 *                                         src/asif_implicit_tb.cpp:261-363, src/asif_robust.cpp:218-252)
 *                                        = updateConstraints + QPWrapperAbstract::{updateCost,updateA,updateb,
 *                                          solve,getSolution} + inputSaturate, fused in one launch.
 *   asif_hip_create_realizable + asif_hip_filter_batch
 *                              replaces  ASIFrealizable::ASIFrealizable / initialize / filter
 *                                        (src/asif_realizable.cpp:5-75,100-267,284-352): facet search incl. the
 *                                        per-facet feasibility QP (:381-441), interval rows, barrier rows, solve.
 *   asif_hip_create_robust_data + asif_hip_filter_batch
 *                              replaces  ASIFrobust with npSSmax < npSS on a half-plane data set, the reference's
 *                                        default-built examples/DoubleIntegrator_Robust.cpp
 *                                        (row selection src/asif_robust.cpp:296-315).
 *   asif_hip_assemble_batch    replaces  updateConstraints alone (src/asif.cpp:233-312,
 *                                        src/asif_implicit.cpp:403-651, src/asif_implicit_tb.cpp:407-733,
 *                                        src/asif_robust.cpp:275-367): rows A, b as handed to updateA/updateb.
 *   asif_hip_qp_solve_batch    replaces  the QPWrapperAbstract solve path for pre-assembled problems
 *                                        (include/qpwrapper_abstract.h:16-51; src/qpwrapper_osqp.cpp:55-261:
 *                                         initialize + solve + getSolution, cold start).
 *   asif_hip_qp_solve_batch_warm  replaces  the second and later solve() calls of one OSQP workspace
 *                                        (src/qpwrapper_osqp.cpp:217-245 on OSQP's default warm_start = 1, which
 *                                         the wrapper leaves on, :68-69): start from the previous x and y.
 *
 * Conventions
 *   - plain C, no C++/torch/HIP types in any signature; `stream` is a hipStream_t passed as void* (NULL = default).
 *   - every array is a DEVICE pointer to FP64 (int32 for codes), structure-of-arrays over the batch:
 *     component k of instance i lives at  base[k * ld + i],  ld >= B  (ld = leading dimension in elements).
 *     Matrices keep the reference's column-major order inside an instance: A(r,c) is component r + c*nc.
 *   - QP convention of the reference (include/qpwrapper_abstract.h:11-15):
 *         min x'Hx + c'x   s.t.  A x >= b  (== b where be[r]),  lb <= x <= ub,   H diagonal.
 *   - return value: 0 on success, a negative ASIF_HIP_E* code, or a positive hipError_t.
 *     Nothing here falls back to the CPU: without a usable GPU every entry point fails.
 *   - calls are asynchronous on `stream`; the caller synchronises.  Handles are not thread-safe.
 */
#ifndef ASIF_HIP_H
#define ASIF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASIF_HIP_VERSION 131 /* 110: realizable / robust-data handles, solver.presolve, scaling_iters 0 = default;
                              * 120: ASIF_HIP_IMPLICIT_RB (options grew at the end), asif_hip_set_learning, asif_hip_affine_replay;
                              * 130: solver.adaptive_rho_interval (struct grew at the end), polish == 2 is the dual active-set
                              *      stage of gi_small.hpp;
                              * 131: that stage eliminates variables pinned by their bounds and solves one-variable problems in
                              *      closed form; class ASIF with one input runs on a kernel without the iterative stages */

enum asif_hip_error {
	ASIF_HIP_OK = 0,
	ASIF_HIP_EINVAL = -1,      /* bad argument / unsupported model-variant pair */
	ASIF_HIP_ENODEVICE = -2,   /* no HIP device, or not gfx950 */
	ASIF_HIP_EUNSUPPORTED = -3 /* QP shape outside the compiled kernels */
};

/* Device models = the user callbacks of the reference's examples, compiled for the GPU
 * (host std::function callbacks cannot run on device, SURVEY 7 "Hard parts"). */
enum asif_hip_model {
	ASIF_HIP_MODEL_DOUBLE_INTEGRATOR = 0,       /* examples/DoubleIntegrator.cpp:12-61          */
	ASIF_HIP_MODEL_INVERTED_PENDULUM = 1,       /* examples/InvertedPendulum_Implicit.cpp:13-80 */
	ASIF_HIP_MODEL_SEGWAY = 2,                  /* examples/segway_implicit_tb.cpp:13-212       */
	ASIF_HIP_MODEL_INVERTED_PENDULUM_ROBUST = 3,/* examples/InvertedPendulum_Robust.cpp:20-79   */
	ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_SAMPLED = 4,/* examples/DoubleIntegrator_RealizableSampled.cpp:16-62 (interval dynamics) */
	ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_ROBUST = 5, /* examples/DoubleIntegrator_Robust.cpp:17-58 (asif_hip_create_robust_data) */
	ASIF_HIP_MODEL_INVERTED_PENDULUM_TB = 6,     /* examples/InvertedPendulum_ImplicitTB.cpp:14-99 (ASIF_HIP_IMPLICIT_TB) */
	ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_IMPLICIT = 7, /* examples/DoubleIntegrator_implicit.cpp:13-90 (ASIF_HIP_IMPLICIT) */
	ASIF_HIP_MODEL_PLANAR_TWO_INPUT = 8, /* NOT an example of the reference: synthetic nx = 2, nu = 2 model (x' = Fx + Gu,
	                                      * five half-planes) so that class ASIF's nu > 1 code (src/asif.cpp:279-303,
	                                      * 314-352) has a device path and a parity test */
	ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_TB = 9 /* examples/DoubleIntegrator_implicit_tb.cpp:13-103 (ASIF_HIP_IMPLICIT_TB) */
};

enum asif_hip_variant {
	ASIF_HIP_EXPLICIT = 0,    /* class ASIF,           include/asif.h:8-101             */
	ASIF_HIP_IMPLICIT = 1,    /* class ASIFimplicit,   include/asif_implicit.h:17-216   */
	ASIF_HIP_IMPLICIT_TB = 2, /* class ASIFimplicitTB, include/asif_implicit_tb.h:17-208 */
	ASIF_HIP_ROBUST = 3,      /* class ASIFrobust,     include/asif_robust.h:11-89      */
	ASIF_HIP_REALIZABLE = 4,  /* class ASIFrealizable, include/asif_realizable.h:9-123 (asif_hip_create_realizable) */
	ASIF_HIP_IMPLICIT_RB = 5  /* class ASIFimplicitRB, include/asif_implicit_robust.h:19-279: ASIFimplicit with the
	                           * backup input held over backContDt, interval safety margins under x_unc and an
	                           * optional learned residual; models INVERTED_PENDULUM, DOUBLE_INTEGRATOR_IMPLICIT */
};

#define ASIF_HIP_MAX_NU 2
#define ASIF_HIP_MAX_NX 4
#define ASIF_HIP_MAX_HALFPLANES 8

/* Union of the reference's four Options structs (include/asif.h:11-17, include/asif_implicit.h:20-34,
 * include/asif_implicit_tb.h:19-33, include/asif_robust.h:14-19) + the input bounds initialize() takes. */
typedef struct asif_hip_options {
	double relaxCost;
	double relaxLb;          /* ASIF/ASIFrobust relaxLb; ASIFimplicit/TB relaxSafeLb */
	double relaxReachLb;     /* ASIFimplicit */
	double relaxTTS;         /* ASIFimplicitTB */
	double relaxMinOrtho;    /* ASIFimplicitTB */
	double backTrajHorizon;
	double backTrajExtend;   /* ASIFimplicitTB */
	double backTrajDt;
	double backTrajMinOrtho; /* ASIFimplicitTB */
	double satSharpness;
	double inf;
	double lb[ASIF_HIP_MAX_NU];
	double ub[ASIF_HIP_MAX_NU];
	/* model data of the robust pendulum (examples/InvertedPendulum_Robust.cpp:35-38,51) */
	double pMin, pMax;
	int32_t nHalfPlanes;
	double halfPlanes[2 * ASIF_HIP_MAX_HALFPLANES]; /* {a0,a1}: 1 - a.x >= 0 */
	/* ASIFimplicitRB::Options extras (include/asif_implicit_robust.h:24-37); ignored by the other variants */
	double backContDt;             /* the backup input is re-sampled every backContDt along the trajectory */
	double x_unc[ASIF_HIP_MAX_NX]; /* state uncertainty radius (Options::x_unc; nullptr there = zeros here) */
	/* in ASIFimplicit::Options too (include/asif_implicit.h:23,33): honoured by ASIF_HIP_IMPLICIT and _IMPLICIT_RB */
	int32_t n_debug;               /* -1: the network sees Dh at the most critical sample; else at this sample */
	int32_t use_learning;          /* needs asif_hip_set_learning before the first filter call */
	/* class ASIF, constructor argument npSSmax (include/asif.h:28, src/asif.cpp:21): <= 0 or >= npSS keeps every safety
	 * function; otherwise the npSSmax rows with the smallest h are kept per call, in ascending order of h
	 * (src/asif.cpp:250-268; ties: lowest index first) and nc = npSSmax */
	int32_t npSSmax;
	/* backup-trajectory integrator of ASIFimplicit and ASIFimplicitTB: 0 = forward Euler (the reference's default
	 * build), 1 = the reference's USE_ODEINT build: dopri5 with dense output at the sample times
	 * (src/asif_implicit.cpp:427-460, src/asif_implicit_tb.cpp:431-463), tolerances Options::backTrajAbsTol /
	 * backTrajRelTol (include/asif_implicit.h:29-30).  ASIFimplicitRB (held input: time-dependent rhs) and the
	 * classes without a trajectory return ASIF_HIP_EUNSUPPORTED for 1. */
	int32_t integrator;
	double backTrajAbsTol, backTrajRelTol;
} asif_hip_options;

/* LearningData (include/asif_learning_utils.h:8-32): two small ReLU networks whose outputs are added to the
 * first row's Lfh / Lgh (update_weights, :123-155).  HOST pointers, dense column-major [rows x cols] as
 * matrixVectorMultiply reads them; asif_hip_set_learning copies.  Inputs >= 2 nx; hidden widths <= 32, input and
 * output widths <= 64 (larger networks: ASIF_HIP_EUNSUPPORTED). */
typedef struct asif_hip_learning_data {
	uint32_t d_drift_in, d_act_in, d_drift_hidden, d_act_hidden, d_drift_hidden_2, d_act_hidden_2, d_drift_out,
	    d_act_out;
	const double *w_1_drift, *w_2_drift, *w_3_drift, *b_1_drift, *b_2_drift, *b_3_drift;
	const double *w_1_act, *w_2_act, *w_3_act, *b_1_act, *b_2_act, *b_3_act;
} asif_hip_learning_data;

/* In-kernel ADMM settings.  Defaults (asif_hip_default_solver) are tuned for |u - u*| <= 1e-6:
 * OSQP-style splitting with power-of-two Ruiz scaling, per-row rho, adaptive rho, infeasibility
 * certificates, and at every termination check an active-set finish seeded by the iterates whose
 * result is accepted only when it is KKT-valid (optimal) or carries a Farkas certificate (infeasible). */
typedef struct asif_hip_solver {
	double rho, sigma, alpha;
	double eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
	double adaptive_rho_tolerance;
	int32_t max_iter;
	int32_t check_interval; /* termination check + polish attempt + rho adaptation every this many iterations; 0 = the path's default */
	int32_t scaling_iters;  /* Ruiz equilibration passes; 0 = the path's default, negative = none */
	int32_t polish;            /* 0: plain ADMM, accuracy set by eps_*; 1: an active-set finish seeded by the iterates at every
	                            * check; 2 (default): additionally, before the first iteration, one run of the in-register dual
	                            * active-set method (gi_small.hpp, problems with nv <= 3 and H > 0), which decides such problems
	                            * outright -- the iterations then only run for what it leaves undecided */
	int32_t active_set_rounds; /* primal-dual working-set corrections per finish */
	int32_t refine_steps;      /* refinement steps of each regularised working-set solve */
	int32_t adaptive_rho;
	int32_t lanes_per_qp;   /* 0 = library default for the shape */
	/* 1: eliminate variables whose bounds pin them (lb == ub) before the solve and, when one free variable is left,
	 * take the closed-form optimum (a clip) instead of iterating.  Applies to the explicit class ASIF, whose
	 * relaxation variable is pinned by construction (src/asif.cpp:88-91).  Same optimum, same return codes.
	 * Default 0.  (Since version 131 the default solver mode, polish == 2, reaches the same one-variable solve through
	 * the dual active-set stage's elimination of pinned variables and takes the same light kernel; presolve = 1 asks
	 * for it under polish 0 / 1 as well.) */
	int32_t presolve;
	/* asif_hip_rollout_batch only.  1 (default): from the second control step on, the first finish attempt starts from
	 * the working set the previous step ended with (iterates still start from zero) -- the batched analogue of the
	 * warm start OSQP gives the reference's closed loops.  The optimum does not depend on it; 0 makes a rollout
	 * bitwise equal to T separate asif_hip_filter_batch calls. */
	int32_t warm_start;
	/* rho is re-estimated every this many iterations (OSQP: adaptive_rho_interval, a multiple of its check period);
	 * 0 = 25.  Independent of check_interval: re-estimating at every check of a 1-2 iteration period never settles. */
	int32_t adaptive_rho_interval;
} asif_hip_solver;

typedef struct asif_hip_dims {
	int32_t nx, nu, npSS, npBS, npBTSS, nv, nc, nrelax, npBT, ndiag;
} asif_hip_dims;

/* Per-instance return codes of asif_hip_filter_batch = the reference's filter() return values. */
#define ASIF_HIP_RC_OK 1                 /* src/asif.cpp:206 */
#define ASIF_HIP_RC_QP_FAILED (-1)       /* src/asif.cpp:209, src/asif_implicit.cpp:354 */
#define ASIF_HIP_RC_IN_BACKUP_SET 2      /* src/asif_implicit_tb.cpp:307 */
#define ASIF_HIP_RC_BACKUP_UNREACHED (-3)/* src/asif_implicit_tb.cpp:360 */
#define ASIF_HIP_RC_OUTSIDE_KERNEL (-2)  /* src/asif_realizable.cpp:324-326: no critical facet and some h < 0 */

/* Solver status values written by asif_hip_qp_solve_batch = QPWrapperOsqp::solve()'s contract
 * (src/qpwrapper_osqp.cpp:225-238): 1 (FEASIBLE) when solved, otherwise the OSQP-style raw status. */
#define ASIF_HIP_STATUS_SOLVED 1
#define ASIF_HIP_STATUS_MAX_ITER (-2)
#define ASIF_HIP_STATUS_PRIMAL_INFEASIBLE (-3)
#define ASIF_HIP_STATUS_DUAL_INFEASIBLE (-4)

typedef struct asif_hip_ctx asif_hip_ctx;

int asif_hip_version(void);
const char *asif_hip_error_string(int code);
int asif_hip_device_count(void);

/* Fills *o with what the named example's main() passes to initialize() (C2..C5 of BASELINE.json). */
int asif_hip_default_options(int model, int variant, asif_hip_options *o);
int asif_hip_default_solver(asif_hip_solver *s);

/* One handle = one (model, variant, options) triple on one device: the analogue of constructing an
 * ASIF* object and calling initialize(lb,ub,opts).  solver may be NULL (defaults).
 * A handle owns staging buffers (assembled rows, trajectory checkpoints) that its launches write: calls on ONE handle
 * must be ordered -- one stream per handle, or the caller's own events between streams.  Handles are independent of
 * one another (asif_hip_create_multi makes one per device and drives each on its own stream). */
int asif_hip_create(asif_hip_ctx **out, int model, int variant, const asif_hip_options *opts,
                    const asif_hip_solver *solver, int device);
int asif_hip_destroy(asif_hip_ctx *ctx);
int asif_hip_get_dims(const asif_hip_ctx *ctx, asif_hip_dims *d);
/* updateOptions(options) of the reference classes (src/asif.cpp:213-231 etc.) */
int asif_hip_update_options(asif_hip_ctx *ctx, const asif_hip_options *opts);
/* Fills learning_data_ (public member of ASIFimplicit, include/asif_implicit.h:125, and of ASIFimplicitRB,
 * include/asif_implicit_robust.h:149) of an ASIF_HIP_IMPLICIT or ASIF_HIP_IMPLICIT_RB handle: uploads the weights.  NULL clears them.  A handle whose options say
 * use_learning without weights fails its filter calls with ASIF_HIP_EINVAL. */
int asif_hip_set_learning(asif_hip_ctx *ctx, const asif_hip_learning_data *L);

/* ---- realizable filter (class ASIFrealizable) ------------------------------------------------------------
 * ASIFrealizable::kernel_t (include/asif_realizable.h:22-36) flattened; HOST arrays, copied by create.
 * nx == 2 (facets are segments), the only dimension the reference ships kernel data for. */
typedef struct asif_hip_kernel_data {
	int32_t nx, nVertices, nFacets;
	int32_t maxCriticalFacets, maxActiveConstraints;
	const double *vertices;       /* [nVertices][nx]                       kernel_t::vertices            */
	const int32_t *facetVertices; /* [nFacets][nx]                         facet_t::verticesIdx          */
	const double *facetNormals;   /* [nFacets][nx]                         facet_t::normal               */
	const int32_t *facetActive;   /* [nFacets][maxActiveConstraints]       facet_t::activeConstraintsSet */
} asif_hip_kernel_data;

/* ASIFrealizable::Options (include/asif_realizable.h:14-20), the constructor's uncertaintyBounds / npSSmax
 * (:40-50), initialize()'s input bounds, and the interval parameters of the device model's dynamics
 * (examples/DoubleIntegrator_RealizableSampled.cpp:27-43: m, K, F as [lo, hi]). */
typedef struct asif_hip_realizable_options {
	double relaxDes, relaxOffset, relaxCost, inf;
	double lb[ASIF_HIP_MAX_NU], ub[ASIF_HIP_MAX_NU];
	double uncertaintyBounds[4];
	int32_t npSSmax; /* barrier rows kept (0..4) */
	double mMin, mMax, Klo, Khi, Flo, Fhi;
} asif_hip_realizable_options;

/* what examples/DoubleIntegrator_RealizableSampled.cpp:19-43,88-94 passes */
int asif_hip_default_realizable_options(int model, asif_hip_realizable_options *o);
/* = new ASIFrealizable(nx,nu,uncertaintyBounds,kernel,dynamics,npSSmax) + initialize(lb,ub,opts).  Uploads the
 * kernel, builds bounding boxes and facet intervals (:137-175) and -- because dynamics_(xFaceInt) does not
 * depend on the state -- the interval Lie derivatives of every (facet, active constraint) pair (:465-506),
 * on the device.  The handle then works with asif_hip_filter_batch / _assemble_batch / _filter_batch_host /
 * _get_dims / _destroy.  filter: relax[2][ld] = {solutionFull[nu], solutionFull[nv-1]} (:346-347), rc 1/-1/-2;
 * diag[ndiag][ld] = {nCriticalFacets, critical facets (maxCriticalFacets, -1 padded), barrier facets (npSSmax),
 * ADMM iterations}. */
int asif_hip_create_realizable(asif_hip_ctx **out, int model, const asif_hip_kernel_data *kernel,
                               const asif_hip_realizable_options *opts, const asif_hip_solver *solver, int device);
/* updateOptions(options), src/asif_realizable.cpp:355-373 (also re-reads bounds and model parameters) */
int asif_hip_update_realizable_options(asif_hip_ctx *ctx, const asif_hip_realizable_options *opts);
/* Copies the device-built tables to HOST arrays (either may be NULL): table[nFacets][maxActive][4] =
 * {lo(Lgh), hi(Lgh), lo(Lfh), hi(Lfh)}, bbox[nFacets][nx][2].  Synchronises the device. */
int asif_hip_realizable_tables(asif_hip_ctx *ctx, double *table, double *bbox);

/* ---- robust filter on a half-plane data set (class ASIFrobust with npSSmax < npSS) --------------------------
 * = the reference's default-built driver examples/DoubleIntegrator_Robust.cpp: safety set  1 - a_i.x >= 0  for the
 * N rows of SafetySetData (include/KernelData_*.h), the npSSmax smallest h kept per call
 * (src/asif_robust.cpp:296-315), dynamics with interval mass, gain and friction (:28-58 of the example). */
typedef struct asif_hip_robust_data_options {
	double relaxCost, relaxLb, inf;            /* ASIFrobust::Options, include/asif_robust.h:14-19 */
	double lb[ASIF_HIP_MAX_NU], ub[ASIF_HIP_MAX_NU];
	int32_t npSSmax;                           /* constructor argument (1..8) */
	double mMin, mMax, Klo, Khi, Flo, Fhi;     /* interval parameters of the device model */
} asif_hip_robust_data_options;

int asif_hip_default_robust_data_options(int model, asif_hip_robust_data_options *o);
/* = new ASIFrobust(nx, nu, N, safetySet, dynamics, npSSmax) + initialize(lb, ub, opts); halfPlanes: HOST [N][2].
 * The handle works with asif_hip_filter_batch / _assemble_batch / _filter_batch_host / _get_dims / _destroy.
 * relax[1][ld]; rc 1 / -1; diag[ndiag][ld] = {kept half-plane indexes (npSSmax), ADMM iterations}. */
int asif_hip_create_robust_data(asif_hip_ctx **out, int model, const double *halfPlanes, int32_t N,
                                const asif_hip_robust_data_options *opts, const asif_hip_solver *solver, int device);
int asif_hip_update_robust_data_options(asif_hip_ctx *ctx, const asif_hip_robust_data_options *opts);

/* B independent filter() calls.  x[nx][ldx], udes[nu][ldx] in; uact[nu][ldx], relax[nrelax][ldx], rc[B] out.
 * Where the reference leaves uAct/relax untouched (QP failed in ASIF/ASIFrobust, relax on any failure)
 * the slots are left untouched too.  diag (may be NULL): [ndiag][ldx] per-instance diagnostics --
 * TB: {TTS_, BTorthoBS_}; all: last slot = ADMM iterations used.  iters (may be NULL): int32[B]. */
int asif_hip_filter_batch(asif_hip_ctx *ctx, int64_t B, int64_t ldx, const double *x, const double *udes,
                          double *uact, double *relax, int32_t *rc, double *diag, void *stream);

/* Class ASIF with caller-supplied Lie derivatives: filter(x, uDes, uAct, Lfh, Lgh[, relax]) (src/asif.cpp:130-165;
 * the override in updateConstraints, :287-292): row i of the QP is [Lgh(i, :) | h_i], b_i = -Lfh_i with h of the kept
 * safety functions and lfh[nc][ldx], lgh[(nc*nu)][ldx] (component i + j*nc) as handed in.  Explicit handles only. */
int asif_hip_filter_batch_lie(asif_hip_ctx *ctx, int64_t B, int64_t ldx, const double *x, const double *udes,
                              const double *lfh, const double *lgh, double *uact, double *relax, int32_t *rc,
                              double *diag, void *stream);

/* Closed loop, T control steps per launch -- the caller's side of filter(), examples/DoubleIntegrator.cpp:81-116:
 * per step  rc = filter(x, uDes, uAct, relax)  (cold start, same arithmetic as asif_hip_filter_batch), then the
 * plant's forward-Euler step  x += dt (f(x) + g(x) uAct)  (:96-110); a failed filter call leaves uAct at its previous
 * value, as the example does.  x[nx][ldx] in/out, udes[nu][ldx] held over the rollout, uact[nu][ldx] in (input applied
 * if the first call fails) / out (last applied), relax[nrelax][ldx] in/out, nfail[B] = steps with rc < 0.  Optional
 * logs (NULL to skip): xlog[T][nx][ldx] = state each filter call saw, ulog[T][nu][ldx] = input applied after it,
 * rclog[T][ldx].
 * ASIF on ASIF_HIP_MODEL_DOUBLE_INTEGRATOR: one fused kernel for the whole rollout.  ASIFimplicit / ASIFimplicitRB /
 * ASIFimplicitTB handles (examples/InvertedPendulum_Implicit.cpp:113-136 and the like): T x (rows kernel, QP kernel,
 * plant-step kernel) queued on `stream` with no host round trip; a failed call applies what filter() wrote to uAct
 * (the saturated backup controller), as those examples do.  Other handles: ASIF_HIP_EUNSUPPORTED. */
int asif_hip_rollout_batch(asif_hip_ctx *ctx, int64_t B, int64_t ldx, int32_t T, double dt, double *x,
                           const double *udes, double *uact, double *relax, int32_t *nfail, double *xlog,
                           double *ulog, int32_t *rclog, void *stream);

/* Rows only: A[(nc*nv)][ldx], b[nc][ldx], code[B] (1; TB: 2 trivial rows, -3 backup set unreached). */
int asif_hip_assemble_batch(asif_hip_ctx *ctx, int64_t B, int64_t ldx, const double *x, double *A, double *b,
                            int32_t *code, double *diag, void *stream);

/* Self-test of the device affine arithmetic (asif_amd/csrc/affine_dev.hpp: libaffa's AAF operations as the robust and
 * realizable rows use them, lib/libaffa/src/aa_aaf*.cpp).  Runs a register program in one lane and returns, per
 * register, centre, symbol count, convert() bounds and the (index, coefficient) pairs ([nreg][16]).  HOST pointers.
 * op: 0 const(imm0), 1 interval(imm0, imm1), 2 a+b, 3 a-b, 4 a*b, 5 a/b, 6 inv(a), 7 -a, 8 a*imm0, 9 sin(a), 10 copy a
 * -- the codes of oracle/ref_affa_shim.cpp, so the golden programs of tests/golden/affa_programs.json replay as they
 * are.  nreg <= 16; a form that would need more than 16 noise symbols returns ASIF_HIP_EUNSUPPORTED. */
typedef struct asif_hip_affine_instr {
	int32_t op, dst, a, b;
	double imm0, imm1;
} asif_hip_affine_instr;
int asif_hip_affine_replay(int device, const asif_hip_affine_instr *prog, int32_t nprog, int32_t nreg, double *center,
                           int32_t *n, double *lo, double *hi, uint32_t *idx, double *coef);

/* Self-test of the device math the trajectory kernels use in place of library calls: n values through one of
 * the functions below, HOST pointers.  kind: 0 sin / cos of a by the fast path (out0 = sin, out1 = cos; |a| <= 1e5),
 * 1 the same with the library fallback beyond that range, 2 tanh (absolute accuracy), 3 1 / a by seed + Newton steps,
 * 4 sqrt(a) and 5 a / b without the IEEE sequences' rescaling steps (their stated operand ranges). */
enum asif_hip_probe {
	ASIF_HIP_PROBE_SINCOS = 0,
	ASIF_HIP_PROBE_SINCOS_CHECKED = 1,
	ASIF_HIP_PROBE_TANH = 2,
	ASIF_HIP_PROBE_RCP = 3,
	ASIF_HIP_PROBE_SQRT_PLAIN = 4,
	ASIF_HIP_PROBE_DIV_PLAIN = 5,
	ASIF_HIP_PROBE_SINCOS_CARRY = 6, /* sin / cos at a + 15 b: evaluated at a, then carried over 15 increments of b */
	ASIF_HIP_PROBE_BEVEL_ARC = 7     /* out0 = sqrt(a), out1 = b / sqrt(a) by the saturation bevel's joint sequence */
};
int asif_hip_math_probe(int device, int32_t kind, int64_t n, const double *a, const double *b, double *out0,
                        double *out1);

/* B pre-assembled QPs of one shape.  Hd[nv][ld] (diagonal of H), c[nv][ld], A[(nc*nv)][ld], b[nc][ld],
 * lb[nv][ld], ub[nv][ld]; be: HOST array of nc flags shared by the batch (NULL = none);
 * sol[nv][ld], status[B], iters[B] (may be NULL).  Cold start per instance.  nv <= 128, nc <= 128.
 * Kernel choice: the filter classes' small shapes (nv <= 3: 2x4, 2x18, 3x41, 3x17, two-variable shapes up to 48
 * rows) run in registers (gi_small.hpp + admm_small.hpp); every other shape runs one wavefront per QP with the
 * factor in LDS (qp_lds.hpp; solver.polish == 0 selects the plain ADMM of admm_wave.hpp where it fits) -- that
 * includes what ASIFrobust / ASIFrealizable hand their solver: 18x12, 22x15, 38x29, 62x47, 86x65
 * (src/asif_robust.cpp:21-22, src/asif_realizable.cpp:19-22).  ASIF_HIP_EUNSUPPORTED only beyond 160 KB of LDS. */
int asif_hip_qp_solve_batch(int device, const asif_hip_solver *solver, int64_t B, int64_t ld, int32_t nv,
                            int32_t nc, const double *Hd, const double *c, const double *A, const double *b,
                            const double *lb, const double *ub, const uint8_t *be, double *sol,
                            int32_t *status, int32_t *iters, void *stream);
/* Same with a full cost matrix: QPWrapperAbstract constructed with diagonalCost = false
 * (src/qpwrapper_osqp.cpp:276-309: P = 2H, the upper triangle is what OSQP is handed, :136-153).
 * H[(nv*nv)][ld], column-major inside an instance; entries (i, j), i <= j are read. */
int asif_hip_qp_solve_batch_dense(int device, const asif_hip_solver *solver, int64_t B, int64_t ld, int32_t nv,
                                  int32_t nc, const double *H, const double *c, const double *A, const double *b,
                                  const double *lb, const double *ub, const uint8_t *be, double *sol,
                                  int32_t *status, int32_t *iters, void *stream);

/* The same solve as the two entries above (pass Hd OR H, the other NULL) with the start OSQP's warm_start = 1 gives
 * the reference's closed loops: between two solve() calls of one workspace OSQP keeps its iterate and multipliers
 * (the wrapper never switches that off, src/qpwrapper_osqp.cpp:68-69 sets max_iter only).
 * warm_x[nv][ld], warm_y[(nc + nv)][ld]: the iterate and the multipliers of the rows [A; I] in the caller's units
 * (OSQP's x and y).  Every call WRITES them -- the final x and y of a problem whose verdict is "solved", zeros
 * (= a cold start) otherwise; warm_in != 0 makes the call READ them first as its starting point (non-finite entries
 * count as zero).  The optimum does not depend on the start: same verdicts, solutions equal within the solver's
 * tolerance, fewer Newton steps (a problem that has not changed since the buffers were written takes none).
 * Applies to the wave-level kernels (qp_inv.hpp / qp_lds.hpp: every shape with nv > 3, full cost matrices), default
 * solver modes; the in-register kernels of the nv <= 3 shapes decide in their exact dual active-set stage and the
 * plain ADMM of solver.polish == 0 keeps its cold two-launch form -- both leave the buffers untouched. */
int asif_hip_qp_solve_batch_warm(int device, const asif_hip_solver *solver, int64_t B, int64_t ld, int32_t nv,
                                 int32_t nc, const double *Hd, const double *H, const double *c, const double *A,
                                 const double *b, const double *lb, const double *ub, const uint8_t *be, double *sol,
                                 int32_t *status, int32_t *iters, double *warm_x, double *warm_y, int32_t warm_in,
                                 void *stream);

/* Host-buffer convenience (pinned or pageable host memory, blocking): H2D, filter, D2H.  AoS->SoA is the
 * caller's business: same [component][ld] layout.  Used by the C++ class mirror for single agents. */
int asif_hip_filter_batch_host(asif_hip_ctx *ctx, int64_t B, const double *x, const double *udes, double *uact,
                               double *relax, int32_t *rc);

/* ---- several GPUs of one node behind one call (BASELINE.json config 4: 262 144 agents over 8 GPUs) ----
 * The batch is cut into contiguous blocks, one per entry of the device list (remainder to the first blocks:
 * asif_hip_partition); every block is uploaded, filtered and downloaded by its own host thread on its own stream.
 * Instances are independent, so there is no collective and no device-to-device traffic: the result arrays are the
 * caller's, each block lands at its own offset.  Results are bitwise those of one handle.  A device may be listed
 * more than once (two streams on one GPU).  asif_hip_create_multi = asif_hip_create once per entry. */
typedef struct asif_hip_multi asif_hip_multi;
int asif_hip_partition(int64_t B, int32_t nblocks, int32_t r, int64_t *first, int64_t *count); /* pure arithmetic */
int asif_hip_create_multi(asif_hip_multi **out, int model, int variant, const asif_hip_options *opts,
                          const asif_hip_solver *solver, int32_t ndev, const int32_t *devs);
int asif_hip_multi_destroy(asif_hip_multi *m);
int asif_hip_multi_size(const asif_hip_multi *m);
asif_hip_ctx *asif_hip_multi_handle(asif_hip_multi *m, int32_t i); /* e.g. for asif_hip_get_dims / _set_learning */
int asif_hip_multi_update_options(asif_hip_multi *m, const asif_hip_options *opts);
/* = asif_hip_filter_batch_host over all handles: HOST buffers x[nx][B], udes[nu][B], uact[nu][B], relax[nrelax][B],
 * rc[B] (pinned or pageable), blocking. */
int asif_hip_filter_batch_host_multi(asif_hip_multi *m, int64_t B, const double *x, const double *udes, double *uact,
                                     double *relax, int32_t *rc);

#ifdef __cplusplus
}
#endif
#endif
