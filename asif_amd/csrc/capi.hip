// capi.hip -- the extern "C" surface declared in include/asif_hip.h.
// Host-side logic only: option defaults, dimension bookkeeping (the constructors and initialize()
// of the reference classes), kernel dispatch.  There is no CPU compute path behind any entry point.
#include "asif_hip.h"
#include "launchers.hpp"
#include "multi_own.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

using namespace asif;

struct asif_hip_ctx {
	int model, variant, device;
	asif_hip_options opts;
	asif_hip_solver solver;
	asif_hip_dims dims;
	DevOptions dev;
	// staging for the host-buffer convenience entry
	double *d_in, *d_out;
	int32_t *d_rc;
	int64_t cap;
	// rows staged between the trajectory kernel and the QP kernel (implicit / TB / robust)
	double *s_rows;
	int64_t s_cap;
	int32_t *s_code;
	int64_t s_code_cap;
	double *s_ckpt; // block checkpoints of the implicit rows kernel
	int64_t s_ckpt_cap;
	// realizable filter: kernel polytope + device-built tables (nullptr for the other variants)
	struct Realizable {
		asif_hip_realizable_options opts;
		RzDev dev;
		double *d_vertices, *d_normals, *d_facetRec, *d_table, *d_pointC;
		int32_t *d_fverts, *d_factive, *d_overflow;
	} *rz;
	// robust filter on a half-plane data set (nullptr otherwise)
	struct RobustData {
		asif_hip_robust_data_options opts;
		RbDev dev;
		double *d_hp, *d_pointC;
	} *rb;
	// ASIFimplicitRB::learning_data_: one device buffer holding both networks (nullptr until set)
	double *d_learn;
	DevOptions::Learn learn;
	// per-step return codes of asif_hip_rollout_batch's stream-ordered path
	int32_t *r_rc;
	int64_t r_cap;
};

extern "C" int asif_hip_version(void) { return ASIF_HIP_VERSION; }

extern "C" const char *asif_hip_error_string(int code)
{
	switch (code) {
	case ASIF_HIP_OK: return "ok";
	case ASIF_HIP_EINVAL: return "invalid argument or unsupported model/variant pair";
	case ASIF_HIP_ENODEVICE: return "no usable HIP device (gfx950 required)";
	case ASIF_HIP_EUNSUPPORTED: return "no compiled kernel for this QP shape";
	default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
	}
}

extern "C" int asif_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// What each example's main() passes to initialize():
//   C2 examples/DoubleIntegrator.cpp:15-16,69-70 (struct defaults of include/asif.h:13-16)
//   C3 examples/InvertedPendulum_Implicit.cpp:19-20,93-97
//   C4 examples/segway_implicit_tb.cpp:18-19,223-230
//   C5 examples/InvertedPendulum_Robust.cpp:23-24,35-38,120-121 + box half-planes
extern "C" int asif_hip_default_options(int model, int variant, asif_hip_options *o)
{
	if (!o) return ASIF_HIP_EINVAL;
	std::memset(o, 0, sizeof(*o));
	o->relaxCost = 50.0;
	o->relaxLb = 5.0;
	o->relaxReachLb = 5.0;
	o->relaxTTS = 5.0;
	o->relaxMinOrtho = 5.0;
	o->backTrajHorizon = 1.0;
	o->backTrajExtend = 0.05;
	o->backTrajDt = 0.01;
	o->backTrajMinOrtho = 0.01;
	o->satSharpness = (variant == ASIF_HIP_EXPLICIT) ? 5.0 : 0.1;
	o->inf = 1e20;
	o->pMin = o->pMax = 1.0;
	o->backContDt = 0.01; // ASIFimplicitRB extras at their header defaults (include/asif_implicit_robust.h:26,31,37)
	o->n_debug = -1;
	o->npSSmax = -1;           // include/asif.h:28: every safety function
	o->integrator = 0;         // forward Euler: the reference's default build (USE_ODEINT off, CMakeLists.txt:20)
	o->backTrajAbsTol = 1.0e-6; // include/asif_implicit.h:29-30
	o->backTrajRelTol = 1.0e-6;
	switch (model) {
	case ASIF_HIP_MODEL_DOUBLE_INTEGRATOR:
		o->lb[0] = -1.0;
		o->ub[0] = 1.0;
		break;
	case ASIF_HIP_MODEL_PLANAR_TWO_INPUT: // synthetic (no reference example has nu > 1): class defaults, |u_k| <= 1
		o->lb[0] = o->lb[1] = -1.0;
		o->ub[0] = o->ub[1] = 1.0;
		break;
	case ASIF_HIP_MODEL_INVERTED_PENDULUM:
		o->lb[0] = -1.5;
		o->ub[0] = 1.5;
		o->backTrajHorizon = 5.0;
		o->backTrajDt = 0.001;
		o->relaxReachLb = 5.0;
		o->relaxLb = 10.0;
		break;
	case ASIF_HIP_MODEL_SEGWAY:
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
	case ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_IMPLICIT: // examples/DoubleIntegrator_implicit.cpp:19-20,89-93
		o->lb[0] = -1.0;
		o->ub[0] = 1.0;
		o->backTrajHorizon = 2.0;
		o->backTrajDt = 0.01;
		o->relaxReachLb = 5.0;
		o->relaxLb = 10.0;
		break;
	case ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_TB: // examples/DoubleIntegrator_implicit_tb.cpp:18-19,107-112
		o->lb[0] = -1.0;
		o->ub[0] = 1.0;
		o->backTrajHorizon = 2.0;
		o->backTrajDt = 0.001;
		o->relaxLb = 10.0;
		o->relaxTTS = 5.0;
		o->relaxMinOrtho = 5.0;
		break;
	case ASIF_HIP_MODEL_INVERTED_PENDULUM_TB: // examples/InvertedPendulum_ImplicitTB.cpp:19-22,106-114
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
	case ASIF_HIP_MODEL_INVERTED_PENDULUM_ROBUST: {
		o->lb[0] = -1.5;
		o->ub[0] = 1.5;
		o->pMin = 0.8;
		o->pMax = 1.2;
		o->nHalfPlanes = 4;
		const double a = 1.0 / M_PI;
		const double hp[8] = {a, 0, -a, 0, 0, a, 0, -a};
		std::memcpy(o->halfPlanes, hp, sizeof(hp));
		break;
	}
	default:
		return ASIF_HIP_EINVAL;
	}
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_default_solver(asif_hip_solver *s)
{
	if (!s) return ASIF_HIP_EINVAL;
	s->rho = 0.1;
	s->sigma = 1e-6;
	s->alpha = 1.6;
	s->eps_abs = 1e-8;
	s->eps_rel = 1e-8;
	s->eps_prim_inf = 1e-4;
	s->eps_dual_inf = 1e-4;
	s->adaptive_rho_tolerance = 5.0;
	s->max_iter = 4000;
	s->check_interval = 0; // 0 = the path's default (1 on the explicit path, 2 elsewhere)
	s->scaling_iters = 0; // power-of-two Ruiz passes; 0 = the path's default (1 explicit / robust pendulum, 2 elsewhere, >= 4 wave kernel)
	s->polish = 2; // finish at every check (1) + the dual active-set stage before the first iteration (2)
	s->active_set_rounds = 12;
	s->refine_steps = 2;
	s->adaptive_rho = 1;
	s->lanes_per_qp = 0;
	s->presolve = 0;
	s->warm_start = 1;
	s->adaptive_rho_interval = 0; // 0 = 25 iterations
	return ASIF_HIP_OK;
}

// Options::inf stands for "no bound" in the rows and bounds the classes write (relaxation upper bounds, the TB class's
// inert rows b = -inf, src/asif_implicit_tb.cpp:727).  OSQP reads any magnitude from OSQP_INFTY = 1e30 on as exactly
// that, so an infinite or huge value (options.inf = INFINITY, 1e300) is the same problem to the reference as 1e30;
// here it is clamped to it, so that the kernels' domain check on the problem data (qp_lane.hpp: qp_data_nonfinite,
// |entry| beyond 1.8e148 or not finite = the solver's max_iter verdict) does not take it for broken data.
static double no_bound_value(double inf) { return inf > 1e30 ? 1e30 : inf; }

static int model_dims(int model, int variant, const asif_hip_options &o, asif_hip_dims &d, DevOptions &dev,
                      bool after_update = false)
{
	std::memset(&d, 0, sizeof(d));
	std::memset(&dev, 0, sizeof(dev));
	dev.relaxCost = o.relaxCost;
	dev.relaxLb = o.relaxLb;
	dev.relaxReachLb = o.relaxReachLb;
	dev.relaxTTS = o.relaxTTS;
	dev.relaxMinOrtho = o.relaxMinOrtho;
	dev.backTrajHorizon = o.backTrajHorizon;
	dev.backTrajDt = o.backTrajDt;
	dev.backTrajMinOrtho = o.backTrajMinOrtho;
	dev.satSharpness = o.satSharpness;
	dev.inf = no_bound_value(o.inf);
	for (int i = 0; i < ASIF_HIP_MAX_NU; i++) {
		dev.lb[i] = o.lb[i];
		dev.ub[i] = o.ub[i];
	}
	dev.pMin = o.pMin;
	dev.pMax = o.pMax;
	dev.nHalfPlanes = o.nHalfPlanes;
	std::memcpy(dev.halfPlanes, o.halfPlanes, sizeof(dev.halfPlanes));
	// src/asif_implicit.cpp:689-703: evaluated with the host libm, once
	dev.bevelL = o.satSharpness * std::tan(M_PI / 8);
	dev.bevelStart = 1 - std::cos(M_PI / 4) * dev.bevelL;
	dev.bevelStop = 1 + dev.bevelL;
	{
		const double lo = std::ldexp(1.0, -100), hi = std::ldexp(1.0, 100);
		const bool ok = o.satSharpness >= lo && o.satSharpness <= hi && dev.bevelStop >= lo && dev.bevelStop <= hi &&
		                dev.bevelStart > 0 && dev.bevelStart < dev.bevelStop && std::isfinite(o.lb[0]) &&
		                std::isfinite(o.ub[0]) && o.lb[0] < o.ub[0] &&
		                // the fast step takes the linear and the clamped region from ONE clamp of u to [lb, ub]: that is
		                // the reference's select only while the thresholds stay clear of 1 by more than the rounding of
		                // uc = (u - middle) 2/range (a sharpness of 1e-15 puts bevelStop an ulp above 1)
		                dev.bevelStop - 1 > 1e-9 && 1 - dev.bevelStart > 1e-9;
		dev.satFastOk = ok ? 1 : 0;
		static const int bevel_free = []() { // developer switch: 0 off, 2 predicts with no margin at all (many repeated blocks)
			const char *v = getenv("ASIF_HIP_BEVEL_FREE");
			return v && v[0] == '0' ? 0 : (v && v[0] == '2' ? 2 : 1);
		}();
		dev.bevelFree = bevel_free;
	}

	dev.satRange = o.ub[0] - o.lb[0];
	dev.satMiddle = (o.ub[0] + o.lb[0]) / 2;
	dev.twoOverRange = 2.0 / dev.satRange;

	// the reference's USE_ODEINT build exists for ASIFimplicit, ASIFimplicitTB and ASIFimplicitRB; the device has it for
	// ASIFimplicit (src/asif_implicit.cpp:427-460)
	if (o.integrator != 0 && o.integrator != 1) return ASIF_HIP_EINVAL;
	if (o.integrator == 1 && variant != ASIF_HIP_IMPLICIT && variant != ASIF_HIP_IMPLICIT_TB) return ASIF_HIP_EUNSUPPORTED; // (ASIFimplicitRB: held input, Euler only)
	if (o.integrator == 1 && !(o.backTrajAbsTol > 0 && o.backTrajRelTol > 0)) return ASIF_HIP_EINVAL;
	dev.integrator = o.integrator;
	dev.trajAbsTol = o.backTrajAbsTol;
	dev.trajRelTol = o.backTrajRelTol;
	if ((model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR || model == ASIF_HIP_MODEL_PLANAR_TWO_INPUT) &&
	    variant == ASIF_HIP_EXPLICIT) {
		const bool p2 = model == ASIF_HIP_MODEL_PLANAR_TWO_INPUT;
		d.nx = 2; d.nu = p2 ? 2 : 1; d.npSS = p2 ? 5 : 4;
		d.nv = d.nu + 1;  // src/asif.cpp:19
		// src/asif.cpp:21-22: npSSmax = -1 (or anything beyond npSS) clamps to npSS
		d.nc = (o.npSSmax > 0 && o.npSSmax < d.npSS) ? o.npSSmax : d.npSS;
		dev.npKeep = d.nc;
		d.nrelax = 1;
		d.ndiag = d.npSS > 3 ? d.npSS : 3; // filter: working-set steps, -, ADMM iterations; assemble: kept safety functions
		return ASIF_HIP_OK;
	}
	if ((model == ASIF_HIP_MODEL_INVERTED_PENDULUM || model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_IMPLICIT) &&
	    (variant == ASIF_HIP_IMPLICIT || variant == ASIF_HIP_IMPLICIT_RB)) {
		d.nx = 2; d.nu = 1; d.npSS = 4; d.npBS = 1;
		// examples/InvertedPendulum_Implicit.cpp:17, examples/DoubleIntegrator_implicit.cpp:17
		d.npBTSS = model == ASIF_HIP_MODEL_INVERTED_PENDULUM ? 10 : 4;
		d.nv = d.nu + 2;                        // src/asif_implicit.cpp:125
		d.nc = d.npBTSS * d.npSS + d.npBS;      // src/asif_implicit.cpp:129
		d.nrelax = 2;
		// src/asif_implicit.cpp:211-216
		double dt = o.backTrajDt;
		int npBT = (int)(std::round(o.backTrajHorizon / dt) + 1);
		if (npBT < d.npBTSS) {
			npBT = d.npBTSS;
			dt = o.backTrajHorizon / (double)(npBT - 1);
		}
		d.npBT = npBT;
		dev.npBT = npBT;
		dev.trajDt = dt;
		d.ndiag = d.npBTSS + 1;                 // critical sample indexes, ADMM iterations
		// n_debug / use_learning exist in both classes (include/asif_implicit.h:23,33; initialize() resets an
		// n_debug outside (-1, npBT-1) to "most critical sample", src/asif_implicit.cpp:219-224)
		dev.nDebug = (o.n_debug > -1 && o.n_debug < npBT - 1) ? o.n_debug : -1;
		dev.useLearning = o.use_learning ? 1 : 0;
		if (variant == ASIF_HIP_IMPLICIT_RB) {
			// critical samples, Dh_index_[nx], Lfh_diff, Lgh_diff[nu], ADMM iterations
			d.ndiag = d.npBTSS + d.nx + 1 + d.nu + 1;
			if (!(o.backContDt > 0)) return ASIF_HIP_EINVAL;
			dev.backContDt = o.backContDt;
			for (int i = 0; i < d.nx; i++) {
				if (!(o.x_unc[i] >= 0)) return ASIF_HIP_EINVAL;
				dev.xUnc[i] = o.x_unc[i];
			}
		}
		return ASIF_HIP_OK;
	}
	if ((model == ASIF_HIP_MODEL_SEGWAY || model == ASIF_HIP_MODEL_INVERTED_PENDULUM_TB ||
	     model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_TB) && variant == ASIF_HIP_IMPLICIT_TB) {
		d.nx = model == ASIF_HIP_MODEL_SEGWAY ? 4 : 2; d.nu = 1; d.npSS = 4; d.npBS = 1;
		d.npBTSS = 4;                           // examples/segway_implicit_tb.cpp:16
		d.nv = d.nu + 1;                        // src/asif_implicit_tb.cpp:122
		d.nc = d.npBTSS * d.npSS + 2;           // src/asif_implicit_tb.cpp:125
		d.nrelax = 1;
		// initialize() stretches the horizon by (1+backTrajExtend) (src/asif_implicit_tb.cpp:177-182);
		// updateOptions() does not (:377-382) -- preserved
		const double T = after_update ? o.backTrajHorizon : o.backTrajHorizon * (1.0 + o.backTrajExtend);
		double dt = o.backTrajDt;
		int npBT = (int)(std::round(T / dt) + 1);
		if (npBT < d.npBTSS) {
			npBT = d.npBTSS;
			dt = T / (double)(npBT - 1);
		}
		d.npBT = npBT;
		dev.npBT = npBT;
		dev.trajDt = dt;
		d.ndiag = 3 + d.npBTSS + 1;             // TTS_, BTorthoBS_, idxHit, critical samples, ADMM iterations
		return ASIF_HIP_OK;
	}
	if (model == ASIF_HIP_MODEL_INVERTED_PENDULUM_ROBUST && variant == ASIF_HIP_ROBUST) {
		if (o.nHalfPlanes < 1 || o.nHalfPlanes > ASIF_HIP_MAX_HALFPLANES) return ASIF_HIP_EINVAL;
		d.nx = 2; d.nu = 1;
		d.npSS = o.nHalfPlanes;                         // npSSmax = npSS
		d.nv = d.nu + 1 + d.npSS * 2 * (d.nu + 1);      // src/asif_robust.cpp:21
		d.nc = d.npSS * (1 + (d.nu + 1));               // src/asif_robust.cpp:22
		d.nrelax = 1;
		d.ndiag = 1;
		return ASIF_HIP_OK;
	}
	return ASIF_HIP_EINVAL;
}

static int check_device(int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ASIF_HIP_ENODEVICE;
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, device) != hipSuccess) return ASIF_HIP_ENODEVICE;
	if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) return ASIF_HIP_ENODEVICE; // code objects are gfx950 only
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_create(asif_hip_ctx **out, int model, int variant, const asif_hip_options *opts,
                               const asif_hip_solver *solver, int device)
{
	if (!out) return ASIF_HIP_EINVAL;
	*out = nullptr;
	asif_hip_options o;
	if (opts) o = *opts;
	else {
		int r = asif_hip_default_options(model, variant, &o);
		if (r) return r;
	}
	asif_hip_ctx *c = new (std::nothrow) asif_hip_ctx();
	if (!c) return ASIF_HIP_EINVAL;
	c->model = model;
	c->variant = variant;
	c->device = device;
	c->opts = o;
	if (solver) c->solver = *solver;
	else asif_hip_default_solver(&c->solver);
	int r = model_dims(model, variant, o, c->dims, c->dev);
	if (r == ASIF_HIP_OK) r = check_device(device);
	if (r != ASIF_HIP_OK) {
		delete c;
		return r;
	}
	c->d_in = c->d_out = nullptr;
	c->d_rc = nullptr;
	c->cap = 0;
	c->s_rows = nullptr;
	c->s_cap = 0;
	c->s_code = nullptr;
	c->s_code_cap = 0;
	c->s_ckpt = nullptr;
	c->s_ckpt_cap = 0;
	c->rz = nullptr;
	c->rb = nullptr;
	c->d_learn = nullptr;
	std::memset(&c->learn, 0, sizeof(c->learn));
	c->r_rc = nullptr;
	c->r_cap = 0;
	*out = c;
	return ASIF_HIP_OK;
}

// ---- ASIFimplicitRB::learning_data_ -------------------------------------------------------------
extern "C" int asif_hip_set_learning(asif_hip_ctx *ctx, const asif_hip_learning_data *L)
{
	if (!ctx || (ctx->variant != ASIF_HIP_IMPLICIT_RB && ctx->variant != ASIF_HIP_IMPLICIT)) return ASIF_HIP_EINVAL;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	// validate and upload the new weights first; the handle keeps its old ones if the call is rejected
	auto drop_old = [&]() {
		if (ctx->d_learn) {
			(void)hipDeviceSynchronize(); // a launch in flight may still read the old weights
			(void)hipFree(ctx->d_learn);
			ctx->d_learn = nullptr;
		}
		std::memset(&ctx->learn, 0, sizeof(ctx->learn));
	};
	if (!L) {
		drop_old();
		return ASIF_HIP_OK;
	}
	const int nx = ctx->dims.nx, nu = ctx->dims.nu;
	const uint32_t din[2] = {L->d_drift_in, L->d_act_in}, h1[2] = {L->d_drift_hidden, L->d_act_hidden},
	               h2[2] = {L->d_drift_hidden_2, L->d_act_hidden_2}, dout[2] = {L->d_drift_out, L->d_act_out};
	const double *w1[2] = {L->w_1_drift, L->w_1_act}, *b1[2] = {L->b_1_drift, L->b_1_act};
	const double *w2[2] = {L->w_2_drift, L->w_2_act}, *b2[2] = {L->b_2_drift, L->b_2_act};
	const double *w3[2] = {L->w_3_drift, L->w_3_act}, *b3[2] = {L->b_3_drift, L->b_3_act};
	size_t total = 0;
	for (int n = 0; n < 2; n++) {
		// update_weights writes 2 nx inputs and reads 1 (drift) / nu (act) outputs (include/asif_learning_utils.h:123-154)
		if ((int)din[n] < 2 * nx || h1[n] < 1 || h2[n] < 1 || (int)dout[n] < (n == 0 ? 1 : nu)) return ASIF_HIP_EINVAL;
		if (h1[n] > 32 || h2[n] > 32 || dout[n] > 64 || din[n] > 64) return ASIF_HIP_EUNSUPPORTED;
		if (!w1[n] || !b1[n] || !w2[n] || !b2[n] || !w3[n] || !b3[n]) return ASIF_HIP_EINVAL;
		// only the first 2 nx input columns of w1 are ever multiplied by non-zeros
		total += (size_t)h1[n] * 2 * nx + h1[n] + (size_t)h2[n] * h1[n] + h2[n] + (size_t)dout[n] * h2[n] + dout[n];
	}
	std::vector<double> host(total);
	size_t off = 0;
	size_t o1[2], ob1[2], o2[2], ob2[2], o3[2], ob3[2];
	for (int n = 0; n < 2; n++) {
		auto put = [&](const double *src, size_t cnt) {
			std::memcpy(host.data() + off, src, cnt * sizeof(double));
			const size_t at = off;
			off += cnt;
			return at;
		};
		o1[n] = put(w1[n], (size_t)h1[n] * 2 * nx); // column-major [h1 x d_in]: the leading 2 nx columns are contiguous
		ob1[n] = put(b1[n], h1[n]);
		o2[n] = put(w2[n], (size_t)h2[n] * h1[n]);
		ob2[n] = put(b2[n], h2[n]);
		o3[n] = put(w3[n], (size_t)dout[n] * h2[n]);
		ob3[n] = put(b3[n], dout[n]);
	}
	double *fresh = nullptr;
	if ((e = hipMalloc((void **)&fresh, total * sizeof(double))) != hipSuccess) return (int)e;
	if ((e = hipMemcpy(fresh, host.data(), total * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) {
		(void)hipFree(fresh);
		return (int)e;
	}
	drop_old();
	ctx->d_learn = fresh;
	for (int n = 0; n < 2; n++) {
		ctx->learn.dHidden[n] = (int)h1[n];
		ctx->learn.dHidden2[n] = (int)h2[n];
		ctx->learn.dOut[n] = (int)dout[n];
		ctx->learn.w1[n] = ctx->d_learn + o1[n];
		ctx->learn.b1[n] = ctx->d_learn + ob1[n];
		ctx->learn.w2[n] = ctx->d_learn + o2[n];
		ctx->learn.b2[n] = ctx->d_learn + ob2[n];
		ctx->learn.w3[n] = ctx->d_learn + o3[n];
		ctx->learn.b3[n] = ctx->d_learn + ob3[n];
	}
	return ASIF_HIP_OK;
}

// ---- realizable filter -------------------------------------------------------------------------
// examples/DoubleIntegrator_RealizableSampled.cpp:19-43,88-94; include/asif_realizable.h:14-20
extern "C" int asif_hip_default_realizable_options(int model, asif_hip_realizable_options *o)
{
	if (!o || model != ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_SAMPLED) return ASIF_HIP_EINVAL;
	std::memset(o, 0, sizeof(*o));
	o->relaxDes = 10.0;
	o->relaxOffset = 0.0;
	o->relaxCost = 100.0;
	o->inf = 1e20;
	o->lb[0] = -20.;
	o->ub[0] = 20.;
	o->uncertaintyBounds[0] = 0.031;
	o->uncertaintyBounds[1] = 0.028;
	o->npSSmax = 2;
	o->mMin = 70.;
	o->mMax = 75.;
	o->Klo = 5.7 - 0.1;
	o->Khi = 5.7 + 0.1;
	o->Flo = 23 - 2; // FInt = interval(F-DF, F-DF), :43
	o->Fhi = 23 - 2;
	return ASIF_HIP_OK;
}

static void rz_free(asif_hip_ctx::Realizable *r)
{
	if (!r) return;
	(void)hipFree(r->d_vertices);
	(void)hipFree(r->d_normals);
	(void)hipFree(r->d_facetRec);
	(void)hipFree(r->d_table);
	(void)hipFree(r->d_pointC);
	(void)hipFree(r->d_fverts);
	(void)hipFree(r->d_factive);
	(void)hipFree(r->d_overflow);
	delete r;
}

// options -> kernel-side struct + dimensions (src/asif_realizable.cpp:19-22), then the table kernel
static int rz_configure(asif_hip_ctx *c, const asif_hip_realizable_options &o)
{
	asif_hip_ctx::Realizable *r = c->rz;
	RzDev &z = r->dev;
	if (o.npSSmax < 0 || o.npSSmax > 4 || o.npSSmax >= z.nF) return ASIF_HIP_EUNSUPPORTED;
	r->opts = o;
	z.npSSmax = o.npSSmax;
	z.npSS = z.maxCrit * z.nA;
	z.nv = (z.npSSmax > 0) ? (1 + z.npSS * 2 * 2 + 1) : (1 + z.npSS * 2 * 2);
	z.nc = z.npSS * 3 + z.npSSmax;
	z.unc[0] = o.uncertaintyBounds[0];
	z.unc[1] = o.uncertaintyBounds[1];
	z.relaxDes = o.relaxDes;
	z.relaxOffset = o.relaxOffset;
	z.relaxCost = o.relaxCost;
	z.inf = no_bound_value(o.inf);
	z.lb = o.lb[0];
	z.ub = o.ub[0];
	z.mMin = o.mMin;
	z.mMax = o.mMax;
	z.Klo = o.Klo;
	z.Khi = o.Khi;
	z.Flo = o.Flo;
	z.Fhi = o.Fhi;
	asif_hip_dims &d = c->dims;
	std::memset(&d, 0, sizeof(d));
	d.nx = 2;
	d.nu = 1;
	d.npSS = z.npSS;
	d.nv = z.nv;
	d.nc = z.nc;
	d.nrelax = 2;
	d.ndiag = 1 + z.maxCrit + z.npSSmax + 1;
	hipError_t e = hipMemset(r->d_overflow, 0, sizeof(int32_t));
	if (e != hipSuccess) return (int)e;
	int rc = launch_realizable_tables(z, r->d_vertices, r->d_fverts, r->d_normals, r->d_factive, r->d_facetRec,
	                                  r->d_table, r->d_pointC, r->d_overflow, nullptr);
	if (rc) return rc;
	int32_t ovf = 0;
	e = hipMemcpy(&ovf, r->d_overflow, sizeof(ovf), hipMemcpyDeviceToHost);
	if (e != hipSuccess) return (int)e;
	return ovf ? ASIF_HIP_EUNSUPPORTED : ASIF_HIP_OK; // affine-form capacity exceeded
}

extern "C" int asif_hip_create_realizable(asif_hip_ctx **out, int model, const asif_hip_kernel_data *k,
                                          const asif_hip_realizable_options *opts, const asif_hip_solver *solver,
                                          int device)
{
	if (!out) return ASIF_HIP_EINVAL;
	*out = nullptr;
	if (model != ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_SAMPLED || !k) return ASIF_HIP_EINVAL;
	if (k->nx != 2) return ASIF_HIP_EUNSUPPORTED;
	if (k->nFacets > 65535) return ASIF_HIP_EUNSUPPORTED; // critical facets are kept as 16-bit indexes
	if (k->nVertices < 2 || k->nFacets < 1 || k->maxCriticalFacets < 1 || k->maxCriticalFacets > 8 ||
	    k->maxActiveConstraints < 1 || !k->vertices || !k->facetVertices || !k->facetNormals || !k->facetActive)
		return ASIF_HIP_EINVAL;
	for (int i = 0; i < k->nFacets * 2; i++)
		if (k->facetVertices[i] < 0 || k->facetVertices[i] >= k->nVertices) return ASIF_HIP_EINVAL;
	for (int i = 0; i < k->nFacets * k->maxActiveConstraints; i++)
		if (k->facetActive[i] < 0 || k->facetActive[i] >= k->nFacets) return ASIF_HIP_EINVAL;
	asif_hip_realizable_options o;
	if (opts) o = *opts;
	else asif_hip_default_realizable_options(model, &o);
	int r = check_device(device);
	if (r) return r;
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) return (int)e;
	asif_hip_ctx *c = new (std::nothrow) asif_hip_ctx();
	if (!c) return ASIF_HIP_EINVAL;
	std::memset(c, 0, sizeof(*c));
	c->model = model;
	c->variant = ASIF_HIP_REALIZABLE;
	c->device = device;
	if (solver) c->solver = *solver;
	else asif_hip_default_solver(&c->solver);
	c->rz = new (std::nothrow) asif_hip_ctx::Realizable();
	if (!c->rz) {
		delete c;
		return ASIF_HIP_EINVAL;
	}
	std::memset(c->rz, 0, sizeof(*c->rz));
	asif_hip_ctx::Realizable *z = c->rz;
	const int nF = k->nFacets, nA = k->maxActiveConstraints, nV = k->nVertices;
	z->dev.nF = nF;
	z->dev.nA = nA;
	z->dev.maxCrit = k->maxCriticalFacets;
	e = hipMalloc((void **)&z->d_vertices, sizeof(double) * nV * 2);
	if (e == hipSuccess) e = hipMalloc((void **)&z->d_normals, sizeof(double) * nF * 2);
	if (e == hipSuccess) e = hipMalloc((void **)&z->d_fverts, sizeof(int32_t) * nF * 2);
	if (e == hipSuccess) e = hipMalloc((void **)&z->d_factive, sizeof(int32_t) * nF * nA);
	if (e == hipSuccess) e = hipMalloc((void **)&z->d_facetRec, sizeof(double) * nF * kRzRec);
	if (e == hipSuccess) e = hipMalloc((void **)&z->d_table, sizeof(double) * nF * nA * 4);
	if (e == hipSuccess) e = hipMalloc((void **)&z->d_pointC, sizeof(double) * kRzPoint);
	if (e == hipSuccess) e = hipMalloc((void **)&z->d_overflow, sizeof(int32_t));
	if (e == hipSuccess) e = hipMemcpy(z->d_vertices, k->vertices, sizeof(double) * nV * 2, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(z->d_normals, k->facetNormals, sizeof(double) * nF * 2, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(z->d_fverts, k->facetVertices, sizeof(int32_t) * nF * 2, hipMemcpyHostToDevice);
	if (e == hipSuccess)
		e = hipMemcpy(z->d_factive, k->facetActive, sizeof(int32_t) * nF * nA, hipMemcpyHostToDevice);
	r = (e == hipSuccess) ? ASIF_HIP_OK : (int)e;
	z->dev.facetRec = z->d_facetRec;
	z->dev.table = z->d_table;
	z->dev.pointC = z->d_pointC;
	if (r == ASIF_HIP_OK) r = rz_configure(c, o);
	if (r != ASIF_HIP_OK) {
		rz_free(c->rz);
		delete c;
		return r;
	}
	*out = c;
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_update_realizable_options(asif_hip_ctx *ctx, const asif_hip_realizable_options *opts)
{
	if (!ctx || !opts || !ctx->rz) return ASIF_HIP_EINVAL;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	const asif_hip_realizable_options keep = ctx->rz->opts;
	int r = rz_configure(ctx, *opts);
	if (r) (void)rz_configure(ctx, keep);
	return r;
}

extern "C" int asif_hip_realizable_tables(asif_hip_ctx *ctx, double *table, double *bbox)
{
	if (!ctx || !ctx->rz) return ASIF_HIP_EINVAL;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	const RzDev &z = ctx->rz->dev;
	if (table) {
		e = hipMemcpy(table, ctx->rz->d_table, sizeof(double) * z.nF * z.nA * 4, hipMemcpyDeviceToHost);
		if (e != hipSuccess) return (int)e;
	}
	if (bbox) {
		double *rec = new (std::nothrow) double[(size_t)z.nF * kRzRec];
		if (!rec) return ASIF_HIP_EINVAL;
		e = hipMemcpy(rec, ctx->rz->d_facetRec, sizeof(double) * z.nF * kRzRec, hipMemcpyDeviceToHost);
		for (int i = 0; i < z.nF && e == hipSuccess; i++)
			for (int q = 0; q < 4; q++) bbox[i * 4 + q] = rec[i * kRzRec + 8 + q];
		delete[] rec;
		if (e != hipSuccess) return (int)e;
	}
	return ASIF_HIP_OK;
}

// ---- robust filter on a half-plane data set ---------------------------------------------------------
// examples/DoubleIntegrator_Robust.cpp:20-37,85-90
extern "C" int asif_hip_default_robust_data_options(int model, asif_hip_robust_data_options *o)
{
	if (!o || model != ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_ROBUST) return ASIF_HIP_EINVAL;
	std::memset(o, 0, sizeof(*o));
	o->relaxCost = 50.0;
	o->relaxLb = 5.0;
	o->inf = 1e20;
	o->lb[0] = -20;
	o->ub[0] = 20;
	o->npSSmax = 5;
	o->mMin = 70.;
	o->mMax = 135.;
	o->Klo = 5.7 - 0.1;
	o->Khi = 5.7 + 0.1;
	o->Flo = 23 - 2; // FInt = interval(F-DF, F-DF), :37
	o->Fhi = 23 - 2;
	return ASIF_HIP_OK;
}

static void rb_free(asif_hip_ctx::RobustData *r)
{
	if (!r) return;
	(void)hipFree(r->d_hp);
	(void)hipFree(r->d_pointC);
	delete r;
}

static int rb_configure(asif_hip_ctx *c, const asif_hip_robust_data_options &o)
{
	asif_hip_ctx::RobustData *r = c->rb;
	RbDev &z = r->dev;
	int M = o.npSSmax;
	if (M < 0 || M > z.N) M = z.N; // src/asif_robust.cpp:20
	if (M < 1 || M > 8) return ASIF_HIP_EUNSUPPORTED;
	r->opts = o;
	z.npSSmax = M;
	z.relaxCost = o.relaxCost;
	z.relaxLb = o.relaxLb;
	z.inf = no_bound_value(o.inf);
	z.lb = o.lb[0];
	z.ub = o.ub[0];
	z.mMin = o.mMin;
	z.mMax = o.mMax;
	z.Klo = o.Klo;
	z.Khi = o.Khi;
	z.Flo = o.Flo;
	z.Fhi = o.Fhi;
	asif_hip_dims &d = c->dims;
	std::memset(&d, 0, sizeof(d));
	d.nx = 2;
	d.nu = 1;
	d.npSS = z.N;
	d.nv = 2 + 4 * M; // src/asif_robust.cpp:21
	d.nc = 3 * M;     // :22
	d.nrelax = 1;
	d.ndiag = M + 1;
	int rc = launch_robust_data_point(z, r->d_pointC, nullptr);
	if (rc) return rc;
	double pc[kRbPoint];
	hipError_t e = hipMemcpy(pc, r->d_pointC, sizeof(pc), hipMemcpyDeviceToHost);
	if (e != hipSuccess) return (int)e;
	return pc[kRbPoint - 1] != 0.0 ? ASIF_HIP_EUNSUPPORTED : ASIF_HIP_OK;
}

extern "C" int asif_hip_create_robust_data(asif_hip_ctx **out, int model, const double *halfPlanes, int32_t N,
                                           const asif_hip_robust_data_options *opts, const asif_hip_solver *solver,
                                           int device)
{
	if (!out) return ASIF_HIP_EINVAL;
	*out = nullptr;
	if (model != ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_ROBUST || !halfPlanes || N < 1) return ASIF_HIP_EINVAL;
	asif_hip_robust_data_options o;
	if (opts) o = *opts;
	else asif_hip_default_robust_data_options(model, &o);
	int r = check_device(device);
	if (r) return r;
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) return (int)e;
	asif_hip_ctx *c = new (std::nothrow) asif_hip_ctx();
	if (!c) return ASIF_HIP_EINVAL;
	std::memset(c, 0, sizeof(*c));
	c->model = model;
	c->variant = ASIF_HIP_ROBUST;
	c->device = device;
	if (solver) c->solver = *solver;
	else asif_hip_default_solver(&c->solver);
	c->rb = new (std::nothrow) asif_hip_ctx::RobustData();
	if (!c->rb) {
		delete c;
		return ASIF_HIP_EINVAL;
	}
	std::memset(c->rb, 0, sizeof(*c->rb));
	c->rb->dev.N = N;
	e = hipMalloc((void **)&c->rb->d_hp, sizeof(double) * 2 * N);
	if (e == hipSuccess) e = hipMalloc((void **)&c->rb->d_pointC, sizeof(double) * kRbPoint);
	if (e == hipSuccess) e = hipMemcpy(c->rb->d_hp, halfPlanes, sizeof(double) * 2 * N, hipMemcpyHostToDevice);
	c->rb->dev.hp = c->rb->d_hp;
	c->rb->dev.pointC = c->rb->d_pointC;
	r = (e == hipSuccess) ? rb_configure(c, o) : (int)e;
	if (r != ASIF_HIP_OK) {
		rb_free(c->rb);
		delete c;
		return r;
	}
	*out = c;
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_update_robust_data_options(asif_hip_ctx *ctx, const asif_hip_robust_data_options *opts)
{
	if (!ctx || !opts || !ctx->rb) return ASIF_HIP_EINVAL;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	const asif_hip_robust_data_options keep = ctx->rb->opts;
	int r = rb_configure(ctx, *opts);
	if (r) (void)rb_configure(ctx, keep);
	return r;
}

extern "C" int asif_hip_destroy(asif_hip_ctx *ctx)
{
	if (!ctx) return ASIF_HIP_EINVAL;
	if (ctx->rz) {
		(void)hipSetDevice(ctx->device);
		rz_free(ctx->rz);
	}
	if (ctx->rb) {
		(void)hipSetDevice(ctx->device);
		rb_free(ctx->rb);
	}
	if (ctx->d_learn) {
		(void)hipSetDevice(ctx->device);
		(void)hipFree(ctx->d_learn);
	}
	if (ctx->r_rc) {
		(void)hipSetDevice(ctx->device);
		(void)hipFree(ctx->r_rc);
	}
	if (ctx->s_rows) {
		(void)hipSetDevice(ctx->device);
		(void)hipFree(ctx->s_rows);
	}
	if (ctx->s_code) {
		(void)hipSetDevice(ctx->device);
		(void)hipFree(ctx->s_code);
	}
	if (ctx->s_ckpt) {
		(void)hipSetDevice(ctx->device);
		(void)hipFree(ctx->s_ckpt);
	}
	if (ctx->d_in) {
		(void)hipSetDevice(ctx->device);
		(void)hipFree(ctx->d_in);
		(void)hipFree(ctx->d_out);
		(void)hipFree(ctx->d_rc);
	}
	delete ctx;
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_get_dims(const asif_hip_ctx *ctx, asif_hip_dims *d)
{
	if (!ctx || !d) return ASIF_HIP_EINVAL;
	*d = ctx->dims;
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_update_options(asif_hip_ctx *ctx, const asif_hip_options *opts)
{
	if (!ctx || !opts) return ASIF_HIP_EINVAL;
	asif_hip_dims d;
	DevOptions dev;
	int r = model_dims(ctx->model, ctx->variant, *opts, d, dev, true);
	if (r) return r;
	ctx->opts = *opts;
	ctx->dims = d;
	ctx->dev = dev;
	return ASIF_HIP_OK;
}

// Two-stage variants park their rows in a per-handle staging buffer, grown on demand (the only
// allocation a filter call can make; size it once with a first call before capturing a graph).
static int stage_rows(asif_hip_ctx *ctx, FilterArgs &a)
{
	const int64_t nA = (int64_t)ctx->dims.nc * ctx->dims.nv, nb = ctx->dims.nc;
	const int64_t need = (nA + nb) * a.ld;
	if (need > ctx->s_cap) {
		if (ctx->s_rows) (void)hipFree(ctx->s_rows);
		ctx->s_rows = nullptr;
		ctx->s_cap = 0;
		hipError_t e = hipMalloc((void **)&ctx->s_rows, sizeof(double) * need);
		if (e != hipSuccess) return (int)e;
		ctx->s_cap = need;
	}
	if (a.B > ctx->s_code_cap) {
		if (ctx->s_code) (void)hipFree(ctx->s_code);
		ctx->s_code = nullptr;
		ctx->s_code_cap = 0;
		hipError_t e = hipMalloc((void **)&ctx->s_code, sizeof(int32_t) * a.B);
		if (e != hipSuccess) return (int)e;
		ctx->s_code_cap = a.B;
	}
	a.A = ctx->s_rows;
	a.b = ctx->s_rows + nA * a.ld;
	a.code = ctx->s_code;
	return 0;
}

// The trajectory kernels keep checkpoints of the backup trajectory in HBM (pass 1) and re-integrate the few blocks
// that hold the critical samples from them (pass 2); sized per handle, grown on demand: the npBTSS selected
// checkpoints, [npBTSS][nz + 2][ld] (the block-start state rides in registers over the block and is stored only
// when the block enters the selection).
static int stage_ckpt(asif_hip_ctx *ctx, FilterArgs &a)
{
	const int64_t nz = ctx->dims.nx + ctx->dims.nx * ctx->dims.nx;
	int64_t need;
	switch (ctx->model) {
	case ASIF_HIP_MODEL_INVERTED_PENDULUM:
	case ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_IMPLICIT:
	case ASIF_HIP_MODEL_INVERTED_PENDULUM_TB:
	case ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_TB:
	case ASIF_HIP_MODEL_SEGWAY: need = (int64_t)ctx->dims.npBTSS * (nz + 2) * a.ld; break;
	default: return ASIF_HIP_EINVAL;
	}
	if (need > ctx->s_ckpt_cap) {
		if (ctx->s_ckpt) (void)hipFree(ctx->s_ckpt);
		ctx->s_ckpt = nullptr;
		ctx->s_ckpt_cap = 0;
		hipError_t e = hipMalloc((void **)&ctx->s_ckpt, sizeof(double) * need);
		if (e != hipSuccess) return (int)e;
		ctx->s_ckpt_cap = need;
	}
	a.ckpt = ctx->s_ckpt;
	return 0;
}

static int run_filter(asif_hip_ctx *ctx, FilterArgs a, bool assemble_only, hipStream_t stream)
{
	if (ctx->variant == ASIF_HIP_REALIZABLE)
		return ctx->rz ? launch_realizable(ctx->rz->dev, ctx->solver, a, assemble_only, stream) : ASIF_HIP_EINVAL;
	if (ctx->rb) return launch_robust_data(ctx->rb->dev, ctx->solver, a, assemble_only, stream);
	if (ctx->model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR && ctx->variant == ASIF_HIP_EXPLICIT)
		return launch_explicit_di(ctx->dev, ctx->solver, a, assemble_only, stream);
	if (ctx->model == ASIF_HIP_MODEL_PLANAR_TWO_INPUT && ctx->variant == ASIF_HIP_EXPLICIT)
		return launch_explicit_p2(ctx->dev, ctx->solver, a, assemble_only, stream);
	// ASIFimplicitRB, and ASIFimplicit with its learned residual switched on (src/asif_implicit.cpp:585-588):
	// the latter runs the RB rows kernel with the hold off (backContDt = 0) and zero uncertainty, which is
	// bitwise the plain kernel plus the residual
	if (ctx->variant == ASIF_HIP_IMPLICIT_RB || (ctx->variant == ASIF_HIP_IMPLICIT && ctx->dev.useLearning)) {
		DevOptions dev = ctx->dev;
		if (dev.useLearning) {
			if (!ctx->d_learn) return ASIF_HIP_EINVAL; // the reference would dereference null weights
			dev.learn = ctx->learn;
		}
		if (!assemble_only) {
			int r = stage_rows(ctx, a);
			if (r) return r;
		}
		if (int r = stage_ckpt(ctx, a)) return r;
		if (ctx->model == ASIF_HIP_MODEL_INVERTED_PENDULUM)
			return launch_implicit_ip(dev, ctx->solver, a, assemble_only, stream, true);
		if (ctx->model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_IMPLICIT)
			return launch_implicit_di(dev, ctx->solver, a, assemble_only, stream, true);
		return ASIF_HIP_EINVAL;
	}
	if (ctx->model == ASIF_HIP_MODEL_INVERTED_PENDULUM && ctx->variant == ASIF_HIP_IMPLICIT) {
		if (!assemble_only) {
			int r = stage_rows(ctx, a);
			if (r) return r;
		}
		if (int r = stage_ckpt(ctx, a)) return r;
		return launch_implicit_ip(ctx->dev, ctx->solver, a, assemble_only, stream);
	}
	if (ctx->model == ASIF_HIP_MODEL_SEGWAY && ctx->variant == ASIF_HIP_IMPLICIT_TB) {
		if (!assemble_only) {
			int r = stage_rows(ctx, a);
			if (r) return r;
		}
		if (int r = stage_ckpt(ctx, a)) return r;
		return launch_tb_segway(ctx->dev, ctx->solver, a, assemble_only, stream);
	}
	if (ctx->model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_IMPLICIT && ctx->variant == ASIF_HIP_IMPLICIT) {
		if (!assemble_only) {
			int r = stage_rows(ctx, a);
			if (r) return r;
		}
		if (int r = stage_ckpt(ctx, a)) return r;
		return launch_implicit_di(ctx->dev, ctx->solver, a, assemble_only, stream);
	}
	if (ctx->model == ASIF_HIP_MODEL_INVERTED_PENDULUM_TB && ctx->variant == ASIF_HIP_IMPLICIT_TB) {
		if (!assemble_only) {
			int r = stage_rows(ctx, a);
			if (r) return r;
		}
		if (int r = stage_ckpt(ctx, a)) return r;
		return launch_tb_pendulum(ctx->dev, ctx->solver, a, assemble_only, stream);
	}
	if (ctx->model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_TB && ctx->variant == ASIF_HIP_IMPLICIT_TB) {
		if (!assemble_only) {
			int r = stage_rows(ctx, a);
			if (r) return r;
		}
		if (int r = stage_ckpt(ctx, a)) return r;
		return launch_tb_di(ctx->dev, ctx->solver, a, assemble_only, stream);
	}
	if (ctx->model == ASIF_HIP_MODEL_INVERTED_PENDULUM_ROBUST && ctx->variant == ASIF_HIP_ROBUST)
		return launch_robust_ip(ctx->dev, ctx->solver, a, assemble_only, stream); // fused, nothing staged
	return ASIF_HIP_EINVAL;
}

extern "C" int asif_hip_filter_batch(asif_hip_ctx *ctx, int64_t B, int64_t ldx, const double *x, const double *udes,
                                     double *uact, double *relax, int32_t *rc, double *diag, void *stream)
{
	if (!ctx || B < 0 || ldx < B || (B > 0 && (!x || !udes || !uact || !relax || !rc))) return ASIF_HIP_EINVAL;
	if (B == 0) return ASIF_HIP_OK;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	FilterArgs a = {B, ldx, x, udes, uact, relax, rc, diag, ctx->dims.ndiag, nullptr, nullptr, nullptr};
	return run_filter(ctx, a, false, (hipStream_t)stream);
}

extern "C" int asif_hip_filter_batch_lie(asif_hip_ctx *ctx, int64_t B, int64_t ldx, const double *x, const double *udes,
                                         const double *lfh, const double *lgh, double *uact, double *relax, int32_t *rc,
                                         double *diag, void *stream)
{
	if (!ctx || B < 0 || ldx < B || (B > 0 && (!x || !udes || !lfh || !lgh || !uact || !relax || !rc)))
		return ASIF_HIP_EINVAL;
	if (ctx->variant != ASIF_HIP_EXPLICIT || ctx->rz || ctx->rb) return ASIF_HIP_EINVAL;
	if (B == 0) return ASIF_HIP_OK;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	FilterArgs a = {B, ldx, x, udes, uact, relax, rc, diag, ctx->dims.ndiag, nullptr, nullptr, nullptr, nullptr, lfh, lgh};
	return run_filter(ctx, a, false, (hipStream_t)stream);
}

extern "C" int asif_hip_rollout_batch(asif_hip_ctx *ctx, int64_t B, int64_t ldx, int32_t T, double dt, double *x,
                                      const double *udes, double *uact, double *relax, int32_t *nfail, double *xlog,
                                      double *ulog, int32_t *rclog, void *stream)
{
	if (!ctx || B < 0 || ldx < B || T < 0 || (B > 0 && (!x || !udes || !uact || !relax || !nfail))) return ASIF_HIP_EINVAL;
	const bool fused = ctx->model == ASIF_HIP_MODEL_DOUBLE_INTEGRATOR && ctx->variant == ASIF_HIP_EXPLICIT;
	const bool staged = !ctx->rz && !ctx->rb &&
	                    (ctx->variant == ASIF_HIP_IMPLICIT || ctx->variant == ASIF_HIP_IMPLICIT_RB ||
	                     ctx->variant == ASIF_HIP_IMPLICIT_TB);
	if (!fused && !staged) return ASIF_HIP_EUNSUPPORTED;
	if (B == 0 || T == 0) return ASIF_HIP_OK;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	if (fused) {
		const RolloutArgs a = {B, ldx, T, dt, x, udes, uact, relax, nfail, xlog, ulog, rclog};
		return launch_rollout_explicit_di(ctx->dev, ctx->solver, a, (hipStream_t)stream);
	}
	// two-stage filters: T x (rows kernel, QP kernel, plant step) in stream order, nothing returns to the host
	hipStream_t s = (hipStream_t)stream;
	if (B > ctx->r_cap) {
		if (ctx->r_rc) (void)hipFree(ctx->r_rc);
		ctx->r_rc = nullptr;
		ctx->r_cap = 0;
		if ((e = hipMalloc((void **)&ctx->r_rc, sizeof(int32_t) * B)) != hipSuccess) return (int)e;
		ctx->r_cap = B;
	}
	if ((e = hipMemsetAsync(nfail, 0, sizeof(int32_t) * B, s)) != hipSuccess) return (int)e;
	const asif_hip_dims &d = ctx->dims;
	for (int32_t t = 0; t < T; t++) {
		FilterArgs a = {B, ldx, x, udes, uact, relax, ctx->r_rc, nullptr, d.ndiag, nullptr, nullptr, nullptr};
		int r = run_filter(ctx, a, false, s);
		if (r) return r;
		r = launch_plant_step(ctx->model, ctx->dev, B, ldx, dt, x, uact, ctx->r_rc, nfail,
		                      xlog ? xlog + (int64_t)t * d.nx * ldx : nullptr,
		                      ulog ? ulog + (int64_t)t * d.nu * ldx : nullptr, rclog ? rclog + (int64_t)t * ldx : nullptr, s);
		if (r) return r;
	}
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_assemble_batch(asif_hip_ctx *ctx, int64_t B, int64_t ldx, const double *x, double *A,
                                       double *b, int32_t *code, double *diag, void *stream)
{
	if (!ctx || B < 0 || ldx < B || (B > 0 && (!x || !A || !b || !code))) return ASIF_HIP_EINVAL;
	if (B == 0) return ASIF_HIP_OK;
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	FilterArgs a = {B, ldx, x, x /*unused*/, nullptr, nullptr, nullptr, diag, ctx->dims.ndiag, A, b, code};
	return run_filter(ctx, a, true, (hipStream_t)stream);
}

static int qp_solve_common(int device, const asif_hip_solver *solver, int64_t B, int64_t ld, int32_t nv, int32_t nc,
                           const double *Hd, const double *H, const double *c, const double *A, const double *b,
                           const double *lb, const double *ub, const uint8_t *be, double *sol, int32_t *status,
                           int32_t *iters, void *stream, double *warm_x = nullptr, double *warm_y = nullptr,
                           int warm_in = 0)
{
	if (B < 0 || ld < B || nv < 1 || nc < 0) return ASIF_HIP_EINVAL;
	if (nc > 128 || nv > 128) return ASIF_HIP_EUNSUPPORTED;
	if (B == 0) return ASIF_HIP_OK;
	if ((!Hd && !H) || !c || (nc > 0 && (!A || !b)) || !lb || !ub || !sol || !status) return ASIF_HIP_EINVAL;
	int r = check_device(device);
	if (r) return r;
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) return (int)e;
	asif_hip_solver S;
	if (solver) S = *solver;
	else asif_hip_default_solver(&S);
	uint64_t mask = 0, mask2 = 0;
	if (be)
		for (int i = 0; i < nc; i++)
			if (be[i]) (i < 64 ? mask : mask2) |= (1ull << (i & 63));
	QpArgs a = {B, ld, nv, nc, Hd, c, A, b, lb, ub, mask, sol, status, iters, H, mask2};
	if (warm_x && warm_y && S.polish != 0) { // the plain ADMM of polish == 0 has its own two-launch form: cold
		a.warm_x = warm_x;
		a.warm_y = warm_y;
		a.warm_in = warm_in;
	}
	return launch_qp_small(S, a, (hipStream_t)stream);
}

extern "C" int asif_hip_qp_solve_batch_warm(int device, const asif_hip_solver *solver, int64_t B, int64_t ld,
                                            int32_t nv, int32_t nc, const double *Hd, const double *H, const double *c,
                                            const double *A, const double *b, const double *lb, const double *ub,
                                            const uint8_t *be, double *sol, int32_t *status, int32_t *iters,
                                            double *warm_x, double *warm_y, int32_t warm_in, void *stream)
{
	if ((!Hd) == (!H)) return B == 0 && (Hd || H) ? ASIF_HIP_OK : ASIF_HIP_EINVAL; // exactly one form of the cost
	if (B > 0 && (!warm_x || !warm_y)) return ASIF_HIP_EINVAL;
	return qp_solve_common(device, solver, B, ld, nv, nc, Hd, H, c, A, b, lb, ub, be, sol, status, iters, stream, warm_x,
	                       warm_y, warm_in != 0);
}

extern "C" int asif_hip_qp_solve_batch(int device, const asif_hip_solver *solver, int64_t B, int64_t ld, int32_t nv,
                                       int32_t nc, const double *Hd, const double *c, const double *A,
                                       const double *b, const double *lb, const double *ub, const uint8_t *be,
                                       double *sol, int32_t *status, int32_t *iters, void *stream)
{
	if (!Hd) return B == 0 ? ASIF_HIP_OK : ASIF_HIP_EINVAL;
	return qp_solve_common(device, solver, B, ld, nv, nc, Hd, nullptr, c, A, b, lb, ub, be, sol, status, iters, stream);
}

extern "C" int asif_hip_qp_solve_batch_dense(int device, const asif_hip_solver *solver, int64_t B, int64_t ld,
                                             int32_t nv, int32_t nc, const double *H, const double *c, const double *A,
                                             const double *b, const double *lb, const double *ub, const uint8_t *be,
                                             double *sol, int32_t *status, int32_t *iters, void *stream)
{
	if (!H) return B == 0 ? ASIF_HIP_OK : ASIF_HIP_EINVAL;
	return qp_solve_common(device, solver, B, ld, nv, nc, nullptr, H, c, A, b, lb, ub, be, sol, status, iters, stream);
}

// Host buffers, SoA with leading dimension ldh (>= B): component k of instance i at base[k * ldh + i].  Strided
// component copies go as one 2-D copy per array; the device side is dense (ld = B).
static int filter_host_strided(asif_hip_ctx *ctx, int64_t B, int64_t ldh, const double *x, const double *udes,
                               double *uact, double *relax, int32_t *rc, hipStream_t s)
{
	hipError_t e = hipSetDevice(ctx->device);
	if (e != hipSuccess) return (int)e;
	const asif_hip_dims &d = ctx->dims;
	const int nin = d.nx + d.nu, nout = d.nu + d.nrelax;
	// Zero copy: when every buffer is page-locked host memory the device can address (hipHostMalloc / hipHostRegister,
	// what a host that cares about transfer time hands over), the kernels read the inputs and write the results in
	// place across the link -- one launch instead of seven staged copies and their launch latencies, and the slots
	// the filter leaves untouched keep the caller's values by construction.  Coalesced 512-byte requests per wave, each
	// byte crosses the link once: the call takes what the link takes.
	{
		void *dp[5];
		const void *hp[5] = {x, udes, uact, relax, rc};
		bool mapped = true;
		for (int k = 0; k < 5 && mapped; k++) {
			hipPointerAttribute_t at;
			if (hipPointerGetAttributes(&at, hp[k]) != hipSuccess) {
				(void)hipGetLastError(); // plain pageable memory: not an error, just not this path
				mapped = false;
			} else {
				mapped = at.type == hipMemoryTypeHost && at.devicePointer != nullptr;
				dp[k] = at.devicePointer;
			}
		}
		if (mapped) {
			const int r = asif_hip_filter_batch(ctx, B, ldh, (const double *)dp[0], (const double *)dp[1], (double *)dp[2],
			                                    (double *)dp[3], (int32_t *)dp[4], nullptr, s);
			if (r) return r;
			return (int)hipStreamSynchronize(s);
		}
	}
	if (B > ctx->cap) {
		if (ctx->d_in) {
			(void)hipFree(ctx->d_in);
			(void)hipFree(ctx->d_out);
			(void)hipFree(ctx->d_rc);
			ctx->d_in = ctx->d_out = nullptr;
			ctx->d_rc = nullptr;
			ctx->cap = 0;
		}
		if ((e = hipMalloc((void **)&ctx->d_in, sizeof(double) * nin * B)) != hipSuccess) return (int)e;
		if ((e = hipMalloc((void **)&ctx->d_out, sizeof(double) * nout * B)) != hipSuccess) return (int)e;
		if ((e = hipMalloc((void **)&ctx->d_rc, sizeof(int32_t) * B)) != hipSuccess) return (int)e;
		ctx->cap = B;
	}
	double *dx = ctx->d_in, *du = ctx->d_in + (int64_t)d.nx * B;
	double *dua = ctx->d_out, *drl = ctx->d_out + (int64_t)d.nu * B;
	const size_t w = sizeof(double) * B, hp = sizeof(double) * ldh;
	auto up = [&](double *dst, const double *src, int rows) {
		return hipMemcpy2DAsync(dst, w, src, hp, w, (size_t)rows, hipMemcpyHostToDevice, s);
	};
	auto down = [&](double *dst, const double *src, int rows) {
		return hipMemcpy2DAsync(dst, hp, src, w, w, (size_t)rows, hipMemcpyDeviceToHost, s);
	};
	if ((e = up(dx, x, d.nx)) != hipSuccess) return (int)e;
	if ((e = up(du, udes, d.nu)) != hipSuccess) return (int)e;
	// slots the kernel leaves untouched must keep the caller's values
	if ((e = up(dua, uact, d.nu)) != hipSuccess) return (int)e;
	if ((e = up(drl, relax, d.nrelax)) != hipSuccess) return (int)e;
	int r = asif_hip_filter_batch(ctx, B, B, dx, du, dua, drl, ctx->d_rc, nullptr, s);
	if (r) return r;
	if ((e = down(uact, dua, d.nu)) != hipSuccess) return (int)e;
	if ((e = down(relax, drl, d.nrelax)) != hipSuccess) return (int)e;
	if ((e = hipMemcpyAsync(rc, ctx->d_rc, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s)) != hipSuccess) return (int)e;
	return (int)hipStreamSynchronize(s);
}

extern "C" int asif_hip_filter_batch_host(asif_hip_ctx *ctx, int64_t B, const double *x, const double *udes,
                                          double *uact, double *relax, int32_t *rc)
{
	if (!ctx || B < 0) return ASIF_HIP_EINVAL;
	if (B == 0) return ASIF_HIP_OK;
	if (!x || !udes || !uact || !relax || !rc) return ASIF_HIP_EINVAL;
	return filter_host_strided(ctx, B, B, x, udes, uact, relax, rc, nullptr);
}

// ---- several devices driven together: contiguous blocks of the batch, one host thread + stream
// per handle, no collective -- the "gather" is the pointer offset of each block in the caller's arrays.
struct asif_hip_multi {
	std::vector<asif_hip_ctx *> h;
	std::vector<hipStream_t> s;
};

extern "C" int asif_hip_partition(int64_t B, int32_t n, int32_t r, int64_t *first, int64_t *count)
{
	if (B < 0 || n < 1 || r < 0 || r >= n || !first || !count) return ASIF_HIP_EINVAL;
	const int64_t q = B / n, rem = B % n; // remainder to the first blocks
	*count = q + (r < rem ? 1 : 0);
	*first = q * r + (r < rem ? r : rem);
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_multi_destroy(asif_hip_multi *m)
{
	if (!m) return ASIF_HIP_EINVAL;
	for (size_t i = 0; i < m->h.size(); i++) {
		if (m->s[i]) {
			(void)hipSetDevice(m->h[i]->device);
			(void)hipStreamDestroy(m->s[i]);
		}
		if (m->h[i]) asif_hip_destroy(m->h[i]);
	}
	delete m;
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_create_multi(asif_hip_multi **out, int model, int variant, const asif_hip_options *opts,
                                     const asif_hip_solver *solver, int32_t ndev, const int32_t *devs)
{
	if (!out || ndev < 1 || !devs) return ASIF_HIP_EINVAL;
	*out = nullptr;
	asif_hip_multi *m = new (std::nothrow) asif_hip_multi();
	if (!m) return ASIF_HIP_EINVAL;
	// one handle per entry of the device list (a device may appear more than once), then one stream per handle;
	// ownership on failure: multi_own.hpp (every handle and stream released exactly once, by this function)
	const int r = create_all(
	    ndev, m->h, m->s,
	    [&](int i, asif_hip_ctx **c) { return asif_hip_create(c, model, variant, opts, solver, devs[i]); },
	    [](asif_hip_ctx *c) { asif_hip_destroy(c); },
	    [](asif_hip_ctx *c, hipStream_t *s) {
		    hipError_t e = hipSetDevice(c->device);
		    if (e == hipSuccess) e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
		    return (int)e;
	    },
	    [](asif_hip_ctx *c, hipStream_t s) {
		    (void)hipSetDevice(c->device);
		    (void)hipStreamDestroy(s);
	    });
	if (r) {
		delete m;
		return r;
	}
	*out = m;
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_multi_size(const asif_hip_multi *m) { return m ? (int)m->h.size() : 0; }

extern "C" asif_hip_ctx *asif_hip_multi_handle(asif_hip_multi *m, int32_t i)
{
	return (m && i >= 0 && (size_t)i < m->h.size()) ? m->h[i] : nullptr;
}

extern "C" int asif_hip_multi_update_options(asif_hip_multi *m, const asif_hip_options *opts)
{
	if (!m || !opts) return ASIF_HIP_EINVAL;
	// all or nothing: the options are checked against every handle before any handle takes them, so that a refusal
	// leaves the blocks of one batch on ONE set of options (the check is host arithmetic; nothing below it can fail)
	for (asif_hip_ctx *h : m->h) {
		asif_hip_dims d;
		DevOptions dev;
		if (int r = model_dims(h->model, h->variant, *opts, d, dev, true)) return r;
	}
	for (asif_hip_ctx *h : m->h)
		if (int r = asif_hip_update_options(h, opts)) return r;
	return ASIF_HIP_OK;
}

extern "C" int asif_hip_filter_batch_host_multi(asif_hip_multi *m, int64_t B, const double *x, const double *udes,
                                                double *uact, double *relax, int32_t *rc)
{
	if (!m || m->h.empty() || B < 0) return ASIF_HIP_EINVAL;
	if (B == 0) return ASIF_HIP_OK;
	if (!x || !udes || !uact || !relax || !rc) return ASIF_HIP_EINVAL;
	const int n = (int)m->h.size();
	std::vector<int> res(n, 0);
	std::vector<std::thread> th;
	for (int r = 0; r < n; r++) {
		int64_t first, count;
		asif_hip_partition(B, n, r, &first, &count);
		if (count == 0) continue;
		th.emplace_back([=, &res]() {
			res[r] = filter_host_strided(m->h[r], count, B, x + first, udes + first, uact + first, relax + first,
			                             rc + first, m->s[r]);
		});
	}
	for (std::thread &t : th) t.join();
	for (int r = 0; r < n; r++)
		if (res[r]) return res[r];
	return ASIF_HIP_OK;
}
