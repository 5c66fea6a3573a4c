// asif_filter.cpp -- host side of the explicit CBF filter (see include/asif_filter.h).
//
// QP handed to the solver (src/asif.cpp:84-98, 295-303, 314-325):
//   variables (u, delta); H = diag(I_nu, relaxCost); c = (-2 uDes, -2 relaxCost relaxLb)
//   rows  Lgh u + h delta >= -Lfh ;  lb_u <= u <= ub_u ;  delta pinned to relaxLb.
#include "asif_filter.h"
#include <algorithm>
#include <cstring>
#include <numeric>

namespace ASIF {

ASIF::ASIF(const uint32_t nx, const uint32_t nu, const uint32_t npSS, SafetySetFn safetySet, DynamicsFn dynamics,
           const uint32_t npSSmax, const QPSOLVER qpSolverType, const bool diagonalCost)
    : nx_(nx), nu_(nu), nv_(nu + 1), npSS_(npSS), npSSmax_(std::min(npSSmax, npSS)), nc_(npSSmax_),
      safetySet_(safetySet), dynamics_(dynamics), options_(), QPsolver_(nullptr), H_(nv_ * nv_, 0.0), c_(nv_, 0.0),
      A_(nc_ * nv_, 0.0), b_(nc_, 0.0), lb_(nv_, 0.0), ub_(nv_, 0.0), LfhUser_(nullptr), LghUser_(nullptr),
      batch_(nullptr), boundModel_(-1)
{
	QPsolver_ = makeQPWrapper(qpSolverType, nv_, nc_, diagonalCost); // src/asif.cpp:36-48
}

ASIF::~ASIF(void)
{
	delete QPsolver_;
	if (batch_) asif_hip_multi_destroy(batch_);
}

int32_t ASIF::initialize(const double lb[], const double ub[])
{
	for (uint32_t j = 0; j < nu_; j++) H_[j + j * nv_] = 1.0;
	H_[nu_ + nu_ * nv_] = options_.relaxCost;
	std::copy(lb, lb + nu_, lb_.begin());
	std::copy(ub, ub + nu_, ub_.begin());
	lb_[nu_] = options_.relaxLb;
	ub_[nu_] = options_.relaxLb; // the relaxation variable is pinned (src/asif.cpp:88-91)
	const std::vector<double> origin(nx_, 0.0), zeroInput(nu_, 0.0);
	updateConstraints(origin.data());
	c_[nu_] = -2.0 * options_.relaxCost * options_.relaxLb;
	updateCost(zeroInput.data());
	const int32_t r = QPsolver_->initialize(H_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data());
	if (r != 0) return r; // unlike the reference, a solver that cannot start is reported
	return 1;
}

int32_t ASIF::initialize(const double lb[], const double ub[], const Options &options)
{
	options_ = options;
	return initialize(lb, ub);
}

int32_t ASIF::filter(const double x[], const double uDes[], double uAct[])
{
	double relax;
	return filter(x, uDes, uAct, relax);
}

int32_t ASIF::filter(const double x[], const double uDes[], double uAct[], double Lfh[], double Lgh[])
{
	double relax;
	return filter(x, uDes, uAct, Lfh, Lgh, relax);
}

int32_t ASIF::filter(const double x[], const double uDes[], double uAct[], double &relax)
{
	updateCost(uDes);
	return filter(x, nullptr, c_.data(), uAct, relax);
}

int32_t ASIF::filter(const double x[], const double uDes[], double uAct[], double Lfh[], double Lgh[], double &relax)
{
	LfhUser_ = Lfh; // sticky, like use_custom_ineq_ in the reference
	LghUser_ = Lgh;
	updateCost(uDes);
	return filter(x, nullptr, c_.data(), uAct, relax);
}

int32_t ASIF::filter(const double x[], const double H[], const double c[], double uAct[])
{
	double relax;
	return filter(x, H, c, uAct, relax);
}

int32_t ASIF::filter(const double x[], const double H[], const double c[], double uAct[], double &relax)
{
	updateConstraints(x);
	if (H == nullptr) QPsolver_->updateCost(nullptr, c);
	else {
		updateH(H);
		QPsolver_->updateCost(H_.data(), c);
	}
	QPsolver_->updateA(A_.data());
	QPsolver_->updateb(b_.data());
	const int32_t status = QPsolver_->solve();
	if (status != (int32_t)QPWrapperAbstract::SOLVER_STATUS::FEASIBLE) return -1; // uAct, relax untouched
	std::vector<double> sol(nv_);
	QPsolver_->getSolution(sol.data());
	std::copy(sol.begin(), sol.begin() + nu_, uAct);
	inputSaturate(uAct);
	relax = sol[nu_];
	return 1;
}

int32_t ASIF::updateOptions(const Options &options)
{
	options_ = options;
	return updateOptions();
}

int32_t ASIF::updateOptions(void)
{
	// src/asif.cpp:223-231: cost and LOWER bound of the relaxation variable move, the upper bound does not
	H_[nu_ + nu_ * nv_] = options_.relaxCost;
	c_[nu_] = -2.0 * options_.relaxCost * options_.relaxLb;
	lb_[nu_] = options_.relaxLb;
	QPsolver_->updateBounds(lb_.data(), nullptr);
	QPsolver_->updateCost(H_.data(), c_.data());
	if (batch_) {
		asif_hip_options o;
		// failures come back as negative library codes; success is the reference's 1, so a positive code (a hipError_t
		// passed through) must not be handed on as it is
		if (int r = deviceOptions(boundModel_, &o)) return r > 0 ? ASIF_HIP_EINVAL : r;
		if (int r = asif_hip_multi_update_options(batch_, &o)) return r > 0 ? ASIF_HIP_EINVAL : r;
	}
	return 1;
}

int32_t ASIF::updateConstraints(const double x[])
{
	std::vector<double> hAll(npSS_), DhAll(npSS_ * nx_), f(nx_), g(nx_ * nu_);
	safetySet_(x, hAll.data(), DhAll.data());
	dynamics_(x, f.data(), g.data());
	// keep the npSSmax smallest margins (src/asif.cpp:250-268); identity when npSSmax == npSS
	std::vector<uint32_t> pick(npSS_);
	std::iota(pick.begin(), pick.end(), 0u);
	if (npSSmax_ < npSS_)
		std::stable_sort(pick.begin(), pick.end(), [&hAll](uint32_t a, uint32_t b) { return hAll[a] < hAll[b]; });
	for (uint32_t r = 0; r < npSSmax_; r++) {
		const uint32_t s = pick[r];
		double Lf = 0.0;
		for (uint32_t k = 0; k < nx_; k++) Lf = Lf + DhAll[s + k * npSS_] * f[k];
		for (uint32_t j = 0; j < nu_; j++) {
			double Lg = 0.0;
			for (uint32_t k = 0; k < nx_; k++) Lg = Lg + DhAll[s + k * npSS_] * g[k + j * nx_];
			A_[r + j * nc_] = LghUser_ ? LghUser_[r + j * npSSmax_] : Lg;
		}
		A_[r + nu_ * nc_] = hAll[s];
		b_[r] = -(LfhUser_ ? LfhUser_[r] : Lf);
	}
	return 1;
}

int32_t ASIF::updateCost(const double uDes[])
{
	for (uint32_t j = 0; j < nu_; j++) c_[j] = -2.0 * uDes[j];
	return 1;
}

int32_t ASIF::updateH(const double H[])
{
	for (uint32_t j = 0; j < nu_; j++)
		for (uint32_t i = 0; i < nu_; i++) H_[i + j * nv_] = H[i + j * nu_];
	return 1;
}

void ASIF::inputSaturate(double u[])
{
	for (uint32_t j = 0; j < nu_; j++) u[j] = std::min(std::max(u[j], lb_[j]), ub_[j]);
}

int32_t ASIF::bindDeviceModel(int model, int device)
{
	const int32_t devs[1] = {device};
	return bindDeviceModel(model, 1, devs);
}

// several GPUs: filterBatch() cuts the batch into contiguous blocks, one per entry of devs (asif_hip_create_multi)
int32_t ASIF::bindDeviceModel(int model, int32_t ndev, const int32_t devs[])
{
	if (batch_) {
		asif_hip_multi_destroy(batch_);
		batch_ = nullptr;
	}
	asif_hip_options o;
	int r = deviceOptions(model, &o);
	if (r) return r;
	r = asif_hip_create_multi(&batch_, model, ASIF_HIP_EXPLICIT, &o, nullptr, ndev, devs);
	if (r) return r;
	asif_hip_dims d;
	asif_hip_get_dims(asif_hip_multi_handle(batch_, 0), &d);
	if ((uint32_t)d.nx != nx_ || (uint32_t)d.nu != nu_ || (uint32_t)d.nc != nc_) {
		asif_hip_multi_destroy(batch_);
		batch_ = nullptr;
		return ASIF_HIP_EINVAL;
	}
	boundModel_ = model;
	return 0;
}

// the device-side options of THIS object for a compiled model: the class's options, its input bounds and its row
// budget (npSSmax: the device keeps the same rows per call, src/asif.cpp:250-268); used when binding and whenever
// updateOptions() runs afterwards, so both describe the same filter
int32_t ASIF::deviceOptions(int model, asif_hip_options *o) const
{
	const int r = asif_hip_default_options(model, ASIF_HIP_EXPLICIT, o);
	if (r) return r;
	o->relaxCost = options_.relaxCost;
	o->relaxLb = options_.relaxLb;
	o->inf = options_.inf;
	o->satSharpness = options_.satSharpness;
	o->npSSmax = (int32_t)npSSmax_;
	for (uint32_t j = 0; j < nu_ && j < ASIF_HIP_MAX_NU; j++) {
		o->lb[j] = lb_[j];
		o->ub[j] = ub_[j];
	}
	return 0;
}

int32_t ASIF::filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[], int32_t rc[])
{
	if (!batch_) return ASIF_HIP_EINVAL;
	return asif_hip_filter_batch_host_multi(batch_, B, x, uDes, uAct, relax, rc);
}

} // namespace ASIF
