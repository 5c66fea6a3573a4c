// asif_robust_filter.cpp -- host side of ASIF::ASIFrobust (see the header).
// Variables x = (u, delta, then per safety row s: l+_s[0..nu], l-_s[0..nu]); per s three kinds of rows
// (SURVEY Appendix A): the interval inequality, nu equalities tying u to l+ - l-, one equality l+_nu - l-_nu = 1.
#include "asif_robust_filter.h"
#include <algorithm>
#include <numeric>

namespace ASIF {

ASIFrobust::ASIFrobust(const uint32_t nx, const uint32_t nu, const uint32_t npSS, SafetySetFn safetySet,
                       DynamicsFn dynamics, const uint32_t npSSmax, const QPSOLVER qpSolverType, const bool diagonalCost)
    : nx_(nx), nu_(nu), npSS_(npSS), npSSmax_(std::min(npSSmax, npSS)), nv_(nu + 1 + npSSmax_ * 2 * (nu + 1)),
      nc_(npSSmax_ * (1 + (nu + 1))), safetySet_(safetySet), dynamics_(dynamics), options_(),
      QPsolver_(makeQPWrapper(qpSolverType, nv_, nc_, diagonalCost)), H_(nv_ * nv_, 0.0), c_(nv_, 0.0), A_(nc_ * nv_, 0.0),
      b_(nc_, 0.0), lb_(nv_, 0.0), ub_(nv_, 0.0), batch_(nullptr)
{
}

ASIFrobust::~ASIFrobust(void)
{
	delete QPsolver_;
	if (batch_) asif_hip_destroy(batch_);
}

// src/asif_robust.cpp:64-181
int32_t ASIFrobust::initialize(const double lb[], const double ub[])
{
	for (uint32_t j = 0; j < nu_; j++) {
		H_[j + j * nv_] = 1.0;
		lb_[j] = lb[j];
		ub_[j] = ub[j];
	}
	H_[nu_ + nu_ * nv_] = options_.relaxCost;
	lb_[nu_] = options_.relaxLb;
	ub_[nu_] = options_.inf;
	for (uint32_t i = nu_ + 1; i < nv_; i++) {
		lb_[i] = 0.0;
		ub_[i] = options_.inf;
	}
	// fixed sparsity pattern (:103-133); note the full -1 block on (i<nu, j<nu): exact for nu == 1 only
	std::fill(A_.begin(), A_.end(), 0.0);
	std::fill(b_.begin(), b_.end(), 0.0);
	uint32_t col = nu_ + 1;
	for (uint32_t row = 0; row < nc_; row += nu_ + 2) {
		for (uint32_t i = 0; i < nu_; i++)
			for (uint32_t j = 0; j < nu_; j++) A_[(row + 1 + i) + j * nc_] = -1.0;
		for (uint32_t i = 0; i < nu_ + 1; i++) {
			A_[(row + 1 + i) + (col + i) * nc_] = 1.0;
			A_[(row + 1 + i) + (col + nu_ + 1 + i) * nc_] = -1.0;
		}
		b_[row + nu_ + 1] = 1.0;
		col += 2 * (nu_ + 1);
	}
	const std::vector<double> origin(nx_, 0.0);
	updateConstraints(origin.data());
	for (uint32_t j = 0; j < nu_; j++) c_[j] = -0.0;
	c_[nu_] = -2.0 * options_.relaxCost * options_.relaxLb;
	bool *be = new bool[nc_]; // every row but the interval inequality of each group is an equality, :145-148
	for (uint32_t i = 0; i < nc_; i++) be[i] = (i % (nu_ + 2)) != 0;
	const int32_t r = QPsolver_->initialize(H_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data(), be);
	delete[] be;
	return r != 0 ? r : 1;
}

int32_t ASIFrobust::initialize(const double lb[], const double ub[], const Options &options)
{
	options_ = options;
	return initialize(lb, ub);
}

int32_t ASIFrobust::filter(const double x[], const double uDes[], double uAct[])
{
	double relax;
	return filter(x, uDes, uAct, relax);
}

int32_t ASIFrobust::filter(const double x[], const double uDes[], double uAct[], double &relax)
{
	for (uint32_t j = 0; j < nu_; j++) c_[j] = -2.0 * uDes[j];
	return filter(x, nullptr, c_.data(), uAct, relax);
}

int32_t ASIFrobust::filter(const double x[], const double H[], const double c[], double uAct[])
{
	double relax;
	return filter(x, H, c, uAct, relax);
}

int32_t ASIFrobust::filter(const double x[], const double H[], const double c[], double uAct[], double &relax)
{
	updateConstraints(x);
	if (H == nullptr) QPsolver_->updateCost(nullptr, c);
	else {
		for (uint32_t j = 0; j < nu_; j++)
			for (uint32_t i = 0; i < nu_; i++) H_[i + j * nv_] = H[i + j * nu_];
		QPsolver_->updateCost(H_.data(), c);
	}
	QPsolver_->updateA(A_.data());
	QPsolver_->updateb(b_.data());
	const int32_t status = QPsolver_->solve();
	if (status != (int32_t)QPWrapperAbstract::SOLVER_STATUS::FEASIBLE) return -1; // :250-251
	std::vector<double> sol(nv_);
	QPsolver_->getSolution(sol.data());
	for (uint32_t j = 0; j < nu_; j++) uAct[j] = std::min(std::max(sol[j], lb_[j]), ub_[j]);
	relax = sol[nu_];
	return 1;
}

int32_t ASIFrobust::updateOptions(const Options &options)
{
	options_ = options;
	return updateOptions();
}

int32_t ASIFrobust::updateOptions(void)
{
	H_[nu_ + nu_ * nv_] = options_.relaxCost;
	c_[nu_] = -2.0 * options_.relaxCost * options_.relaxLb;
	lb_[nu_] = options_.relaxLb;
	QPsolver_->updateBounds(lb_.data(), nullptr);
	QPsolver_->updateCost(H_.data(), c_.data());
	if (batch_ && batchIsData_) {
		batchDataOpts_.relaxCost = options_.relaxCost;
		batchDataOpts_.relaxLb = options_.relaxLb;
		batchDataOpts_.inf = options_.inf;
		asif_hip_update_robust_data_options(batch_, &batchDataOpts_);
	} else if (batch_) {
		batchOpts_.relaxCost = options_.relaxCost;
		batchOpts_.relaxLb = options_.relaxLb;
		batchOpts_.inf = options_.inf;
		asif_hip_update_options(batch_, &batchOpts_);
	}
	return 1;
}

// src/asif_robust.cpp:275-367
int32_t ASIFrobust::updateConstraints(const double x[])
{
	std::vector<interval_t> xI(nx_), f(nx_), g(nx_ * nu_);
	for (uint32_t i = 0; i < nx_; i++) xI[i] = interval(x[i]);
	std::vector<double> hAll(npSS_), DhAll(npSS_ * nx_);
	safetySet_(x, hAll.data(), DhAll.data());
	dynamics_(xI.data(), f.data(), g.data());
	std::vector<uint32_t> pick(npSS_);
	std::iota(pick.begin(), pick.end(), 0u);
	if (npSSmax_ < npSS_)
		std::stable_sort(pick.begin(), pick.end(), [&hAll](uint32_t a, uint32_t b) { return hAll[a] < hAll[b]; });
	std::vector<interval_t> DhI(npSSmax_ * nx_);
	for (uint32_t e = 0; e < npSSmax_ * nx_; e++) { // symbols created in column-major order, like the reference
		const uint32_t i = e % npSSmax_, j = e / npSSmax_;
		DhI[e] = interval(DhAll[pick[i] + j * npSS_]);
	}
	std::vector<interval_t> Lfh(npSSmax_), Lgh(npSSmax_ * nu_);
	for (uint32_t i = 0; i < npSSmax_; i++) {
		Lfh[i] = 0.0;
		for (uint32_t k = 0; k < nx_; k++) Lfh[i] = Lfh[i] + DhI[i + k * npSSmax_] * f[k];
	}
	for (uint32_t i = 0; i < npSSmax_; i++)
		for (uint32_t j = 0; j < nu_; j++) {
			Lgh[i + j * npSSmax_] = 0.0;
			for (uint32_t k = 0; k < nx_; k++)
				Lgh[i + j * npSSmax_] = Lgh[i + j * npSSmax_] + DhI[i + k * npSSmax_] * g[k + j * nx_];
		}
	uint32_t col = nu_ + 1, s = 0;
	for (uint32_t row = 0; row < nc_; row += nu_ + 2, s++) {
		A_[row + nu_ * nc_] = hAll[pick[s]];
		for (uint32_t j = 0; j < nu_; j++) {
			const interval t = Lgh[s + j * npSSmax_].convert();
			A_[row + (col + j) * nc_] = t.left();
			A_[row + (col + (nu_ + 1) + j) * nc_] = -t.right();
		}
		const interval t = Lfh[s].convert();
		A_[row + (col + nu_) * nc_] = t.left();
		A_[row + (col + (nu_ + 1) + nu_) * nc_] = -t.right();
		col += 2 * (nu_ + 1);
	}
	return 1;
}

int32_t ASIFrobust::bindDeviceModel(int model, const asif_hip_options &modelData, int device)
{
	if (batch_) asif_hip_destroy(batch_);
	batch_ = nullptr;
	batchIsData_ = false;
	batchOpts_ = modelData;
	batchOpts_.relaxCost = options_.relaxCost;
	batchOpts_.relaxLb = options_.relaxLb;
	batchOpts_.inf = options_.inf;
	for (uint32_t j = 0; j < nu_ && j < ASIF_HIP_MAX_NU; j++) {
		batchOpts_.lb[j] = lb_[j];
		batchOpts_.ub[j] = ub_[j];
	}
	int r = asif_hip_create(&batch_, model, ASIF_HIP_ROBUST, &batchOpts_, nullptr, device);
	if (r) return r;
	asif_hip_dims d;
	asif_hip_get_dims(batch_, &d);
	if ((uint32_t)d.nx != nx_ || (uint32_t)d.nu != nu_ || (uint32_t)d.nc != nc_ || (uint32_t)d.nv != nv_) {
		asif_hip_destroy(batch_);
		batch_ = nullptr;
		return ASIF_HIP_EINVAL;
	}
	return 0;
}

int32_t ASIFrobust::bindDeviceData(int model, const double halfPlanes[], int32_t N,
                                   const asif_hip_robust_data_options &modelData, int device)
{
	if (batch_) asif_hip_destroy(batch_);
	batch_ = nullptr;
	if ((uint32_t)N != npSS_) return ASIF_HIP_EINVAL;
	asif_hip_robust_data_options o = modelData;
	o.relaxCost = options_.relaxCost;
	o.relaxLb = options_.relaxLb;
	o.inf = options_.inf;
	o.lb[0] = lb_[0];
	o.ub[0] = ub_[0];
	o.npSSmax = (int32_t)npSSmax_;
	int r = asif_hip_create_robust_data(&batch_, model, halfPlanes, N, &o, nullptr, device);
	if (r) return r;
	batchIsData_ = true;
	batchDataOpts_ = o;
	asif_hip_dims d;
	asif_hip_get_dims(batch_, &d);
	if ((uint32_t)d.nx != nx_ || (uint32_t)d.nu != nu_ || (uint32_t)d.nc != nc_ || (uint32_t)d.nv != nv_) {
		asif_hip_destroy(batch_);
		batch_ = nullptr;
		return ASIF_HIP_EINVAL;
	}
	return 0;
}

int32_t ASIFrobust::filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[],
                                int32_t rc[])
{
	if (!batch_) return ASIF_HIP_EINVAL;
	return asif_hip_filter_batch_host(batch_, B, x, uDes, uAct, relax, rc);
}

} // namespace ASIF
