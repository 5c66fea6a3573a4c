// asif_realizable_filter.cpp -- host side of ASIF::ASIFrealizable (see the header).
// Variables of the reference's QP: x = (u, then per row group s: l+_s[0..nu], l-_s[0..nu], last: delta);
// per group three kinds of rows (ineq on the interval ends, nu equalities tying u to l+ - l-, one equality
// l+_nu - l-_nu = 1), then npSSmax barrier rows  Lgh u + delta >= -Lfh - relaxDes (h - relaxOffset).
#include "asif_realizable_filter.h"
#include <algorithm>
#include <numeric>

namespace ASIF {

// src/asif_realizable.cpp:5-75
ASIFrealizable::ASIFrealizable(const uint32_t nx, const uint32_t nu, const double uncertaintyBounds[],
                               const kernel_t &kernel, DynamicsFn dynamics, const uint32_t npSSmax, const QPSOLVER qpSolverType,
                               const bool diagonalCost)
    : nx_(nx), nu_(nu), uncertaintyBounds_(uncertaintyBounds, uncertaintyBounds + nx), kernel_(kernel),
      nFacets_((uint32_t)kernel.facets.size()), npSS_(kernel.maxCriticalFacets * kernel.maxActiveConstraints),
      npSSmax_((npSSmax > nFacets_) ? nFacets_ : npSSmax),
      nv_((npSSmax_ > 0) ? (nu + npSS_ * 2 * (nu + 1) + 1) : (nu + npSS_ * 2 * (nu + 1))),
      nc_(npSS_ * (nu + 2) + npSSmax_), dynamics_(dynamics), options_(),
      QPsolver_(makeQPWrapper(qpSolverType, nv_, nc_, diagonalCost)), // the lifted problem itself, src/asif_realizable.cpp:48-53
      facetSolver_(makeQPWrapper(qpSolverType, nx, 2 * nx + 1, true)), H_(nv_ * nv_, 0.0), c_(nv_, 0.0), A_(nc_ * nv_, 0.0),
      b_(nc_, 0.0), lb_(nv_, 0.0), ub_(nv_, 0.0), A_facet_((2 * nx + 1) * nx, 0.0), b_facet_(2 * nx + 1, 0.0),
      batch_(nullptr), criticalFacets_(kernel.maxCriticalFacets), nCriticalFacets_(0)
{
	// facetSolver_ only answers "feasible or not" (:428-429).  When the facet grazes the uncertainty box its two
	// active rows are nearly parallel and the iterates creep along them; feasibility to 1e-6 is decided long
	// before the minimiser is resolved to 1e-8 (the reference runs OSQP at 1e-3 here).
	// (QPSOLVER::HOST decides a facet of up to three states exactly, by the active-set method; beyond that at 1e-6 too.)
	if (QPWrapperHip *fs = dynamic_cast<QPWrapperHip *>(facetSolver_)) {
		fs->settings.eps_abs = 1e-6;
		fs->settings.eps_rel = 1e-6;
		fs->settings.max_iter = 20000;
	} else if (QPWrapperHost *fh = dynamic_cast<QPWrapperHost *>(facetSolver_)) {
		fh->epsRel = 1e-6;
	}
	if (npSSmax_ > 0) {
		criticalBarrierFacets_.resize(npSSmax_);
		hBarrier_.resize(npSSmax_);
		DhBarrier_.resize(npSSmax_ * nx_);
	}
}

ASIFrealizable::~ASIFrealizable(void)
{
	delete QPsolver_;
	delete facetSolver_;
	if (batch_) asif_hip_destroy(batch_);
}

// src/asif_realizable.cpp:100-267
int32_t ASIFrealizable::initialize(const double lb[], const double ub[])
{
	if (nu_ != 1) return ASIF_HIP_EUNSUPPORTED; // the reference's row structure is only consistent for nu == 1 (:214-220)
	// facet feasibility problem (:113-135): min |lam|^2, sum lam = 1, V lam >= x - unc, -V lam >= -x - unc, 0 <= lam <= 1
	std::vector<double> Hf(nx_ * nx_, 0.0), cf(nx_, 0.0), lbf(nx_, 0.0), ubf(nx_, 1.0);
	for (uint32_t j = 0; j < nx_; j++) Hf[j + j * nx_] = 1.0;
	bool *bef = new bool[2 * nx_ + 1]();
	bef[0] = true;
	std::fill(A_facet_.begin(), A_facet_.end(), 0.0);
	std::fill(b_facet_.begin(), b_facet_.end(), 0.0);
	for (uint32_t i = 0; i < nx_; i++) A_facet_[i * (2 * nx_ + 1)] = 1.0;
	b_facet_[0] = 1.0;
	int32_t r = facetSolver_->initialize(Hf.data(), cf.data(), A_facet_.data(), b_facet_.data(), lbf.data(), ubf.data(), bef);
	delete[] bef;
	if (r != 0) return r;

	const std::vector<std::vector<double>> &vertices = kernel_.vertices;
	for (uint32_t i = 0; i < nFacets_; i++) { // xFaceInt, :137-157
		std::vector<interval_t> &xFaceInt = kernel_.facets[i].xFaceInt;
		xFaceInt.resize(nx_);
		const std::vector<uint32_t> &verticesIdx = kernel_.facets[i].verticesIdx;
		for (uint32_t j = 0; j < nx_; j++) xFaceInt[j] = vertices[verticesIdx[0]][j];
		for (uint32_t j = 1; j <= nx_ - 1; j++) {
			interval_t lamdaCurr = interval_t(0., 1.);
			const std::vector<double> &vertex = vertices[verticesIdx[j]];
			for (uint32_t k = 0; k < nx_; k++) xFaceInt[k] = lamdaCurr * xFaceInt[k] + (1. - lamdaCurr) * vertex[k];
		}
	}
	for (uint32_t i = 0; i < nFacets_; i++) { // bounding boxes, :160-175
		std::vector<std::pair<double, double>> &boundingBox = kernel_.facets[i].boundingBox;
		boundingBox.resize(nx_);
		const std::vector<uint32_t> &verticesIdx = kernel_.facets[i].verticesIdx;
		for (uint32_t j = 0; j < nx_; j++) {
			double lo = vertices[verticesIdx[0]][j], hi = lo;
			for (uint32_t k = 1; k < nx_; k++) {
				lo = std::min(lo, vertices[verticesIdx[k]][j]);
				hi = std::max(hi, vertices[verticesIdx[k]][j]);
			}
			boundingBox[j] = std::make_pair(lo, hi);
		}
	}
	// bounds (:186-193) and the fixed part of A, b (:196-238)
	for (uint32_t j = 0; j < nu_; j++) {
		lb_[j] = lb[j];
		ub_[j] = ub[j];
	}
	for (uint32_t i = nu_; i < nv_; i++) {
		lb_[i] = 0.0;
		ub_[i] = options_.inf;
	}
	std::fill(A_.begin(), A_.end(), 0.0);
	std::fill(b_.begin(), b_.end(), 0.0);
	uint32_t col = nu_;
	for (uint32_t row = 0; row < npSS_ * (nu_ + 2); row += nu_ + 2) {
		for (uint32_t i = 0; i < nu_; i++)
			for (uint32_t j = 0; j < nu_; j++) A_[(row + 1 + i) + j * nc_] = -1.0;
		for (uint32_t i = 0; i < nu_ + 1; i++) {
			A_[(row + 1 + i) + (col + i) * nc_] = 1.0;
			A_[(row + 1 + i) + (col + nu_ + 1 + i) * nc_] = -1.0;
		}
		b_[row + nu_ + 1] = 1.0;
		col += 2 * (nu_ + 1);
	}
	// H = diag(I_nu, 0 ... 0, relaxCost) (:176-189)
	std::fill(H_.begin(), H_.end(), 0.0);
	for (uint32_t j = 0; j < nu_; j++) H_[j + j * nv_] = 1.0;
	if (npSSmax_ > 0) H_[nv_ * nv_ - 1] = options_.relaxCost;
	// rows of the groups: equalities except the first of each group (:233-236)
	bool *be = new bool[nc_]();
	for (uint32_t i = 0; i < npSS_ * (nu_ + 2); i++) be[i] = (i % (nu_ + 2)) != 0;
	const std::vector<double> origin(nx_, 0.0);
	updateConstraints(origin.data());
	std::fill(c_.begin(), c_.end(), 0.0);
	r = QPsolver_->initialize(H_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data(), be);
	delete[] be;
	return r != 0 ? r : 1;
}

int32_t ASIFrealizable::initialize(const double lb[], const double ub[], const Options &options)
{
	options_ = options;
	return initialize(lb, ub);
}

int32_t ASIFrealizable::filter(const double x[], const double uDes[], double uAct[])
{
	double relax[2];
	return filter(x, uDes, uAct, relax);
}

int32_t ASIFrealizable::filter(const double x[], const double uDes[], double uAct[], double relax[2])
{
	for (uint32_t i = 0; i < nu_; i++) c_[i] = -2.0 * uDes[i]; // updateCost, :686-697
	return filter(x, nullptr, c_.data(), uAct, relax);
}

int32_t ASIFrealizable::filter(const double x[], const double H[], const double c[], double uAct[])
{
	double relax[2];
	return filter(x, H, c, uAct, relax);
}

// src/asif_realizable.cpp:311-352; c has nv entries like the reference's c_
int32_t ASIFrealizable::filter(const double x[], const double H[], const double c[], double uAct[], double relax[2])
{
	if (updateConstraints(x) < 0) return -2;
	if (H == nullptr) QPsolver_->updateCost(nullptr, c);
	else {
		for (uint32_t j = 0; j < nu_; j++) // updateH, :699-717: the nu x nu block
			for (uint32_t i = 0; i < nu_; i++) H_[i + j * nv_] = H[i + j * nu_];
		QPsolver_->updateCost(H_.data(), c);
	}
	QPsolver_->updateA(A_.data());
	QPsolver_->updateb(b_.data());
	const int32_t status = QPsolver_->solve();
	if (status != (int32_t)QPWrapperAbstract::SOLVER_STATUS::FEASIBLE) return -1;
	std::vector<double> sol(nv_);
	QPsolver_->getSolution(sol.data());
	for (uint32_t j = 0; j < nu_; j++) uAct[j] = std::min(std::max(sol[j], lb_[j]), ub_[j]);
	relax[0] = sol[nu_]; // a multiplier (l+_0 of group 0): the cost is zero on it, any feasible value is optimal (:346)
	relax[1] = sol[nv_ - 1];
	return 1;
}

int32_t ASIFrealizable::updateOptions(const Options &options)
{
	options_ = options;
	return updateOptions();
}

// src/asif_realizable.cpp:365-373
int32_t ASIFrealizable::updateOptions(void)
{
	if (npSSmax_ > 0) H_[nv_ * nv_ - 1] = options_.relaxCost;
	QPsolver_->updateBounds(lb_.data(), nullptr);
	QPsolver_->updateCost(H_.data(), c_.data());
	if (batch_) {
		asif_hip_realizable_options o = batchOpts();
		return asif_hip_update_realizable_options(batch_, &o) == 0 ? 1 : -1;
	}
	return 1;
}

// src/asif_realizable.cpp:375-610
int32_t ASIFrealizable::updateConstraints(const double x[])
{
	nCriticalFacets_ = 0;
	for (uint32_t i = 0; i < nx_; i++) {
		b_facet_[1 + i] = x[i] - uncertaintyBounds_[i];
		b_facet_[1 + i + nx_] = -x[i] - uncertaintyBounds_[i];
	}
	facetSolver_->updateb(b_facet_.data());
	std::vector<double> hFull(nFacets_);
	for (uint32_t i = 0; i < nFacets_; i++) {
		const std::vector<double> &normal = kernel_.facets[i].normal;
		hFull[i] = 1.;
		for (uint32_t j = 0; j < nx_; j++) hFull[i] -= normal[j] * x[j];
	}
	const std::vector<std::vector<double>> &vertices = kernel_.vertices;
	for (uint32_t i = 0; i < nFacets_; i++) {
		const std::vector<uint32_t> &verticesIdx = kernel_.facets[i].verticesIdx;
		const std::vector<std::pair<double, double>> &boundingBox = kernel_.facets[i].boundingBox;
		bool potentialFacet = true;
		for (uint32_t j = 0; j < nx_; j++)
			if (x[j] < (boundingBox[j].first - uncertaintyBounds_[j]) ||
			    x[j] > (boundingBox[j].second + uncertaintyBounds_[j])) {
				potentialFacet = false;
				break;
			}
		if (!potentialFacet) continue;
		for (uint32_t j = 0; j < nx_; j++) {
			const std::vector<double> &vertex = vertices[verticesIdx[j]];
			for (uint32_t k = 0; k < nx_; k++) {
				A_facet_[1 + k + (j * (2 * nx_ + 1))] = vertex[k];
				A_facet_[1 + k + nx_ + (j * (2 * nx_ + 1))] = -vertex[k];
			}
		}
		facetSolver_->updateA(A_facet_.data());
		if (facetSolver_->solve() == (int32_t)QPWrapperAbstract::SOLVER_STATUS::FEASIBLE) {
			criticalFacets_[nCriticalFacets_] = i;
			nCriticalFacets_++;
			if (nCriticalFacets_ >= kernel_.maxCriticalFacets) break;
		}
	}

	// interval Lie derivatives of the active constraints of every critical facet, :445-506
	std::vector<interval_t> Lfh(npSS_, interval_t(0.)), Lgh(npSS_ * nu_, interval_t(0.));
	if (nCriticalFacets_ > 0) {
		uint32_t total = 0;
		std::vector<uint32_t> facetOf, srcOf;
		for (uint32_t i = 0; i < nCriticalFacets_; i++) {
			const std::vector<uint32_t> &act = kernel_.facets[criticalFacets_[i]].activeConstraintsSet;
			for (uint32_t j = 0; j < act.size(); j++) {
				facetOf.push_back(criticalFacets_[i]);
				srcOf.push_back(act[j]);
				total++;
			}
		}
		std::vector<interval_t> DhInt(total * nx_); // symbols created row by row, column inside (:483-487)
		for (uint32_t i = 0; i < total; i++)
			for (uint32_t j = 0; j < nx_; j++) DhInt[i + j * total] = interval(-kernel_.facets[srcOf[i]].normal[j]);
		for (uint32_t i = 0; i < total; i++) {
			const std::vector<interval_t> &xFaceInt = kernel_.facets[facetOf[i]].xFaceInt;
			std::vector<interval_t> f(nx_), g(nx_ * nu_);
			dynamics_(xFaceInt.data(), f.data(), g.data());
			Lfh[i] = 0.;
			for (uint32_t j = 0; j < nx_; j++) Lfh[i] = Lfh[i] + f[j] * DhInt[i + (j * total)];
			for (uint32_t j = 0; j < nu_; j++) {
				Lgh[i + j * npSS_] = 0.;
				for (uint32_t k = 0; k < nx_; k++) Lgh[i + j * npSS_] = Lgh[i + j * npSS_] + g[k + j * nx_] * DhInt[i + (k * total)];
			}
		}
	}
	// rows of the critical groups, :509-527
	uint32_t col = nu_, s = 0;
	for (uint32_t row = 0; row < npSS_ * (nu_ + 2); row += nu_ + 2, s++) {
		const interval tg = Lgh[s].convert(), tf = Lfh[s].convert();
		A_[row + (col + 0) * nc_] = tg.left();
		A_[row + (col + (nu_ + 1) + 0) * nc_] = -tg.right();
		A_[row + (col + nu_) * nc_] = tf.left();
		A_[row + (col + (nu_ + 1) + nu_) * nc_] = -tf.right();
		col += 2 * (nu_ + 1);
	}
	// barrier rows, :530-600
	if (npSSmax_ > 0) {
		std::vector<interval_t> xInt(nx_), fInt(nx_), gInt(nx_ * nu_);
		for (uint32_t i = 0; i < nx_; i++) xInt[i] = interval(x[i]);
		dynamics_(xInt.data(), fInt.data(), gInt.data());
		std::vector<double> f(nx_), g(nx_ * nu_);
		for (uint32_t i = 0; i < nx_; i++) {
			f[i] = fInt[i].convert().mid();
			for (uint32_t j = 0; j < nu_; j++) g[i + j * nx_] = gInt[i + j * nx_].convert().mid();
		}
		std::vector<uint32_t> order(nFacets_);
		std::iota(order.begin(), order.end(), 0u);
		if (npSSmax_ < nFacets_)
			std::stable_sort(order.begin(), order.end(), [&hFull](uint32_t a, uint32_t b) { return hFull[a] < hFull[b]; });
		for (uint32_t i = 0; i < npSSmax_; i++) {
			const uint32_t fi = order[i];
			criticalBarrierFacets_[i] = fi;
			hBarrier_[i] = hFull[fi];
			double lfh = 0.0, lgh = 0.0; // nu == 1
			for (uint32_t k = 0; k < nx_; k++) {
				DhBarrier_[i + k * npSSmax_] = -kernel_.facets[fi].normal[k];
				lfh += DhBarrier_[i + k * npSSmax_] * f[k];
			}
			for (uint32_t k = 0; k < nx_; k++) lgh += DhBarrier_[i + k * npSSmax_] * g[k];
			const uint32_t row = npSS_ * (nu_ + 2) + i;
			A_[row + 0 * nc_] = lgh;
			A_[row + (nv_ - 1) * nc_] = 1.0;
			b_[row] = -lfh - options_.relaxDes * (hBarrier_[i] - options_.relaxOffset);
		}
	}
	if (nCriticalFacets_ == 0 && std::any_of(hFull.begin(), hFull.end(), [](double v) { return v < 0.; })) return -1;
	return 1;
}

asif_hip_realizable_options ASIFrealizable::batchOpts(void) const
{
	asif_hip_realizable_options o = batchModel_;
	o.relaxDes = options_.relaxDes;
	o.relaxOffset = options_.relaxOffset;
	o.relaxCost = options_.relaxCost;
	o.inf = options_.inf;
	o.lb[0] = lb_[0];
	o.ub[0] = ub_[0];
	for (uint32_t i = 0; i < nx_ && i < 4; i++) o.uncertaintyBounds[i] = uncertaintyBounds_[i];
	o.npSSmax = (int32_t)npSSmax_;
	return o;
}

int32_t ASIFrealizable::bindDeviceModel(int model, const asif_hip_realizable_options &modelData, int device)
{
	if (batch_) asif_hip_destroy(batch_);
	batch_ = nullptr;
	batchModel_ = modelData;
	const asif_hip_realizable_options o = batchOpts();
	const uint32_t nA = kernel_.maxActiveConstraints;
	std::vector<double> v(kernel_.vertices.size() * nx_), n(nFacets_ * nx_);
	std::vector<int32_t> fv(nFacets_ * nx_), fa(nFacets_ * nA);
	for (size_t i = 0; i < kernel_.vertices.size(); i++)
		for (uint32_t k = 0; k < nx_; k++) v[i * nx_ + k] = kernel_.vertices[i][k];
	for (uint32_t i = 0; i < nFacets_; i++) {
		const facet_t &f = kernel_.facets[i];
		if (f.activeConstraintsSet.size() != nA) return ASIF_HIP_EINVAL;
		for (uint32_t k = 0; k < nx_; k++) {
			n[i * nx_ + k] = f.normal[k];
			fv[i * nx_ + k] = (int32_t)f.verticesIdx[k];
		}
		for (uint32_t j = 0; j < nA; j++) fa[i * nA + j] = (int32_t)f.activeConstraintsSet[j];
	}
	asif_hip_kernel_data k = {(int32_t)nx_, (int32_t)kernel_.vertices.size(), (int32_t)nFacets_,
	                          (int32_t)kernel_.maxCriticalFacets, (int32_t)nA, v.data(), fv.data(), n.data(), fa.data()};
	int r = asif_hip_create_realizable(&batch_, model, &k, &o, nullptr, device);
	if (r) return r;
	asif_hip_dims d;
	asif_hip_get_dims(batch_, &d);
	if ((uint32_t)d.nv != nv_ || (uint32_t)d.nc != nc_) {
		asif_hip_destroy(batch_);
		batch_ = nullptr;
		return ASIF_HIP_EINVAL;
	}
	return 0;
}

int32_t ASIFrealizable::filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[],
                                    int32_t rc[])
{
	if (!batch_) return ASIF_HIP_EINVAL;
	return asif_hip_filter_batch_host(batch_, B, x, uDes, uAct, relax, rc);
}

} // namespace ASIF
