// asif_backup_filters.cpp -- host side of ASIF::ASIFimplicit / ASIF::ASIFimplicitTB (see the header).
// Row and variable order of the QPs:
//   implicit: x = (u, d_safe, d_reach); safe row (k,i): [Lgh, h, 0]; backup row: [Lgh, 0, h_B]; b = -Lfh
//   TB:       x = (u, d_safe);          safe rows as above [Lgh, h]; TTS row [Lgh, 0], b = -Lfh - relaxTTS (T - t_hit);
//             orthogonality row [Lgh, 0], b = -Lfh - relaxMinOrtho (cos - minOrtho); trivial mode A = 0, b = -inf
#include "asif_backup_filters.h"
#include <algorithm>
#include <cmath>
#include <numeric>

namespace ASIF {

// ------------------------------------------------------------------------------------------------
BackupLoopHost::BackupLoopHost(bool hasGradient, uint32_t nx, uint32_t nu, uint32_t npSS, SetFn safetySet,
                               DynFn dynamics, DynGradFn dynamicsGradients, DynWithGradFn dynamicsWithGradient,
                               CtrlFn backupController)
    : hasGradient_(hasGradient), nx_(nx), nu_(nu), npSS_(npSS), safetySet_(safetySet), dynamics_(dynamics),
      dynamicsGradients_(dynamicsGradients), dynamicsWithGradient_(dynamicsWithGradient),
      backupController_(backupController), lbU_(nu, 0.0), ubU_(nu, 0.0), satSharpness_(0.1), holdDt_(0.0),
      holdStep_(0.0), tLastHold_(-1.0), uHold_(nu, 0.0), DuHold_(nu * nx, 0.0)
{
	if (hasGradient_) {
		// plain dynamics = the fused callback at zero input (src/asif_implicit.cpp:81-89)
		dynamics_ = [this](const double *x, double *f, double *g) {
			const std::vector<double> u0(nu_, 0.0);
			std::vector<double> scratch(nx_ * nx_, 0.0);
			dynamicsWithGradient_(x, u0.data(), f, g, scratch.data());
		};
	}
}

// bevelled smooth saturation (src/asif_implicit.cpp:682-737)
void BackupLoopHost::saturateSoft(const double u[], double uSat[], double DuSat[]) const
{
	const double r = satSharpness_;
	for (uint32_t i = 0; i < nu_; i++) {
		const double mi = lbU_[i], ma = ubU_[i], range = ma - mi, middle = (ma + mi) / 2;
		const double uc = 2 * (u[i] - middle) / range;
		const double bevelL = r * std::tan(M_PI / 8);
		const double start = 1 - std::cos(M_PI / 4) * bevelL, stop = 1 + bevelL, yc = 1 - r;
		uSat[i] = u[i];
		DuSat[i] = 1;
		if (uc >= stop) { uSat[i] = ma; DuSat[i] = 0; }
		else if (uc <= -stop) { uSat[i] = mi; DuSat[i] = 0; }
		else if (uc > start) {
			const double s = std::sqrt(r * r - (uc - stop) * (uc - stop));
			DuSat[i] = (stop - uc) / s;
			uSat[i] = 0.5 * (s + yc) * range + middle;
		} else if (uc < -start) {
			const double s = std::sqrt(r * r - (uc + stop) * (uc + stop));
			DuSat[i] = (stop + uc) / s;
			uSat[i] = 0.5 * (-s - yc) * range + middle;
		}
	}
}

void BackupLoopHost::saturateHard(double u[]) const
{
	for (uint32_t i = 0; i < nu_; i++) u[i] = std::min(std::max(u[i], lbU_[i]), ubU_[i]);
}

// src/asif_implicit.cpp:751-815; with holdDt_ > 0: src/asif_implicit_robust.cpp:878-953
void BackupLoopHost::closedLoop(const double x[], double fCL[], double DfCL[], double t) const
{
	std::vector<double> f(nx_), g(nx_ * nu_), u(nu_), Du(nu_ * nx_), uSat(nu_), DuSat(nu_);
	backupController_(x, u.data(), Du.data());
	const bool held = holdDt_ > 0;
	if (held) {
		if (t <= holdStep_) tLastHold_ = -1.; // first rhs of a trajectory resets the clock (:891-893)
		if (t >= (tLastHold_ + holdDt_ - 0.0001)) {
			uHold_ = u;
			DuHold_ = Du;
			tLastHold_ = t;
		}
	}
	saturateSoft(held ? uHold_.data() : u.data(), uSat.data(), DuSat.data());
	if (hasGradient_) {
		// the fused-gradient branch of the held class uses the HELD controller Jacobian (:913-919); the
		// separate branch below keeps the fresh one (:936-938) -- both as in the reference
		const std::vector<double> &DuJ = held ? DuHold_ : Du;
		std::vector<double> dfcl(nx_ * nx_);
		dynamicsWithGradient_(x, uSat.data(), f.data(), g.data(), dfcl.data());
		for (uint32_t i = 0; i < nx_; i++)
			for (uint32_t j = 0; j < nx_; j++) {
				double v = dfcl[i + j * nx_];
				for (uint32_t k = 0; k < nu_; k++) v += g[i + k * nx_] * DuSat[k] * DuJ[k + j * nu_];
				DfCL[i + j * nx_] = v;
			}
	} else {
		std::vector<double> Df(nx_ * nx_), Dg(nx_ * nu_ * nx_);
		dynamics_(x, f.data(), g.data());
		dynamicsGradients_(x, Df.data(), Dg.data());
		for (uint32_t i = 0; i < nx_; i++)
			for (uint32_t j = 0; j < nx_; j++) {
				double v = Df[i + j * nx_];
				for (uint32_t k = 0; k < nu_; k++)
					v += Dg[i + k * nx_ + j * nx_ * nu_] * uSat[k] + g[i + k * nx_] * DuSat[k] * Du[k + j * nu_];
				DfCL[i + j * nx_] = v;
			}
	}
	for (uint32_t i = 0; i < nx_; i++) {
		double v = 0.0;
		for (uint32_t k = 0; k < nu_; k++) v = v + g[i + k * nx_] * uSat[k];
		fCL[i] = v + f[i];
	}
}

// forward Euler on z = [x; vec Q], Q(0) = I (src/asif_implicit.cpp:417-425,461-484)
void BackupLoopHost::integrate(const double x[], uint32_t npBT, double dt)
{
	const uint32_t nz = nx_ + nx_ * nx_;
	traj_.assign(npBT, std::pair<double, state_t>(0.0, state_t(nz, 0.0)));
	hAll_.assign((size_t)npBT * npSS_, 0.0);
	DhAll_.assign((size_t)npBT * npSS_ * nx_, 0.0);
	hMin_.assign(npBT, 0.0);
	state_t &z0 = traj_[0].second;
	for (uint32_t i = 0; i < nx_; i++) z0[i] = x[i];
	for (uint32_t i = nx_; i < nz; i += nx_ + 1) z0[i] = 1.0;
	std::vector<double> zd(nz), DfCL(nx_ * nx_);
	for (uint32_t s = 0; s < npBT; s++) {
		state_t &z = traj_[s].second;
		if (s > 0) {
			const state_t &zp = traj_[s - 1].second;
			traj_[s].first = traj_[s - 1].first + dt;
			closedLoop(zp.data(), zd.data(), DfCL.data(), (double)s * dt); // :567: stamped s*backTrajDt
			for (uint32_t i = 0; i < nx_; i++)
				for (uint32_t j = 0; j < nx_; j++) {
					double v = 0.0;
					for (uint32_t k = 0; k < nx_; k++) v = v + DfCL[i + k * nx_] * zp[nx_ + k + j * nx_];
					zd[nx_ + i + j * nx_] = v;
				}
			for (uint32_t k = 0; k < nz; k++) z[k] = zd[k] * dt + zp[k];
		}
		double *h = &hAll_[(size_t)s * npSS_];
		safetySet_(z.data(), h, &DhAll_[(size_t)s * npSS_ * nx_]);
		hMin_[s] = *std::min_element(h, h + npSS_);
	}
}

void BackupLoopHost::lowestFirst(std::vector<uint32_t> &order, uint32_t count) const
{
	order.resize(count);
	std::iota(order.begin(), order.end(), 0u);
	std::stable_sort(order.begin(), order.end(), [this](uint32_t a, uint32_t b) { return hMin_[a] < hMin_[b]; });
}

static uint32_t trajectoryLength(double T, double &dt, uint32_t npBTSS)
{
	uint32_t n = (uint32_t)(std::round(T / dt) + 1);
	if (n < npBTSS) {
		n = npBTSS;
		dt = T / static_cast<double>(n - 1);
	}
	return n;
}

// ================================================================================= ASIFimplicit
ASIFimplicit::ASIFimplicit(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS,
                           const uint32_t npBTSS, SetFn safetySet, SetFn backupSet, DynFn dynamics,
                           DynGradFn dynamicsGradients, CtrlFn backupController, const QPSOLVER qpSolverType, const bool diagonalCost)
    : BackupLoopHost(false, nx, nu, npSS, safetySet, dynamics, dynamicsGradients, nullptr, backupController),
      hBackupEnd_(0.), hSafetyNow_(0.), nv_(nu + 2), npBS_(npBS), npBTSS_(npBTSS), npTC_(npBTSS * npSS + npBS),
      backupSet_(backupSet), options_(), QPsolver_(makeQPWrapper(qpSolverType, nu + 2, npBTSS * npSS + npBS, diagonalCost)),
      npBT_(0), H_(nv_ * nv_, 0.0), c_(nv_, 0.0), A_(npTC_ * nv_, 0.0), b_(npTC_, 0.0), lb_(nv_, 0.0), ub_(nv_, 0.0),
      batch_(nullptr)
{
	index_debug_ = 0;
	Dh_index_.assign((size_t)nx * npSS, 0.0);
	h_index_.assign(npSS, 0.0);
}

ASIFimplicit::ASIFimplicit(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS,
                           const uint32_t npBTSS, SetFn safetySet, SetFn backupSet, DynWithGradFn dynamicsWithGradient,
                           CtrlFn backupController, const QPSOLVER qpSolverType, const bool diagonalCost)
    : BackupLoopHost(true, nx, nu, npSS, safetySet, nullptr, nullptr, dynamicsWithGradient, backupController),
      hBackupEnd_(0.), hSafetyNow_(0.), nv_(nu + 2), npBS_(npBS), npBTSS_(npBTSS), npTC_(npBTSS * npSS + npBS),
      backupSet_(backupSet), options_(), QPsolver_(makeQPWrapper(qpSolverType, nu + 2, npBTSS * npSS + npBS, diagonalCost)),
      npBT_(0), H_(nv_ * nv_, 0.0), c_(nv_, 0.0), A_(npTC_ * nv_, 0.0), b_(npTC_, 0.0), lb_(nv_, 0.0), ub_(nv_, 0.0),
      batch_(nullptr)
{
	index_debug_ = 0;
	Dh_index_.assign((size_t)nx * npSS, 0.0);
	h_index_.assign(npSS, 0.0);
}

ASIFimplicit::~ASIFimplicit(void)
{
	delete QPsolver_;
	if (batch_) asif_hip_multi_destroy(batch_);
}

int32_t ASIFimplicit::initialize(const double lb[], const double ub[])
{
	npBT_ = trajectoryLength(options_.backTrajHorizon, options_.backTrajDt, npBTSS_); // :211-216
	holdStep_ = options_.backTrajDt;
	if (options_.n_debug != -1) { // :219-224
		if (options_.n_debug > -1 && options_.n_debug < (int)npBT_ - 1) index_debug_ = options_.n_debug;
		else options_.n_debug = -1;
	}
	satSharpness_ = options_.satSharpness;
	for (uint32_t j = 0; j < nu_; j++) {
		H_[j + j * nv_] = 1.0;
		lb_[j] = lbU_[j] = lb[j];
		ub_[j] = ubU_[j] = ub[j];
	}
	H_[(nv_ - 2) + (nv_ - 2) * nv_] = options_.relaxCost;
	H_[(nv_ - 1) + (nv_ - 1) * nv_] = options_.relaxCost;
	lb_[nv_ - 2] = options_.relaxSafeLb;
	lb_[nv_ - 1] = options_.relaxReachLb;
	ub_[nv_ - 2] = ub_[nv_ - 1] = options_.inf;
	const std::vector<double> origin(nx_, 0.0);
	updateConstraints(options_.x0 ? options_.x0 : origin.data());
	c_[nv_ - 2] = -2.0 * options_.relaxCost * options_.relaxSafeLb;
	c_[nv_ - 1] = -2.0 * options_.relaxCost * options_.relaxReachLb;
	for (uint32_t j = 0; j < nu_; j++) c_[j] = -0.0;
	const int32_t r = QPsolver_->initialize(H_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data());
	return r != 0 ? r : 1;
}

int32_t ASIFimplicit::initialize(const double lb[], const double ub[], const Options &options)
{
	options_ = options;
	return initialize(lb, ub);
}

int32_t ASIFimplicit::filter(const double x[], const double uDes[], double uAct[])
{
	double relax[2];
	return filter(x, uDes, uAct, relax);
}

int32_t ASIFimplicit::filter(const double x[], const double uDes[], double uAct[], double relax[2])
{
	for (uint32_t j = 0; j < nu_; j++) c_[j] = -2.0 * uDes[j];
	return filter(x, nullptr, c_.data(), uAct, relax);
}

int32_t ASIFimplicit::filter(const double x[], const double H[], const double c[], double uAct[])
{
	double relax[2];
	return filter(x, H, c, uAct, relax);
}

int32_t ASIFimplicit::filter(const double x[], const double H[], const double c[], double uAct[], double relax[2])
{
	// diagnostics of the reference: backup margin at the end of the PREVIOUS trajectory (:315-318)
	if (!traj_.empty()) {
		std::vector<double> hb(npBS_), Dhb(npBS_ * nx_);
		backupSet_(traj_.back().second.data(), hb.data(), Dhb.data());
		hBackupEnd_ = *std::min_element(hb.begin(), hb.end());
	}
	std::vector<double> hs(npSS_), Dhs(npSS_ * nx_);
	safetySet_(x, hs.data(), Dhs.data());
	hSafetyNow_ = *std::min_element(hs.begin(), hs.end());

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
	if (status == (int32_t)QPWrapperAbstract::SOLVER_STATUS::FEASIBLE) {
		std::vector<double> sol(nv_);
		QPsolver_->getSolution(sol.data());
		std::copy(sol.begin(), sol.begin() + nu_, uAct);
		saturateHard(uAct);
		relax[0] = sol[nu_];
		relax[1] = sol[nu_ + 1];
		return 1;
	}
	std::vector<double> Du(nu_ * nx_);
	backupController_(x, uAct, Du.data()); // :348-355
	saturateHard(uAct);
	return -1;
}

int32_t ASIFimplicit::updateOptions(const Options &options)
{
	options_ = options;
	return updateOptions();
}

int32_t ASIFimplicit::updateOptions(void)
{
	npBT_ = trajectoryLength(options_.backTrajHorizon, options_.backTrajDt, npBTSS_);
	holdStep_ = options_.backTrajDt;
	H_[(nv_ - 2) + (nv_ - 2) * nv_] = options_.relaxCost;
	H_[(nv_ - 1) + (nv_ - 1) * nv_] = options_.relaxCost;
	c_[nv_ - 2] = -2.0 * options_.relaxCost * options_.relaxSafeLb;
	c_[nv_ - 1] = -2.0 * options_.relaxCost * options_.relaxReachLb;
	lb_[nv_ - 2] = options_.relaxSafeLb;
	lb_[nv_ - 1] = options_.relaxReachLb;
	QPsolver_->updateBounds(lb_.data(), nullptr);
	QPsolver_->updateCost(H_.data(), c_.data());
	int32_t rc = 1; // :387-400
	if (options_.satSharpness > 2) { options_.satSharpness = 2; rc = 2; }
	else if (options_.satSharpness < 0.01) { options_.satSharpness = 0.01; rc = 3; }
	satSharpness_ = options_.satSharpness;
	if (batch_) {
		asif_hip_options o;
		fillOptions(o);
		asif_hip_multi_update_options(batch_, &o);
	}
	return rc;
}

// src/asif_implicit.cpp:403-651
int32_t ASIFimplicit::updateConstraints(const double x[])
{
	std::vector<double> f(nx_), g(nx_ * nu_);
	dynamics_(x, f.data(), g.data());
	integrate(x, npBT_, options_.backTrajDt);
	std::vector<uint32_t> order;
	lowestFirst(order, npBT_);
	backTrajCritIdx_.assign(order.begin(), order.begin() + npBTSS_);
	std::vector<double> h(npTC_, 0.0), Dh(npTC_ * nx_, 0.0);
	// Dh_SS(x_s) Q_s, column-major npSS x nx
	auto sampleProduct = [this](uint32_t s, double *out) {
		const double *Q = traj_[s].second.data() + nx_;
		for (uint32_t i = 0; i < npSS_; i++)
			for (uint32_t j = 0; j < nx_; j++) {
				double v = 0.0;
				for (uint32_t c = 0; c < nx_; c++) v = v + DhAll_[(size_t)s * npSS_ * nx_ + i + c * npSS_] * Q[c + j * nx_];
				out[i + j * npSS_] = v;
			}
	};
	std::vector<double> prod(npSS_ * nx_);
	if (options_.n_debug != -1) { // :500-515: the sample the network sees is fixed by the caller
		sampleProduct((uint32_t)index_debug_, Dh_index_.data());
		for (uint32_t i = 0; i < npSS_; i++) h_index_[i] = hAll_[(size_t)index_debug_ * npSS_ + i];
	}
	for (uint32_t k = 0; k < npBTSS_; k++) {
		const uint32_t s = order[k];
		sampleProduct(s, prod.data());
		for (uint32_t i = 0; i < npSS_; i++) {
			h[k * npSS_ + i] = hAll_[(size_t)s * npSS_ + i];
			for (uint32_t j = 0; j < nx_; j++) Dh[(k * npSS_ + i) + j * npTC_] = prod[i + j * npSS_];
		}
		if (k == 0 && options_.n_debug == -1) { // :533-537
			Dh_index_ = prod;
			for (uint32_t i = 0; i < npSS_; i++) h_index_[i] = h[i];
			index_debug_ = (int)s;
		}
		safeMargins(traj_[s].second.data(), &h[k * npSS_]); // ASIFimplicitRB: interval lower ends
	}
	const state_t &zend = traj_.back().second;
	std::vector<double> DhB(npBS_ * nx_);
	backupSet_(zend.data(), &h[npBTSS_ * npSS_], DhB.data());
	for (uint32_t i = 0; i < npBS_; i++)
		for (uint32_t j = 0; j < nx_; j++) {
			double v = 0.0;
			for (uint32_t c = 0; c < nx_; c++) v = v + DhB[i + c * npBS_] * zend[nx_ + c + j * nx_];
			Dh[(npBTSS_ * npSS_ + i) + j * npTC_] = v;
		}
	std::vector<double> Lfh(npTC_), Lgh(npTC_ * nu_);
	for (uint32_t r = 0; r < npTC_; r++) {
		double Lf = 0.0;
		for (uint32_t c = 0; c < nx_; c++) Lf = Lf + Dh[r + c * npTC_] * f[c];
		Lfh[r] = Lf;
		for (uint32_t j = 0; j < nu_; j++) {
			double Lg = 0.0;
			for (uint32_t c = 0; c < nx_; c++) Lg = Lg + Dh[r + c * npTC_] * g[c + j * nx_];
			Lgh[r + j * npTC_] = Lg;
		}
	}
	// :556-583 (the reference grows the three by push_back at every initialize(); only the first npTC entries are ever
	// written or meaningful)
	Lfh_out_.assign(Lfh.begin(), Lfh.end());
	Lgh_out_.assign(npTC_, std::vector<double>(nu_, 0.0));
	Dh_out_.assign(npTC_, std::vector<double>(nx_, 0.0));
	for (uint32_t i = 0; i < npTC_; i++) {
		for (uint32_t j = 0; j < nu_; j++) Lgh_out_[i][j] = Lgh[nu_ * i + j];
		for (uint32_t j = 0; j < nx_; j++) Dh_out_[i][j] = Dh[nx_ * i + j];
	}
	if (options_.use_learning) // :585-588
		update_weights(&learning_data_, x, nx_, Dh_index_.data(), Lfh.data(), Lgh.data(), nu_);
	std::fill(A_.begin(), A_.end(), 0.0);
	for (uint32_t r = 0; r < npTC_; r++) {
		for (uint32_t j = 0; j < nu_; j++) A_[r + j * npTC_] = Lgh[r + j * npTC_];
		A_[r + (r < npBTSS_ * npSS_ ? nu_ : nu_ + 1) * npTC_] = h[r];
		b_[r] = -Lfh[r];
	}
	return 1;
}

std::string ASIFimplicit::filterErrorMsgString(const int32_t rc)
{
	return rc == 1 ? "Success" : (rc == -1 ? "QP failed" : "Unkown");
}

void ASIFimplicit::fillOptions(asif_hip_options &o) const
{
	asif_hip_default_options(ASIF_HIP_MODEL_INVERTED_PENDULUM, ASIF_HIP_IMPLICIT, &o);
	o.relaxCost = options_.relaxCost;
	o.relaxLb = options_.relaxSafeLb;
	o.relaxReachLb = options_.relaxReachLb;
	o.backTrajHorizon = options_.backTrajHorizon;
	o.backTrajDt = options_.backTrajDt;
	o.satSharpness = options_.satSharpness;
	o.inf = options_.inf;
	o.n_debug = options_.n_debug;
	o.use_learning = options_.use_learning ? 1 : 0;
	for (uint32_t j = 0; j < nu_ && j < ASIF_HIP_MAX_NU; j++) {
		o.lb[j] = lbU_[j];
		o.ub[j] = ubU_[j];
	}
}

int32_t ASIFimplicit::bindDeviceModel(int model, int device)
{
	const int32_t devs[1] = {device};
	return bindDeviceModel(model, 1, devs);
}

int32_t ASIFimplicit::bindDeviceModel(int model, int32_t ndev, const int32_t devs[])
{
	if (batch_) asif_hip_multi_destroy(batch_);
	batch_ = nullptr;
	asif_hip_options o;
	fillOptions(o);
	int r = asif_hip_create_multi(&batch_, model, deviceVariant(), &o, nullptr, ndev, devs);
	if (r) return r;
	if (options_.use_learning) {
		const LearningData &d = learning_data_;
		asif_hip_learning_data L = {d.d_drift_in, d.d_act_in, d.d_drift_hidden, d.d_act_hidden, d.d_drift_hidden_2,
		                            d.d_act_hidden_2, d.d_drift_out, d.d_act_out, d.w_1_drift, d.w_2_drift, d.w_3_drift,
		                            d.b_1_drift, d.b_2_drift, d.b_3_drift, d.w_1_act, d.w_2_act, d.w_3_act, d.b_1_act,
		                            d.b_2_act, d.b_3_act};
		for (int i = 0; i < asif_hip_multi_size(batch_) && r == 0; i++)
			r = asif_hip_set_learning(asif_hip_multi_handle(batch_, i), &L);
		if (r) {
			asif_hip_multi_destroy(batch_);
			batch_ = nullptr;
			return r;
		}
	}
	asif_hip_dims d;
	asif_hip_get_dims(asif_hip_multi_handle(batch_, 0), &d);
	if ((uint32_t)d.nx != nx_ || (uint32_t)d.nu != nu_ || (uint32_t)d.nc != npTC_) {
		asif_hip_multi_destroy(batch_);
		batch_ = nullptr;
		return ASIF_HIP_EINVAL;
	}
	return 0;
}

int32_t ASIFimplicit::filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[],
                                  int32_t rc[])
{
	if (!batch_) return ASIF_HIP_EINVAL;
	return asif_hip_filter_batch_host_multi(batch_, B, x, uDes, uAct, relax, rc);
}

// =============================================================================== ASIFimplicitTB
ASIFimplicitTB::ASIFimplicitTB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBTSS,
                               SetFn safetySet, SetHessFn backupSet, DynFn dynamics, DynGradFn dynamicsGradients,
                               CtrlFn backupController, const QPSOLVER qpSolverType, const bool diagonalCost)
    : BackupLoopHost(false, nx, nu, npSS, safetySet, dynamics, dynamicsGradients, nullptr, backupController), TTS_(0.),
      BTorthoBS_(0.), hBackupEnd_(0.), hSafetyNow_(0.), nv_(nu + 1), npBTSS_(npBTSS), npTC_(npBTSS * npSS + 2),
      backupSet_(backupSet), options_(), QPsolver_(makeQPWrapper(qpSolverType, nu + 1, npBTSS * npSS + 2, diagonalCost)), npBT_(0),
      afterUpdate_(false), H_(nv_ * nv_, 0.0), c_(nv_, 0.0), A_(npTC_ * nv_, 0.0), b_(npTC_, 0.0), lb_(nv_, 0.0),
      ub_(nv_, 0.0), batch_(nullptr)
{
}

ASIFimplicitTB::ASIFimplicitTB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBTSS,
                               SetFn safetySet, SetHessFn backupSet, DynWithGradFn dynamicsWithGradient,
                               CtrlFn backupController, const QPSOLVER qpSolverType, const bool diagonalCost)
    : BackupLoopHost(true, nx, nu, npSS, safetySet, nullptr, nullptr, dynamicsWithGradient, backupController), TTS_(0.),
      BTorthoBS_(0.), hBackupEnd_(0.), hSafetyNow_(0.), nv_(nu + 1), npBTSS_(npBTSS), npTC_(npBTSS * npSS + 2),
      backupSet_(backupSet), options_(), QPsolver_(makeQPWrapper(qpSolverType, nu + 1, npBTSS * npSS + 2, diagonalCost)), npBT_(0),
      afterUpdate_(false), H_(nv_ * nv_, 0.0), c_(nv_, 0.0), A_(npTC_ * nv_, 0.0), b_(npTC_, 0.0), lb_(nv_, 0.0),
      ub_(nv_, 0.0), batch_(nullptr)
{
}

ASIFimplicitTB::~ASIFimplicitTB(void)
{
	delete QPsolver_;
	if (batch_) asif_hip_multi_destroy(batch_);
}

int32_t ASIFimplicitTB::initialize(const double lb[], const double ub[])
{
	// the horizon is stretched by (1 + backTrajExtend) here and only here (:177-182 vs :377-382)
	npBT_ = trajectoryLength(options_.backTrajHorizon * (1.0 + options_.backTrajExtend), options_.backTrajDt, npBTSS_);
	afterUpdate_ = false;
	traj_.assign(npBT_, std::pair<double, state_t>(0.0, state_t(nx_ + nx_ * nx_, 0.0)));
	satSharpness_ = options_.satSharpness;
	for (uint32_t j = 0; j < nu_; j++) {
		H_[j + j * nv_] = 1.0;
		lb_[j] = lbU_[j] = lb[j];
		ub_[j] = ubU_[j] = ub[j];
	}
	H_[(nv_ - 1) + (nv_ - 1) * nv_] = options_.relaxCost;
	lb_[nv_ - 1] = options_.relaxSafeLb;
	ub_[nv_ - 1] = options_.inf;
	updateConstraintsTrivial();
	c_[nv_ - 1] = -2.0 * options_.relaxCost * options_.relaxSafeLb;
	for (uint32_t j = 0; j < nu_; j++) c_[j] = -0.0;
	const int32_t r = QPsolver_->initialize(H_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data());
	return r != 0 ? r : 1;
}

int32_t ASIFimplicitTB::initialize(const double lb[], const double ub[], const Options &options)
{
	options_ = options;
	return initialize(lb, ub);
}

int32_t ASIFimplicitTB::filter(const double x[], const double uDes[], double uAct[])
{
	double relax;
	return filter(x, uDes, uAct, relax);
}

int32_t ASIFimplicitTB::filter(const double x[], const double uDes[], double uAct[], double &relax)
{
	for (uint32_t j = 0; j < nu_; j++) c_[j] = -2.0 * uDes[j];
	return filter(x, nullptr, c_.data(), uAct, relax);
}

int32_t ASIFimplicitTB::filter(const double x[], const double H[], const double c[], double uAct[])
{
	double relax;
	return filter(x, H, c, uAct, relax);
}

int32_t ASIFimplicitTB::solveAndFinish(const double x[], const double H[], const double c[], double uAct[], double &relax,
                                       int32_t okCode, bool leakSolverCode)
{
	if (H == nullptr) QPsolver_->updateCost(nullptr, c);
	else {
		for (uint32_t j = 0; j < nu_; j++)
			for (uint32_t i = 0; i < nu_; i++) H_[i + j * nv_] = H[i + j * nu_];
		QPsolver_->updateCost(H_.data(), c);
	}
	QPsolver_->updateA(A_.data());
	QPsolver_->updateb(b_.data());
	const int32_t status = QPsolver_->solve();
	if (status == 1) {
		std::vector<double> sol(nv_);
		QPsolver_->getSolution(sol.data());
		std::copy(sol.begin(), sol.begin() + nu_, uAct);
		saturateHard(uAct);
		relax = sol[nu_];
		return okCode;
	}
	std::vector<double> Du(nu_ * nx_);
	backupController_(x, uAct, Du.data());
	saturateHard(uAct);
	return leakSolverCode ? status : -1; // :351 vs :315
}

// src/asif_implicit_tb.cpp:261-363
int32_t ASIFimplicitTB::filter(const double x[], const double H[], const double c[], double uAct[], double &relax)
{
	std::vector<double> hb(1), Dhb(nx_), DDhb(nx_ * nx_);
	backupSet_(traj_.back().second.data(), hb.data(), Dhb.data(), DDhb.data());
	hBackupEnd_ = hb[0];
	backupSet_(x, hb.data(), Dhb.data(), DDhb.data());
	std::vector<double> hs(npSS_), Dhs(npSS_ * nx_);
	safetySet_(x, hs.data(), Dhs.data());
	hSafetyNow_ = *std::min_element(hs.begin(), hs.end());
	if (hb[0] >= 0) {
		updateConstraintsTrivial();
		return solveAndFinish(x, H, c, uAct, relax, 2, false);
	}
	if (updateConstraints(x) == 1) {
		double he;
		backupSet_(traj_.back().second.data(), &he, Dhb.data(), DDhb.data());
		hBackupEnd_ = he;
		return solveAndFinish(x, H, c, uAct, relax, 1, true);
	}
	std::vector<double> Du(nu_ * nx_);
	backupController_(x, uAct, Du.data());
	saturateHard(uAct);
	return -3;
}

int32_t ASIFimplicitTB::updateOptions(const Options &options)
{
	options_ = options;
	return updateOptions();
}

int32_t ASIFimplicitTB::updateOptions(void)
{
	npBT_ = trajectoryLength(options_.backTrajHorizon, options_.backTrajDt, npBTSS_); // no (1+extend): :377
	afterUpdate_ = true;
	traj_.assign(npBT_, std::pair<double, state_t>(0.0, state_t(nx_ + nx_ * nx_, 0.0)));
	H_[(nv_ - 1) + (nv_ - 1) * nv_] = options_.relaxCost;
	c_[nv_ - 1] = -2.0 * options_.relaxCost * options_.relaxSafeLb;
	lb_[nv_ - 1] = options_.relaxSafeLb;
	QPsolver_->updateBounds(lb_.data(), nullptr);
	QPsolver_->updateCost(H_.data(), c_.data());
	int32_t rc = 1;
	if (options_.satSharpness > 2) { options_.satSharpness = 2; rc = 2; }
	else if (options_.satSharpness < 0.01) { options_.satSharpness = 0.01; rc = 3; }
	satSharpness_ = options_.satSharpness;
	if (batch_) {
		asif_hip_options o;
		fillOptions(o);
		asif_hip_multi_update_options(batch_, &o);
	}
	return rc;
}

// :716-733
int32_t ASIFimplicitTB::updateConstraintsTrivial(void)
{
	std::fill(A_.begin(), A_.end(), 0.0);
	std::fill(b_.begin(), b_.end(), -options_.inf);
	TTS_ = 0.0;
	BTorthoBS_ = 1.0;
	return 1;
}

// :407-714
int32_t ASIFimplicitTB::updateConstraints(const double x[])
{
	std::vector<double> f(nx_), g(nx_ * nu_);
	dynamics_(x, f.data(), g.data());
	integrate(x, npBT_, options_.backTrajDt);
	// first sample inside the backup set ("first hit wins", :505-528)
	std::vector<double> DhBS(nx_), DDhBS(nx_ * nx_), fCl(nx_), DfCl(nx_ * nx_);
	double hBS = -1.0;
	uint32_t idxHit = 0;
	bool hit = false;
	for (uint32_t s = 1; s < npBT_ && !hit; s++) {
		backupSet_(traj_[s].second.data(), &hBS, DhBS.data(), DDhBS.data());
		if (hBS >= 0.0) {
			hit = true;
			idxHit = s;
		}
	}
	if (!hit) {
		BTorthoBS_ = 0;
		return -1;
	}
	const state_t &zh = traj_[idxHit].second;
	closedLoop(zh.data(), fCl.data(), DfCl.data());
	double cosT = 0.0, n1 = 0.0, n2 = 0.0;
	for (uint32_t c = 0; c < nx_; c++) {
		cosT = cosT + DhBS[c] * fCl[c];
		n1 += DhBS[c] * DhBS[c];
		n2 += fCl[c] * fCl[c];
	}
	const double den1 = std::sqrt(n1), den2 = std::sqrt(n2), den = den1 * den2;
	BTorthoBS_ = cosT / den;

	std::vector<uint32_t> order;
	lowestFirst(order, idxHit + 1);
	backTrajCritIdx_.assign(order.begin(), order.begin() + std::min<uint32_t>(npBTSS_, idxHit + 1));
	std::vector<double> h(npTC_, 0.0), Dh(npTC_ * nx_, 0.0);
	for (uint32_t k = 0; k < npBTSS_; k++) {
		if (k > idxHit) { // inert padding rows, :556-566
			for (uint32_t i = 0; i < npSS_; i++) h[k * npSS_ + i] = 1.0;
			continue;
		}
		const uint32_t s = order[k];
		const double *Q = traj_[s].second.data() + nx_;
		for (uint32_t i = 0; i < npSS_; i++) {
			h[k * npSS_ + i] = hAll_[(size_t)s * npSS_ + i];
			for (uint32_t j = 0; j < nx_; j++) {
				double v = 0.0;
				for (uint32_t c = 0; c < nx_; c++) v = v + DhAll_[(size_t)s * npSS_ * nx_ + i + c * npSS_] * Q[c + j * nx_];
				Dh[(k * npSS_ + i) + j * npTC_] = v;
			}
		}
	}
	TTS_ = traj_[idxHit].first;
	const double hReach = options_.backTrajHorizon - traj_[idxHit].first;
	const double *Qh = zh.data() + nx_;
	std::vector<double> DhQ(nx_);
	for (uint32_t j = 0; j < nx_; j++) {
		double v = 0.0;
		for (uint32_t c = 0; c < nx_; c++) v = v + DhBS[c] * Qh[c + j * nx_];
		DhQ[j] = v;
	}
	const uint32_t rT = npBTSS_ * npSS_, rO = rT + 1;
	h[rT] = hReach;
	for (uint32_t j = 0; j < nx_; j++) Dh[rT + j * npTC_] = DhQ[j] / cosT;
	h[rO] = BTorthoBS_ - options_.backTrajMinOrtho;
	std::vector<double> DxHit(nx_ * nx_);
	for (uint32_t r = 0; r < nx_; r++)
		for (uint32_t j = 0; j < nx_; j++) DxHit[r + j * nx_] = Qh[r + j * nx_] - fCl[r] * DhQ[j];
	for (uint32_t j = 0; j < nx_; j++) {
		double Dnum = 0.0, Dd1 = 0.0, Dd2 = 0.0;
		for (uint32_t k = 0; k < nx_; k++) {
			double t1 = 0.0, t2 = 0.0;
			for (uint32_t l = 0; l < nx_; l++) {
				t1 += DDhBS[k + l * nx_] * DxHit[l + j * nx_];
				t2 += DfCl[k + l * nx_] * DxHit[l + j * nx_];
			}
			const double t3 = DhBS[k] * t2, t4 = t1 * fCl[k];
			Dd1 += t3;
			Dd2 += t4;
			Dnum += t3 + t4;
		}
		const double Dden = den2 * Dd1 / den1 + den1 * Dd2 / den2;
		Dh[rO + j * npTC_] = (Dnum * den - cosT * Dden) / (den * den);
	}
	std::fill(A_.begin(), A_.end(), 0.0);
	for (uint32_t r = 0; r < npTC_; r++) {
		double Lf = 0.0;
		for (uint32_t c = 0; c < nx_; c++) Lf = Lf + Dh[r + c * npTC_] * f[c];
		for (uint32_t j = 0; j < nu_; j++) {
			double Lg = 0.0;
			for (uint32_t c = 0; c < nx_; c++) Lg = Lg + Dh[r + c * npTC_] * g[c + j * nx_];
			A_[r + j * npTC_] = Lg;
		}
		if (r < rT) A_[r + nu_ * npTC_] = h[r];
		b_[r] = -Lf;
	}
	b_[rT] -= options_.relaxTTS * h[rT];
	b_[rO] -= options_.relaxMinOrtho * h[rO];
	return 1;
}

std::string ASIFimplicitTB::filterErrorMsgString(const int32_t rc)
{
	switch (rc) {
	case 1: return "Success";
	case 2: return "Success: inside the backup set, trivial constraints";
	case -1: return "QP failed inside the backup set: backup controller applied";
	case -3: return "Backup set not reached within the horizon (or QP primal infeasible): backup controller applied";
	default: return "QP failed with solver status " + std::to_string(rc) + ": backup controller applied";
	}
}

void ASIFimplicitTB::fillOptions(asif_hip_options &o) const
{
	asif_hip_default_options(ASIF_HIP_MODEL_SEGWAY, ASIF_HIP_IMPLICIT_TB, &o);
	o.relaxCost = options_.relaxCost;
	o.relaxLb = options_.relaxSafeLb;
	o.relaxTTS = options_.relaxTTS;
	o.relaxMinOrtho = options_.relaxMinOrtho;
	o.backTrajHorizon = options_.backTrajHorizon;
	o.backTrajExtend = options_.backTrajExtend;
	o.backTrajDt = options_.backTrajDt;
	o.backTrajMinOrtho = options_.backTrajMinOrtho;
	o.satSharpness = options_.satSharpness;
	o.inf = options_.inf;
	for (uint32_t j = 0; j < nu_ && j < ASIF_HIP_MAX_NU; j++) {
		o.lb[j] = lbU_[j];
		o.ub[j] = ubU_[j];
	}
}

int32_t ASIFimplicitTB::bindDeviceModel(int model, int device)
{
	const int32_t devs[1] = {device};
	return bindDeviceModel(model, 1, devs);
}

// BASELINE.json config 4: the agents of one call spread over the GPUs of a node, no collective
int32_t ASIFimplicitTB::bindDeviceModel(int model, int32_t ndev, const int32_t devs[])
{
	if (batch_) asif_hip_multi_destroy(batch_);
	batch_ = nullptr;
	asif_hip_options o;
	fillOptions(o);
	int r = asif_hip_create_multi(&batch_, model, ASIF_HIP_IMPLICIT_TB, &o, nullptr, ndev, devs);
	if (r) return r;
	if (afterUpdate_) asif_hip_multi_update_options(batch_, &o); // same trajectory length as the host object
	asif_hip_dims d;
	asif_hip_get_dims(asif_hip_multi_handle(batch_, 0), &d);
	if ((uint32_t)d.nx != nx_ || (uint32_t)d.nu != nu_ || (uint32_t)d.nc != npTC_) {
		asif_hip_multi_destroy(batch_);
		batch_ = nullptr;
		return ASIF_HIP_EINVAL;
	}
	return 0;
}

int32_t ASIFimplicitTB::filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[],
                                    int32_t rc[])
{
	if (!batch_) return ASIF_HIP_EINVAL;
	return asif_hip_filter_batch_host_multi(batch_, B, x, uDes, uAct, relax, rc);
}

} // namespace ASIF
