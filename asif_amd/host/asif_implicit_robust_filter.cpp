// asif_implicit_robust_filter.cpp -- host side of ASIF::ASIFimplicitRB (see the header).  Everything the class
// shares with ASIFimplicit (trajectory, rows, learned residual, QP, epilogue) is the base class; this file adds
// the option mapping, the hold period and the interval margins.
#include "asif_implicit_robust_filter.h"

namespace ASIF {

ASIFimplicitRB::ASIFimplicitRB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS,
                               const uint32_t npBTSS, SetFn safetySet, IntervalFn safetySet_int, SetFn backupSet,
                               IntervalFn backupSet_int, DynFn dynamics, IntervalFn dynamics_int,
                               DynGradFn dynamicsGradients, IntervalFn dynamicsGradients_int, CtrlFn backupController,
                               const QPSOLVER qpSolverType, const bool diagonalCost)
    : ASIFimplicit(nx, nu, npSS, npBS, npBTSS, safetySet, backupSet, dynamics, dynamicsGradients, backupController,
                   qpSolverType, diagonalCost),
      rbOptions_(), xUnc_(nx, 0.0), safetySet_int_(safetySet_int), backupSet_int_(backupSet_int),
      dynamics_int_(dynamics_int), dynamicsGradients_int_(dynamicsGradients_int), dynamicsWithGradient_int_(nullptr)
{
}

ASIFimplicitRB::ASIFimplicitRB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS,
                               const uint32_t npBTSS, SetFn safetySet, IntervalFn safetySet_int, SetFn backupSet,
                               IntervalFn backupSet_int, DynWithGradFn dynamicsWithGradient,
                               IntervalDynWithGradFn dynamicsWithGradient_int, CtrlFn backupController,
                               const QPSOLVER qpSolverType, const bool diagonalCost)
    : ASIFimplicit(nx, nu, npSS, npBS, npBTSS, safetySet, backupSet, dynamicsWithGradient, backupController,
                   qpSolverType, diagonalCost),
      rbOptions_(), xUnc_(nx, 0.0), safetySet_int_(safetySet_int), backupSet_int_(backupSet_int), dynamics_int_(nullptr),
      dynamicsGradients_int_(nullptr), dynamicsWithGradient_int_(dynamicsWithGradient_int)
{
}

void ASIFimplicitRB::adopt(const Options &o)
{
	rbOptions_ = o;
	options_.x0 = o.x0;
	options_.n_debug = o.n_debug;
	options_.relaxCost = o.relaxCost;
	options_.relaxReachLb = o.relaxReachLb;
	options_.relaxSafeLb = o.relaxSafeLb;
	options_.backTrajHorizon = o.backTrajHorizon;
	options_.backTrajDt = o.backTrajDt;
	options_.backTrajAbsTol = o.backTrajAbsTol;
	options_.backTrajRelTol = o.backTrajRelTol;
	options_.satSharpness = o.satSharpness;
	options_.inf = o.inf;
	options_.use_learning = o.use_learning;
	holdDt_ = o.backContDt;
	for (uint32_t i = 0; i < nx_; i++) xUnc_[i] = o.x_unc ? o.x_unc[i] : 0.0; // :276-279
}

int32_t ASIFimplicitRB::initialize(const double lb[], const double ub[])
{
	adopt(rbOptions_);
	return ASIFimplicit::initialize(lb, ub);
}

int32_t ASIFimplicitRB::initialize(const double lb[], const double ub[], const Options &options)
{
	adopt(options);
	return ASIFimplicit::initialize(lb, ub);
}

int32_t ASIFimplicitRB::updateOptions(const Options &options)
{
	adopt(options);
	return ASIFimplicit::updateOptions();
}

// src/asif_implicit_robust.cpp:635-647
void ASIFimplicitRB::safeMargins(const double xs[], double h[]) const
{
	std::vector<interval_t> xi(nx_), hi(npSS_), Dhi(npSS_ * nx_);
	for (uint32_t i = 0; i < nx_; i++) xi[i] = interval(xs[i] - xUnc_[i], xs[i] + xUnc_[i]);
	safetySet_int_(xi.data(), hi.data(), Dhi.data());
	for (uint32_t i = 0; i < npSS_; i++) h[i] = hi[i].convert().left();
}

void ASIFimplicitRB::fillOptions(asif_hip_options &o) const
{
	ASIFimplicit::fillOptions(o);
	o.backContDt = rbOptions_.backContDt;
	for (uint32_t i = 0; i < nx_ && i < ASIF_HIP_MAX_NX; i++) o.x_unc[i] = xUnc_[i];
}

} // namespace ASIF
