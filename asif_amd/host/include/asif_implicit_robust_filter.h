// asif_implicit_robust_filter.h -- ASIF::ASIFimplicitRB with the reference's public interface
// (include/asif_implicit_robust.h:19-279): the backup-trajectory filter with
//   * the backup input held over Options::backContDt along the trajectory (src/asif_implicit_robust.cpp:891-903),
//   * the margins of the critical samples replaced by the lower end of the interval safety set over
//     x +- Options::x_unc (:635-647; safetySet_int is called with interval_t operands, here asif_affine.h's AAF),
//   * the learned residual and n_debug selection it shares with ASIFimplicit (:590-605,624-632,713-715).
// The interval Lie derivatives the reference also forms from backupSet_int / dynamics_int / dynamicsGradients_int
// (:698-709) feed nothing there; those callbacks are accepted and kept, never called.
// Single-agent filter(): trajectory, rows and intervals on the host with the user's callbacks, QP on the GPU.
// filterBatch(): everything on the GPU for a compiled device model (variant ASIF_HIP_IMPLICIT_RB).
#pragma once
#include "asif_affine.h"
#include "asif_backup_filters.h"

typedef AAF interval_t; // include/asif_implicit_robust.h:12

namespace ASIF {

typedef std::function<void(const interval_t * /*x*/, interval_t * /*a*/, interval_t * /*b*/)> IntervalFn;
typedef std::function<void(const interval_t * /*x*/, const double * /*u*/, interval_t * /*f*/, interval_t * /*g*/,
                           interval_t * /*d_fcl_dx*/)> IntervalDynWithGradFn;

class ASIFimplicitRB : public ASIFimplicit {
public:
	typedef struct {
		double *x0 = nullptr;
		double *x_unc = nullptr;
		int n_debug = -1;
		double relaxCost = 50.0;
		double relaxReachLb = 5.0;
		double relaxSafeLb = 5.0;
		double backTrajHorizon = 1.0;
		double backContDt = 0.01;
		double backTrajDt = 0.01;
		double backTrajAbsTol = 1.0e-6;
		double backTrajRelTol = 1.0e-6;
		double satSharpness = 0.1;
		double inf = 1e20;
		bool use_learning = false;
	} Options;

	ASIFimplicitRB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS, const uint32_t npBTSS,
	               SetFn safetySet, IntervalFn safetySet_int, SetFn backupSet, IntervalFn backupSet_int, DynFn dynamics,
	               IntervalFn dynamics_int, DynGradFn dynamicsGradients, IntervalFn dynamicsGradients_int,
	               CtrlFn backupController, const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);
	ASIFimplicitRB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS, const uint32_t npBTSS,
	               SetFn safetySet, IntervalFn safetySet_int, SetFn backupSet, IntervalFn backupSet_int,
	               DynWithGradFn dynamicsWithGradient, IntervalDynWithGradFn dynamicsWithGradient_int,
	               CtrlFn backupController, const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);

	int32_t initialize(const double lb[], const double ub[]);
	int32_t initialize(const double lb[], const double ub[], const Options &options);
	int32_t updateOptions(const Options &options);

protected:
	void fillOptions(asif_hip_options &o) const override;
	int deviceVariant(void) const override { return ASIF_HIP_IMPLICIT_RB; }
	void safeMargins(const double xs[], double h[]) const override;
	void adopt(const Options &o); // Options -> the base class's options + hold / uncertainty

	Options rbOptions_;
	std::vector<double> xUnc_;
	IntervalFn safetySet_int_, backupSet_int_, dynamics_int_, dynamicsGradients_int_;
	IntervalDynWithGradFn dynamicsWithGradient_int_;
};

} // namespace ASIF
