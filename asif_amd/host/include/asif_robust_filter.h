// asif_robust_filter.h -- class ASIF::ASIFrobust with the reference's public interface
// (include/asif_robust.h:11-89): safety set on doubles, dynamics on affine forms (interval_t = AAF),
// initialize, the filter overloads, updateOptions; plus filterBatch() on a compiled device model.
// Single-agent filter(): interval Lie derivatives and the 3N x (2+4N) rows on the host
// (src/asif_robust.cpp:275-367), the full QP -- multipliers included -- on the GPU's wave-per-QP kernel.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>
#include "asif_affine.h"
#include "qpwrappers.h"

namespace ASIF {

class ASIFrobust {
public:
	typedef struct {
		double relaxLb = 5.0;
		double relaxCost = 50.0;
		double inf = 1e20;
	} Options;
	typedef std::function<void(const double * /*x*/, double * /*h*/, double * /*Dh*/)> SafetySetFn;
	typedef std::function<void(const interval_t * /*x*/, interval_t * /*f*/, interval_t * /*g*/)> DynamicsFn;

	ASIFrobust(const uint32_t nx, const uint32_t nu, const uint32_t npSS, SafetySetFn safetySet, DynamicsFn dynamics,
	           const uint32_t npSSmax = -1, const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);
	~ASIFrobust(void);

	int32_t initialize(const double lb[], const double ub[]);
	int32_t initialize(const double lb[], const double ub[], const Options &options);
	int32_t filter(const double x[], const double uDes[], double uAct[]);
	int32_t filter(const double x[], const double uDes[], double uAct[], double &relax);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[]);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[], double &relax);
	int32_t updateOptions(void);
	int32_t updateOptions(const Options &options);

	// modelData carries what only the callbacks know: half-planes, pMin/pMax (asif_hip_options fields)
	int32_t bindDeviceModel(int asif_hip_model_id, const asif_hip_options &modelData, int device = 0);
	// half-plane safety set as data (examples/DoubleIntegrator_Robust.cpp: SafetySetData, [N][2] flattened);
	// modelData carries the interval parameters of the device model, the rest is taken from this object
	int32_t bindDeviceData(int asif_hip_model_id, const double halfPlanes[], int32_t N,
	                       const asif_hip_robust_data_options &modelData, int device = 0);
	int32_t filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[], int32_t rc[]);

	const double *rowsA(void) const { return A_.data(); } // last assembled rows (nc x nv, column-major)
	uint32_t nv(void) const { return nv_; }
	uint32_t nc(void) const { return nc_; }

protected:
	int32_t updateConstraints(const double x[]);
	const uint32_t nx_, nu_, npSS_, npSSmax_, nv_, nc_;
	SafetySetFn safetySet_;
	DynamicsFn dynamics_;
	Options options_;
	QPWrapperAbstract *QPsolver_;
	std::vector<double> H_, c_, A_, b_, lb_, ub_;
	asif_hip_ctx *batch_;
	asif_hip_options batchOpts_;
	bool batchIsData_ = false;
	asif_hip_robust_data_options batchDataOpts_;
};

} // namespace ASIF
