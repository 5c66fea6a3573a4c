// asif_filter.h -- class ASIF::ASIF, the explicit CBF filter, with the reference's public interface
// (include/asif.h:8-101: constructor, initialize, the six filter overloads, updateOptions) on top of
// ASIF::QPWrapperHip, plus filterBatch() for thousands of agents per call on a compiled device model.
//
// Single-agent filter(): user std::function callbacks run on the host exactly as in the reference;
// rows are assembled on the host (src/asif.cpp:233-312) and the QP goes to the GPU solver.
// filterBatch(): rows + solve + clamp fused on the GPU (asif_hip_filter_batch).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>
#include "qpwrappers.h"

namespace ASIF {

class ASIF {
public:
	typedef struct {
		double relaxLb = 5.0;
		double relaxCost = 50.0;
		double satSharpness = 5.0;
		double inf = 1e20;
	} Options;

	typedef std::function<void(const double * /*x*/, double * /*h*/, double * /*Dh*/)> SafetySetFn;
	typedef std::function<void(const double * /*x*/, double * /*f*/, double * /*g*/)> DynamicsFn;

	ASIF(const uint32_t nx, const uint32_t nu, const uint32_t npSS, SafetySetFn safetySet, DynamicsFn dynamics,
	     const uint32_t npSSmax = -1, const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);
	~ASIF(void);

	int32_t initialize(const double lb[], const double ub[]);
	int32_t initialize(const double lb[], const double ub[], const Options &options);

	int32_t filter(const double x[], const double uDes[], double uAct[]);
	int32_t filter(const double x[], const double uDes[], double uAct[], double Lfh[], double Lgh[]);
	int32_t filter(const double x[], const double uDes[], double uAct[], double &relax);
	int32_t filter(const double x[], const double uDes[], double uAct[], double Lfh[], double Lgh[], double &relax);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[]);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[], double &relax);

	int32_t updateOptions(void);
	int32_t updateOptions(const Options &options);

	// ---- batched extension -------------------------------------------------------------------
	// Binds the compiled device model that corresponds to the host callbacks (e.g.
	// ASIF_HIP_MODEL_DOUBLE_INTEGRATOR for examples/DoubleIntegrator.cpp).  Returns 0 or an asif_hip error.
	int32_t bindDeviceModel(int asif_hip_model_id, int device = 0);
	int32_t bindDeviceModel(int asif_hip_model_id, int32_t ndev, const int32_t devs[]); // one block of the batch per entry
	// B independent filter() calls on HOST structure-of-arrays buffers x[nx][B], uDes[nu][B] ->
	// uAct[nu][B], relax[B], rc[B] (reference return codes; untouched slots stay untouched).
	int32_t filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[], int32_t rc[]);

protected:
	int32_t updateConstraints(const double x[]);
	int32_t updateCost(const double uDes[]);
	int32_t updateH(const double H[]);
	void inputSaturate(double u[]);
	int32_t deviceOptions(int asif_hip_model_id, asif_hip_options *o) const;

	const uint32_t nx_, nu_, nv_, npSS_, npSSmax_, nc_;
	SafetySetFn safetySet_;
	DynamicsFn dynamics_;
	Options options_;
	QPWrapperAbstract *QPsolver_;
	std::vector<double> H_, c_, A_, b_, lb_, ub_;
	const double *LfhUser_, *LghUser_; // caller-owned overrides, retained like the reference (src/asif.cpp:137-139)
	asif_hip_multi *batch_;
	int boundModel_; // the compiled model filterBatch() runs, -1 before bindDeviceModel
};

} // namespace ASIF
