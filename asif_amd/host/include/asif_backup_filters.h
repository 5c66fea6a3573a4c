// asif_backup_filters.h -- ASIF::ASIFimplicit and ASIF::ASIFimplicitTB with the reference's public
// interface (include/asif_implicit.h:17-216, include/asif_implicit_tb.h:17-208): constructors taking the
// model callbacks (separate dynamics + dynamicsGradients, or the fused dynamicsWithGradient), initialize,
// the filter overloads, updateOptions and the public diagnostics, on top of ASIF::QPWrapperHip; plus
// filterBatch() on a compiled device model.
//
// Single-agent filter(): the backup trajectory, its sensitivity and the rows are computed on the host
// with the user's std::function callbacks (the callbacks cannot run in a kernel); the QP goes to the GPU
// solver.  filterBatch(): everything on the GPU (asif_hip_filter_batch).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <utility>
#include <vector>
#include "asif_learning.h"
#include "qpwrappers.h"

namespace ASIF {

typedef std::function<void(const double * /*x*/, double * /*h*/, double * /*Dh*/)> SetFn;
typedef std::function<void(const double * /*x*/, double * /*h*/, double * /*Dh*/, double * /*DDh*/)> SetHessFn;
typedef std::function<void(const double * /*x*/, double * /*f*/, double * /*g*/)> DynFn;
typedef std::function<void(const double * /*x*/, double * /*Df*/, double * /*Dg*/)> DynGradFn;
typedef std::function<void(const double * /*x*/, const double * /*u*/, double * /*f*/, double * /*g*/,
                           double * /*d_fcl_dx*/)> DynWithGradFn;
typedef std::function<void(const double * /*x*/, double * /*u*/, double * /*Du*/)> CtrlFn;

// The backup closed loop shared by both classes: soft input saturation, closed-loop vector field with
// its Jacobian, forward-Euler trajectory of [x; vec Q] with the safety set sampled at every point.
class BackupLoopHost {
public:
	typedef std::vector<double> state_t;
	BackupLoopHost(bool hasGradient, uint32_t nx, uint32_t nu, uint32_t npSS, SetFn safetySet, DynFn dynamics,
	               DynGradFn dynamicsGradients, DynWithGradFn dynamicsWithGradient, CtrlFn backupController);

protected:
	void saturateSoft(const double u[], double uSat[], double DuSat[]) const;
	void saturateHard(double u[]) const;
	// t: the time the reference stamps on this rhs; only the held-input class (ASIFimplicitRB) reads it
	void closedLoop(const double x[], double fCL[], double DfCL[], double t = 0.0) const;
	void integrate(const double x[], uint32_t npBT, double dt);   // fills traj_, hAll_, DhAll_, hMin_
	void lowestFirst(std::vector<uint32_t> &order, uint32_t count) const; // ties -> lowest sample index

	const bool hasGradient_;
	const uint32_t nx_, nu_, npSS_;
	SetFn safetySet_;
	DynFn dynamics_;
	DynGradFn dynamicsGradients_;
	DynWithGradFn dynamicsWithGradient_;
	CtrlFn backupController_;
	std::vector<double> lbU_, ubU_;
	double satSharpness_;
	// zero-order hold of the backup input (ASIFimplicitRB: t_last_zoh_, u_zoh_, Du_zoh_,
	// src/asif_implicit_robust.cpp:891-903); holdDt_ <= 0 switches it off
	double holdDt_, holdStep_;
	mutable double tLastHold_;
	mutable std::vector<double> uHold_, DuHold_;
	std::vector<std::pair<double, state_t>> traj_;
	std::vector<double> hAll_, DhAll_, hMin_;
};

class ASIFimplicit : public BackupLoopHost {
public:
	typedef struct {
		double *x0 = nullptr;
		int n_debug = -1;
		double relaxCost = 50.0;
		double relaxReachLb = 5.0;
		double relaxSafeLb = 5.0;
		double backTrajHorizon = 1.0;
		double backTrajDt = 0.01;
		double backTrajAbsTol = 1.0e-6;
		double backTrajRelTol = 1.0e-6;
		double satSharpness = 0.1;
		double inf = 1e20;
		bool use_learning = false; // adds update_weights(learning_data_, ...) to the first row (src/asif_implicit.cpp:585-588)
	} Options;
	typedef std::vector<double> state_t;

	ASIFimplicit(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS, const uint32_t npBTSS,
	             SetFn safetySet, SetFn backupSet, DynFn dynamics, DynGradFn dynamicsGradients, CtrlFn backupController,
	             const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);
	ASIFimplicit(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBS, const uint32_t npBTSS,
	             SetFn safetySet, SetFn backupSet, DynWithGradFn dynamicsWithGradient, CtrlFn backupController,
	             const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);
	~ASIFimplicit(void);

	int32_t initialize(const double lb[], const double ub[]);
	int32_t initialize(const double lb[], const double ub[], const Options &options);
	int32_t filter(const double x[], const double uDes[], double uAct[]);
	int32_t filter(const double x[], const double uDes[], double uAct[], double relax[2]);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[]);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[], double relax[2]);
	int32_t updateOptions(const Options &options);
	static std::string filterErrorMsgString(const int32_t rc);

	// diagnostics, same names as the reference
	std::vector<std::pair<double, state_t>> &backTraj_ = traj_;
	std::vector<uint32_t> backTrajCritIdx_;
	double hBackupEnd_, hSafetyNow_;
	int index_debug_;
	std::vector<double> Dh_index_, h_index_;
	// the rows' ingredients as the class publishes them for the learning pipeline (include/asif_implicit.h:122-124),
	// before the learned residual is added; index arithmetic as upstream (src/asif_implicit.cpp:556-583: entry [i][j]
	// is element nx*i + j of the column-major npTC x nx array Dh, resp. nu*i + j of Lgh -- for nu = 1 that is Lgh of
	// row i; for Dh it is NOT row i's gradient)
	std::vector<double> Lfh_out_;
	std::vector<std::vector<double>> Lgh_out_, Dh_out_;
	LearningData learning_data_; // include/asif_implicit.h:125

	// with options.use_learning the weights of learning_data_ are uploaded too (fill it first)
	int32_t bindDeviceModel(int asif_hip_model_id, int device = 0);
	int32_t bindDeviceModel(int asif_hip_model_id, int32_t ndev, const int32_t devs[]); // one block of the batch per entry
	int32_t filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[], int32_t rc[]);

protected:
	int32_t updateOptions(void);
	int32_t updateConstraints(const double x[]);
	virtual void fillOptions(asif_hip_options &o) const;
	virtual int deviceVariant(void) const { return ASIF_HIP_IMPLICIT; }
	// hook of the derived ASIFimplicitRB: replace the margins of one critical sample (state xs) in place
	virtual void safeMargins(const double xs[], double h[]) const {}
	const uint32_t nv_, npBS_, npBTSS_, npTC_;
	SetFn backupSet_;
	Options options_;
	QPWrapperAbstract *QPsolver_;
	uint32_t npBT_;
	std::vector<double> H_, c_, A_, b_, lb_, ub_;
	asif_hip_multi *batch_;
};

class ASIFimplicitTB : public BackupLoopHost {
public:
	typedef struct {
		double relaxCost = 50.0;
		double relaxSafeLb = 5.0;
		double relaxTTS = 5.0;
		double relaxMinOrtho = 5.0;
		double backTrajHorizon = 1.0;
		double backTrajExtend = 0.05;
		double backTrajDt = 0.01;
		double backTrajMinOrtho = 0.01;
		double backTrajAbsTol = 1.0e-6;
		double backTrajRelTol = 1.0e-6;
		double satSharpness = 0.1;
		double inf = 1e20;
	} Options;
	typedef std::vector<double> state_t;

	ASIFimplicitTB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBTSS, SetFn safetySet,
	               SetHessFn backupSet, DynFn dynamics, DynGradFn dynamicsGradients, CtrlFn backupController,
	               const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);
	ASIFimplicitTB(const uint32_t nx, const uint32_t nu, const uint32_t npSS, const uint32_t npBTSS, SetFn safetySet,
	               SetHessFn backupSet, DynWithGradFn dynamicsWithGradient, CtrlFn backupController,
	               const QPSOLVER qpSolverType = QPSOLVER::OSQP, const bool diagonalCost = true);
	~ASIFimplicitTB(void);

	int32_t initialize(const double lb[], const double ub[]);
	int32_t initialize(const double lb[], const double ub[], const Options &options);
	int32_t filter(const double x[], const double uDes[], double uAct[]);
	int32_t filter(const double x[], const double uDes[], double uAct[], double &relax);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[]);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[], double &relax);
	int32_t updateOptions(const Options &options);
	static std::string filterErrorMsgString(const int32_t rc);

	std::vector<std::pair<double, state_t>> &backTraj_ = traj_;
	std::vector<uint32_t> backTrajCritIdx_;
	double TTS_, BTorthoBS_, hBackupEnd_, hSafetyNow_;

	int32_t bindDeviceModel(int asif_hip_model_id, int device = 0);
	int32_t bindDeviceModel(int asif_hip_model_id, int32_t ndev, const int32_t devs[]); // one block of the batch per entry
	int32_t filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[], int32_t rc[]);

protected:
	int32_t updateOptions(void);
	int32_t updateConstraints(const double x[]);
	int32_t updateConstraintsTrivial(void);
	int32_t solveAndFinish(const double x[], const double H[], const double c[], double uAct[], double &relax,
	                       int32_t okCode, bool leakSolverCode);
	void fillOptions(asif_hip_options &o) const;
	const uint32_t nv_, npBTSS_, npTC_;
	SetHessFn backupSet_;
	Options options_;
	QPWrapperAbstract *QPsolver_;
	uint32_t npBT_;
	bool afterUpdate_;
	std::vector<double> H_, c_, A_, b_, lb_, ub_;
	asif_hip_multi *batch_;
};

} // namespace ASIF
