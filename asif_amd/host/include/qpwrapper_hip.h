// qpwrapper_hip.h -- ASIF::QPWrapperHip: the reference's solver plug-in slot filled by the MI355X
// in-kernel ADMM (replaces ASIF::QPWrapperOsqp, include/qpwrapper_osqp.h:9-61 / src/qpwrapper_osqp.cpp).
// One solve() = asif_hip_qp_solve_batch with a batch of one; cold start every call.
#pragma once
#if __has_include("qpwrapper_abstract.h")
#include "qpwrapper_abstract.h" // building inside the reference tree
#else
#include "asif_qp_interface.h"
#endif
#include "asif_hip.h"
#include <cstddef>
#include <vector>

namespace ASIF {

class QPWrapperHip : public QPWrapperAbstract {
public:
	QPWrapperHip(const uint32_t nv, const uint32_t nc, const bool diagonalCost, int device = 0);
	virtual ~QPWrapperHip(void);

	// 0 on success (like osqp_setup's exit flag, src/qpwrapper_osqp.cpp:121), otherwise an asif_hip error
	virtual int32_t initialize(const double H[], const double c[], const double A[], const double b[],
	                           const double lb[], const double ub[], const bool be[] = nullptr);
	// nullptr = unchanged (src/qpwrapper_osqp.cpp:128,157,199,209); always return 1
	virtual int32_t updateCost(const double H[], const double c[]);
	virtual int32_t updateA(const double A[]);
	virtual int32_t updateb(const double b[]);
	virtual int32_t updateBounds(const double lb[], const double ub[]);
	// 1 (FEASIBLE) or the raw OSQP-style status (src/qpwrapper_osqp.cpp:225-238)
	virtual int32_t solve(void);
	virtual int32_t getSolution(double sol[]);

	asif_hip_solver settings; // in-kernel ADMM settings (defaults from asif_hip_default_solver)
	int32_t lastIterations(void) const { return iters_; }

private:
	int device_;
	std::vector<double> host_; // [Hd | c | A | b | lb | ub] staged contiguously
	std::vector<uint8_t> be8_;
	std::vector<double> sol_;
	double *dev_;
	int32_t *devStatus_;
	int32_t status_, iters_;
	bool dirty_;
	size_t offHd() const { return 0; }
	size_t offC() const { return nv_; }
	size_t offA() const { return 2 * (size_t)nv_; }
	size_t offB() const { return offA() + (size_t)nc_ * nv_; }
	size_t offLb() const { return offB() + nc_; }
	size_t offUb() const { return offLb() + nv_; }
	size_t offSol() const { return offUb() + nv_; }
	size_t total() const { return offSol() + nv_; }
};

} // namespace ASIF
