// qpwrapper_hip.h -- ASIF::QPWrapperHip: the reference's solver plug-in slot filled by the MI355X kernels
// (replaces ASIF::QPWrapperOsqp, include/qpwrapper_osqp.h:9-61 / src/qpwrapper_osqp.cpp).
// One solve() = asif_hip_qp_solve_batch_warm with a batch of one.  `warmStart = true` (or ASIF_HIP_QP_WARM=1 in the
// environment) makes the second and later solve() calls start from the previous call's iterate and multipliers, as an
// OSQP workspace does (warm_start = 1 by default, which the reference's wrapper leaves on), where the problem runs on
// a wave-level kernel (nv > 3: the lifted problems of ASIFrobust / ASIFrealizable, full cost matrices).  It is OFF by
// default: the method's cold start ends in 4 Newton steps on the lifted robust problem and 9 on the realizable ones
// with |u - u_ref| at 4e-9 ... 1e-7, and a warm one gains where the cost is strictly convex (full H 20 x 30: 18 -> 7
// steps) or the loop moves little (ASIFrobust's example loop: 3.7 -> 0.9 steps, 42 -> 38 us per filter()), not on a
// state that moved at random (18 x 12: 4.3 -> 7.0; 86 x 65: 9.5 -> 14) -- DESIGN 4.4, tests/test_gpu_qp_warm.py.  Any shape the
// reference's classes construct a solver with is accepted (up to 128 variables / 128 rows within 160 KB of LDS:
// ASIFrealizable's 86 x 65 included), diagonalCost = true or false.
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
	// what solve() returns when the call itself failed (no device, launch error): OSQP's "unsolved" value, so that
	// it can never be mistaken for a solver verdict; lastError() then holds the asif_hip / HIP error code
	static constexpr int32_t STATUS_UNSOLVED = -10;

	QPWrapperHip(const uint32_t nv, const uint32_t nc, const bool diagonalCost, int device = 0);
	virtual ~QPWrapperHip(void);

	// 0 on success (like osqp_setup's exit flag, src/qpwrapper_osqp.cpp:121), otherwise an asif_hip error (shape
	// beyond the kernels, no device).  Like the reference (src/asif.cpp:101-105) it solves once and ignores the
	// verdict of that solve: an infeasible problem at x0 is not a set-up failure.
	virtual int32_t initialize(const double H[], const double c[], const double A[], const double b[],
	                           const double lb[], const double ub[], const bool be[] = nullptr);
	// nullptr = unchanged (src/qpwrapper_osqp.cpp:128,157,199,209); always return 1
	virtual int32_t updateCost(const double H[], const double c[]);
	virtual int32_t updateA(const double A[]);
	virtual int32_t updateb(const double b[]);
	virtual int32_t updateBounds(const double lb[], const double ub[]);
	// 1 (FEASIBLE) or the raw OSQP-style status (src/qpwrapper_osqp.cpp:225-238): solver verdicts only
	virtual int32_t solve(void);
	virtual int32_t getSolution(double sol[]);

	asif_hip_solver settings; // kernel settings (defaults from asif_hip_default_solver)
	bool warmStart;           // OSQPSettings::warm_start; false by default here (see above)
	int32_t lastIterations(void) const { return iters_; }
	int lastError(void) const { return error_; } // 0, a negative ASIF_HIP_E* code or a positive hipError_t

private:
	int device_;
	double *host_; // pinned: [H (nv or nv*nv) | c | A | b | lb | ub | sol | status, iterations | warm x | warm y]
	bool haveWarm_; // the warm block holds what a solve() of this problem's shape left there
	std::vector<uint8_t> be8_;
	double *dev_;  // the device-side address of that same block (zero copy: the kernel reads and writes it in place)
	void *stream_;
	int32_t status_, iters_;
	int error_;
	size_t nH() const { return diagonalCost_ ? (size_t)nv_ : (size_t)nv_ * nv_; }
	size_t offH() const { return 0; }
	size_t offC() const { return nH(); }
	size_t offA() const { return offC() + nv_; }
	size_t offB() const { return offA() + (size_t)nc_ * nv_; }
	size_t offLb() const { return offB() + nc_; }
	size_t offUb() const { return offLb() + nv_; }
	size_t offSol() const { return offUb() + nv_; }
	size_t offStatus() const { return offSol() + nv_; } // two int32 in one double slot
	size_t offWarmX() const { return offStatus() + 1; }
	size_t offWarmY() const { return offWarmX() + nv_; }
	size_t total() const { return offWarmY() + nc_ + nv_; }
	int setup(void);
};

} // namespace ASIF
