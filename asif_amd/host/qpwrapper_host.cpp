// qpwrapper_host.cpp -- see include/qpwrapper_host.h.  The arithmetic is gi_small.hpp's (one lane per QP, every group
// operation the identity); this file copies the caller's arrays in and maps the verdict to the reference's codes.
#include "qpwrappers.h"
#include "asif_hip.h"
#include "gi_small.hpp"
#include <cstring>

namespace ASIF {

namespace hostqp {
int solve_alm(int nv, int nc, bool diag, const double *H, const double *c, const double *A, const double *b,
              const double *lb, const double *ub, const bool *be, double eps_rel, int max_newton, double *sol, int *newton_out,
              double *warm_x, double *warm_y, bool warm_in);
}

namespace {

// row capacities the method is instantiated for (rows beyond nc are inert: 0 . x >= -1e20, as on the device)
template <int NV, int NC>
int run(const uint32_t nc, const double *Hd, const double *c, const double *A, const double *b, const double *lb,
        const double *ub, const bool *be, double *sol, int32_t &steps)
{
	asif::QpLaneData<NV, NC> qp;
	for (int j = 0; j < NV; j++) {
		qp.Hd[j] = Hd[j];
		qp.c[j] = c[j];
		qp.lb[j] = lb[j];
		qp.ub[j] = ub[j];
	}
	for (int r = 0; r < NC; r++) {
		const bool valid = (uint32_t)r < nc;
		for (int j = 0; j < NV; j++) qp.A[r][j] = valid ? A[r + (size_t)j * nc] : 0.0; // column-major nc x nv
		qp.b[r] = valid ? b[r] : -1e20;
		qp.eq[r] = valid && be[r];
	}
	double x[NV];
	int st = 0;
	// step budget as the kernels': 8 nv + 4 working-set changes (k_explicit.hip, admm_small.hpp)
	const int verdict = asif::GiSmall<NV, NC, 1>::solve(qp, 0, 8 * NV + 4, x, st);
	for (int j = 0; j < NV; j++) sol[j] = x[j];
	steps = st;
	return verdict;
}

template <int NV>
int run_nv(const uint32_t nc, const double *Hd, const double *c, const double *A, const double *b, const double *lb,
           const double *ub, const bool *be, double *sol, int32_t &steps)
{
	if (nc <= 8) return run<NV, 8>(nc, Hd, c, A, b, lb, ub, be, sol, steps);
	if (nc <= 24) return run<NV, 24>(nc, Hd, c, A, b, lb, ub, be, sol, steps);
	return run<NV, 64>(nc, Hd, c, A, b, lb, ub, be, sol, steps);
}

} // namespace

QPWrapperHost::QPWrapperHost(const uint32_t nv, const uint32_t nc, const bool diagonalCost)
    : QPWrapperAbstract(nv, nc, diagonalCost), H_(activeSet(nv, nc, diagonalCost) ? 0 : (size_t)nv * nv, 0.0), Hd_(nv, 0.0), c_(nv, 0.0), A_((size_t)nc * nv, 0.0), b_(nc, 0.0),
      lb_(nv, 0.0), ub_(nv, 0.0), sol_(nv, 0.0), warmX_(nv, 0.0), warmY_((size_t)nc + nv, 0.0), status_(-10), steps_(0), ready_(false)
{
}

QPWrapperHost::~QPWrapperHost(void) {}

int32_t QPWrapperHost::initialize(const double H[], const double c[], const double A[], const double b[],
                                  const double lb[], const double ub[], const bool be[])
{
	if (!supports(nv_, nc_, diagonalCost_)) return ASIF_HIP_EUNSUPPORTED;
	if (be != nullptr)
		for (uint32_t i = 0; i < nc_; i++) be_[i] = be[i];
	ready_ = true;
	haveWarm_ = false; // osqp_setup: a fresh workspace, its first solve is cold
	updateCost(H, c);
	updateA(A);
	updateb(b);
	updateBounds(lb, ub);
	(void)solve(); // the reference solves once at the end of initialize() and ignores the outcome (src/asif.cpp:101-105)
	return 0;
}

int32_t QPWrapperHost::updateCost(const double H[], const double c[])
{
	if (H != nullptr) {
		for (uint32_t i = 0; i < nv_; i++) Hd_[i] = H[i + (size_t)i * nv_]; // the diagonal, src/qpwrapper_osqp.cpp:267-272
		if (!H_.empty()) std::memcpy(H_.data(), H, sizeof(double) * nv_ * nv_); // (upper triangle read when !diagonalCost_)
	}
	if (c != nullptr) std::memcpy(c_.data(), c, sizeof(double) * nv_);
	return 1;
}

int32_t QPWrapperHost::updateA(const double A[])
{
	std::memcpy(A_.data(), A, sizeof(double) * nc_ * nv_);
	return 1;
}

int32_t QPWrapperHost::updateb(const double b[])
{
	std::memcpy(b_.data(), b, sizeof(double) * nc_);
	return 1;
}

int32_t QPWrapperHost::updateBounds(const double lb[], const double ub[])
{
	if (lb != nullptr) std::memcpy(lb_.data(), lb, sizeof(double) * nv_);
	if (ub != nullptr) std::memcpy(ub_.data(), ub, sizeof(double) * nv_);
	return 1;
}

int32_t QPWrapperHost::solve(void)
{
	if (!ready_) return (status_ = -10); // OSQP's "unsolved": solve() before initialize()
	if (!activeSet(nv_, nc_, diagonalCost_)) {
		int nw = 0;
		status_ = hostqp::solve_alm((int)nv_, (int)nc_, diagonalCost_, H_.data(), c_.data(), A_.data(), b_.data(), lb_.data(),
		                            ub_.data(), be_, epsRel, maxNewton, sol_.data(), &nw, warmX_.data(), warmY_.data(),
		                            warmStart && haveWarm_);
		haveWarm_ = true;
		steps_ = nw;
		return status_;
	}
	int verdict;
	switch (nv_) {
	case 1: verdict = run_nv<1>(nc_, Hd_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data(), be_, sol_.data(), steps_); break;
	case 2: verdict = run_nv<2>(nc_, Hd_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data(), be_, sol_.data(), steps_); break;
	default: verdict = run_nv<3>(nc_, Hd_.data(), c_.data(), A_.data(), b_.data(), lb_.data(), ub_.data(), be_, sol_.data(), steps_); break;
	}
	// QPWrapperOsqp::solve (src/qpwrapper_osqp.cpp:225-238): FEASIBLE, or the raw OSQP status
	status_ = verdict == asif::kGiOptimal ? 1 : (verdict == asif::kGiInfeasible ? -3 : -2);
	return status_;
}

int32_t QPWrapperHost::getSolution(double sol[])
{
	for (uint32_t i = 0; i < nv_; i++) sol[i] = sol_[i];
	return 1;
}

QPWrapperAbstract *makeQPWrapper(const QPSOLVER type, const uint32_t nv, const uint32_t nc, const bool diagonalCost)
{
	if (type == QPSOLVER::HOST && QPWrapperHost::supports(nv, nc, diagonalCost)) return new QPWrapperHost(nv, nc, diagonalCost);
	return new QPWrapperHip(nv, nc, diagonalCost);
}

} // namespace ASIF
