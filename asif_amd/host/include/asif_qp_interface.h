// asif_qp_interface.h -- the solver plug-in interface of the reference, for standalone builds.
//
// Inside the reference tree use its own header (include/qpwrapper_abstract.h:16-51); this file
// declares the same class -- same namespace, names, argument order and return conventions -- so that
// ASIF::QPWrapperHip is a drop-in there and buildable here.
//   min x'Hx + c'x   s.t.  A x >= b (== b where be[i]),  lb <= x <= ub
//   all arrays caller-owned, dense column-major; H is nv x nv, A is nc x nv.
#pragma once
#include <cstdint>

namespace ASIF {

class QPWrapperAbstract {
public:
	enum class SOLVER_STATUS : int32_t { INFEASIBLE = 0, FEASIBLE = 1 };

	QPWrapperAbstract(const uint32_t nv, const uint32_t nc, const bool diagonalCost)
	    : nv_(nv), nc_(nc), diagonalCost_(diagonalCost), be_(new bool[nc > 0 ? nc : 1])
	{
		for (uint32_t i = 0; i < nc_; i++) be_[i] = false;
	}
	virtual ~QPWrapperAbstract(void) { delete[] be_; }

	virtual int32_t initialize(const double H[], const double c[], const double A[], const double b[],
	                           const double lb[], const double ub[], const bool be[] = nullptr) = 0;
	virtual int32_t updateCost(const double H[], const double c[]) = 0;
	virtual int32_t updateA(const double A[]) = 0;
	virtual int32_t updateb(const double b[]) = 0;
	virtual int32_t updateBounds(const double lb[], const double ub[]) = 0;
	virtual int32_t solve(void) = 0;
	virtual int32_t getSolution(double sol[]) = 0;

protected:
	const uint32_t nv_;
	const uint32_t nc_;
	const bool diagonalCost_;
	bool *be_;
};

} // namespace ASIF
