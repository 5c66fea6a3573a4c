// asif_realizable_filter.h -- class ASIF::ASIFrealizable with the reference's public interface
// (include/asif_realizable.h:9-123): Options, facet_t, kernel_t, the constructor, initialize, the four
// filter overloads, updateOptions and the public diagnostics; plus filterBatch() on a compiled device model.
//
// Single-agent filter(): facet search with the per-facet feasibility QP on facetSolver_ (src/asif_realizable.cpp:
// 381-441), interval Lie derivatives over the critical facets with the host AAF (:445-506), barrier rows
// (:530-600) and the full nc x nv rows A_, b_ exactly as the reference builds them (rowsA()/rowsb()).
// Difference a caller can see only in speed: the QP handed to QPsolver_ is the multiplier-eliminated form of
// those rows (nu == 1: row group s is satisfiable for a given u iff lo(Lgh_s) u + lo(Lfh_s) >= 0 and
// hi(Lgh_s) u + lo(Lfh_s) >= 0), i.e. (u, delta) with 2*npSS + npSSmax rows instead of nv = nu + 4 npSS + 1
// variables -- same (u*, delta*), and a shape the GPU kernels solve to 1e-8 instead of OSQP's 1e-3.
// relax[0] = solutionFull[nu] is a multiplier the QP does not determine (H is zero on it); its smallest
// feasible value max(u*, 0) is reported.
#pragma once
#include <cstdint>
#include <functional>
#include <utility>
#include <vector>
#include "asif_affine.h"
#include "qpwrappers.h"

namespace ASIF {

class ASIFrealizable {
public:
	typedef struct {
		double relaxDes = 5.0;
		double relaxOffset = 5.0;
		double relaxCost = 50.0;
		double inf = 1e20;
	} Options;

	typedef struct {
		std::vector<uint32_t> verticesIdx;
		std::vector<double> normal;
		std::vector<uint32_t> activeConstraintsSet;
		std::vector<interval_t> xFaceInt;
		std::vector<std::pair<double, double>> boundingBox;
	} facet_t;

	typedef struct {
		std::vector<std::vector<double>> vertices;
		std::vector<facet_t> facets;
		uint32_t maxCriticalFacets;
		uint32_t maxActiveConstraints;
	} kernel_t;

	typedef std::function<void(const interval_t * /*x*/, interval_t * /*f*/, interval_t * /*g*/)> DynamicsFn;

	ASIFrealizable(const uint32_t nx, const uint32_t nu, const double uncertaintyBounds[], const kernel_t &kernel,
	               DynamicsFn dynamics, const uint32_t npSSmax = 0, const QPSOLVER qpSolverType = QPSOLVER::OSQP,
	               const bool diagonalCost = true);
	~ASIFrealizable(void);

	int32_t initialize(const double lb[], const double ub[]);
	int32_t initialize(const double lb[], const double ub[], const Options &options);
	int32_t filter(const double x[], const double uDes[], double uAct[]);
	int32_t filter(const double x[], const double uDes[], double uAct[], double relax[2]);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[]);
	int32_t filter(const double x[], const double H[], const double c[], double uAct[], double relax[2]);
	int32_t updateOptions(void);
	int32_t updateOptions(const Options &options);

	// modelData: interval parameters of the compiled device model (mMin..Fhi); the other fields are
	// taken from this object (options, bounds, uncertainty bounds, npSSmax)
	int32_t bindDeviceModel(int asif_hip_model_id, const asif_hip_realizable_options &modelData, int device = 0);
	// SoA host buffers: x[2][B], uDes[1][B] -> uAct[1][B], relax[2][B], rc[B]
	int32_t filterBatch(int64_t B, const double x[], const double uDes[], double uAct[], double relax[], int32_t rc[]);

	const double *rowsA(void) const { return A_.data(); } // last assembled rows (nc x nv, column-major)
	const double *rowsb(void) const { return b_.data(); }
	uint32_t nv(void) const { return nv_; }
	uint32_t nc(void) const { return nc_; }

protected:
	int32_t updateConstraints(const double x[]);

	const uint32_t nx_, nu_;
	std::vector<double> uncertaintyBounds_;
	kernel_t kernel_;
	const uint32_t nFacets_, npSS_, npSSmax_, nv_, nc_;
	DynamicsFn dynamics_;
	Options options_;
	QPWrapperAbstract *QPsolver_;
	QPWrapperAbstract *facetSolver_;
	std::vector<double> H_, c_, A_, b_, lb_, ub_; // the reference's QP, handed to QPsolver_ as it is (:300-340)
	std::vector<double> A_facet_, b_facet_;
	asif_hip_ctx *batch_;
	asif_hip_realizable_options batchModel_;
	asif_hip_realizable_options batchOpts(void) const;

public:
	std::vector<uint32_t> criticalFacets_;
	uint32_t nCriticalFacets_;
	std::vector<uint32_t> criticalBarrierFacets_;
	std::vector<double> hBarrier_;
	std::vector<double> DhBarrier_;
};

} // namespace ASIF
