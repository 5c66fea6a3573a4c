// launchers.hpp -- host-side launch functions, one per kernel family (each lives in its own .hip TU).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "asif_hip.h"
#include "models.hpp"

namespace asif {

struct FilterArgs {
	int64_t B, ld;
	const double *x, *udes;
	double *uact, *relax;
	int32_t *rc;
	double *diag; // [ndiag][ld] or nullptr
	int ndiag;
	// assemble-only mode: rows out, no solve
	double *A, *b;
	int32_t *code;
	// implicit / implicit-RB: block checkpoints of the backup trajectory, [ceil(npBT / kTrajBlock)][nx + nx*nx + 2][ld]
	double *ckpt;
	// class ASIF with caller-supplied Lie derivatives (src/asif.cpp:287-292): lfh[nc][ld], lgh[nc*nu][ld], or nullptr
	const double *lfh, *lgh;
	// ASIFimplicit's rows as the filter's own two launches pass them: [Lgh_r, h_r] in columns 0 and 1 of the staging
	// block and -Lfh_r in b -- three doubles per row.  The third column of the reference's nc x 3 block is zero on the
	// safety rows and holds h on the backup row, whose second is zero (src/asif_implicit.cpp:591-611): structure the
	// solver's loader restores instead of 8 nc bytes per instance written and read back.  0: the reference's full block
	// (asif_hip_assemble_batch hands that out).
	int compactRows;
	// ASIFimplicitTB, default solver mode: the rows kernel solves each instance's QP itself (k_tb.hip) and stages nothing;
	// stage 2 runs for the instances it marks pending in `code`
	int fuseQp;
};

// explicit CBF filter (class ASIF), model = DoubleIntegrator / PlanarTwoInput
int launch_explicit_di(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream);
int launch_explicit_p2(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream);

// closed loop: T x (explicit filter + plant Euler step) per launch, model = DoubleIntegrator
struct RolloutArgs {
	int64_t B, ld;
	int T;
	double dt;
	double *x;          // [nx][ld] in: initial state, out: state after T steps
	const double *udes; // [nu][ld], held over the rollout
	double *uact;       // [nu][ld] in: input applied if the first filter call fails, out: last applied input
	double *relax;      // [1][ld] in/out
	int32_t *nfail;     // [B] number of steps whose filter call failed
	double *xlog;       // [T][nx][ld] state seen by each filter call, or nullptr
	double *ulog;       // [T][nu][ld] input applied after each filter call, or nullptr
	int32_t *rclog;     // [T][ld], or nullptr
};
int launch_rollout_explicit_di(const DevOptions &o, const asif_hip_solver &S, const RolloutArgs &a, hipStream_t stream);
// plant Euler step x += dt (f(x) + g(x) uact) between two filter calls of the two-stage filters (k_rollout.hip);
// logs the state / input / rc of the call just made (nullptr to skip) and counts rc < 0 into nfail
int launch_plant_step(int model, const DevOptions &o, int64_t B, int64_t ld, double dt, double *x, const double *uact,
                      const int32_t *rc, int32_t *nfail, double *xlog, double *ulog, int32_t *rclog, hipStream_t s);

// implicit backup-trajectory filter (class ASIFimplicit), model = InvertedPendulum.
// Filter mode needs a.A / a.b to point at staging rows of (nc*nv + nc) * ld doubles.
// rb: class ASIFimplicitRB on the same model (held backup input, interval margins, learned residual).
int launch_implicit_ip(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream, bool rb = false);

// same class, model = DoubleIntegratorImplicit (examples/DoubleIntegrator_implicit.cpp, npBTSS = 4)
int launch_implicit_di(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream, bool rb = false);

// time-to-backup-set filter (class ASIFimplicitTB), model = Segway.
// Filter mode additionally needs a.code to point at B staged int32 branch codes.
int launch_tb_segway(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                     hipStream_t stream);

// same class, model = InvertedPendulumTB (examples/InvertedPendulum_ImplicitTB.cpp)
int launch_tb_pendulum(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream);

// same class, model = DoubleIntegratorTB (examples/DoubleIntegrator_implicit_tb.cpp)
int launch_tb_di(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only, hipStream_t stream);

// robust explicit filter (class ASIFrobust), model = InvertedPendulumRobust (half-plane safety set)
int launch_robust_ip(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                     hipStream_t stream);

// realizable filter (class ASIFrealizable), model = DoubleIntegratorSampled; k_realizable.hip
// facet record: va[2] (segment end with the smaller first coordinate), extents a0, |a1|, sign of a1, pad,
// normal[2], bounding box {lo,hi}[2]
constexpr int kRzRec = 12;
constexpr int kRzPoint = 8; // state-independent constants of the model's point-state dynamics
struct RzDev {
	const double *facetRec; // [nF][kRzRec], device
	const double *table;    // [nF][nA][4] = lo(Lgh), hi(Lgh), lo(Lfh), hi(Lfh), device
	const double *pointC;   // [kRzPoint], device
	int nF, nA, maxCrit, npSSmax, npSS, nv, nc;
	double unc[2];
	double relaxDes, relaxOffset, relaxCost, inf, lb, ub;
	double mMin, mMax, Klo, Khi, Flo, Fhi;
};
int launch_realizable_tables(const RzDev &z, const double *vertices, const int32_t *fverts, const double *normals,
                             const int32_t *factive, double *facetRec, double *table, double *pointC,
                             int32_t *overflow, hipStream_t stream);
int launch_realizable(const RzDev &z, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                      hipStream_t stream);

// robust filter (class ASIFrobust) on a half-plane data set, model = DoubleIntegratorRobust; k_robust_data.hip
constexpr int kRbPoint = 12; // state-independent constants of the point-state interval dynamics
struct RbDev {
	const double *hp;     // [N][2] half-planes 1 - a.x >= 0, device
	const double *pointC; // [kRbPoint], device
	int N, npSSmax;
	double relaxCost, relaxLb, inf, lb, ub;
	double mMin, mMax, Klo, Khi, Flo, Fhi;
};
int launch_robust_data_point(const RbDev &z, double *pointC, hipStream_t stream);
int launch_robust_data(const RbDev &z, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream);

// One workgroup per QP, QPs stored structure-of-arrays: component k of QP q sits at base[k * ld + q], so eight
// consecutive QPs share every 64-byte line of the problem data.  Workgroups are dealt round-robin to the eight XCDs
// (each with its own L2): with q = blockIdx the eight QPs of a line land on eight different L2s and every line is
// fetched eight times (PMC: 163 MB for 21 MB of problems).  This mapping gives each XCD a contiguous range of QPs,
// so the workgroups that share a line run on the same XCD, next to one another in time.
constexpr int kXcds = 8;
__host__ __device__ inline unsigned xcd_grid(int64_t B) { return (unsigned)(kXcds * ((B + kXcds - 1) / kXcds)); }
__device__ __forceinline__ int64_t xcd_contiguous_index(unsigned block, int64_t B)
{
	const int64_t per = (B + kXcds - 1) / kXcds;
	return (int64_t)(block % kXcds) * per + block / kXcds; // may be >= B in the last range: that workgroup exits
}

struct QpArgs {
	int64_t B, ld;
	int nv, nc;
	const double *Hd, *c, *A, *b, *lb, *ub;
	uint64_t be_mask; // equality flags of rows 0..63
	double *sol;
	int32_t *status, *iters;
	const double *H;   // full cost matrix [nv*nv][ld] (column-major, upper triangle read) or nullptr: Hd is used
	uint64_t be_mask2; // rows 64..127
	int only_status = 0; // != 0: a second pass -- only instances whose status[] holds this value are solved (again)
	int keep_kj = 0;     // qp_lds.hpp: the shape's LDS has room for the unfactored K_J between Newton steps (set by the launcher)
	// asif_hip_qp_solve_batch_warm (wave-level kernels only): the iterate and the multipliers of [A; I] in the caller's
	// units, x [nv][ld] and y [nc + nv][ld] -- written at the end of every solve (zeros unless the verdict is "solved"),
	// read as the starting point when warm_in != 0: OSQP's warm_start = 1 between two solve() calls of one workspace
	double *warm_x = nullptr, *warm_y = nullptr;
	int warm_in = 0;
};
// pre-assembled QPs; returns ASIF_HIP_EUNSUPPORTED for shapes without a compiled kernel
int launch_qp_small(const asif_hip_solver &S, const QpArgs &a, hipStream_t stream);

// asif_hip_solver::scaling_iters: 0 = the path's default number of Ruiz passes, negative = no scaling;
// check_interval: 0 = the path's default
inline asif_hip_solver resolve_scaling(const asif_hip_solver &S, int path_default, int check_default = 2)
{
	asif_hip_solver r = S;
	if (r.scaling_iters == 0) r.scaling_iters = path_default;
	else if (r.scaling_iters < 0) r.scaling_iters = 0;
	// without a finish a check only tests residuals: OSQP-ish period instead of the finish paths' 1-2 iterations
	if (r.check_interval <= 0) r.check_interval = r.polish == 0 ? 10 : check_default;
	return r;
}

// Waves per workgroup for kernels whose waves are independent of one another (no LDS sharing, no barrier).  One-wave
// workgroups are the natural unit, but the dispatcher places a workgroup's waves on different SIMDs of a CU and makes no
// such promise ACROSS workgroups: from three one-wave workgroups per CU on, some SIMDs get two waves while others get
// none, and kernels that run one latency-bound wave per SIMD lose 30-50 % (C12 at 65 536 instances: 372 us at one
// workgroup per CU, 538 us at four; C3 at 65 536: 1 388 us, 968 us as four-wave workgroups).  So: four waves
// per workgroup once the launch has more than two waves per CU.
// Dynamic LDS above the 48 KB a kernel may ask for by default: the limit is raised to the whole 160 KB of a CU, never to
// the size of this launch -- the attribute belongs to the kernel, not to the launch, and two host threads driving two
// handles on one device (asif_hip_filter_batch_host_multi) may launch the same kernel with different sizes: whoever sets
// the smaller one last must not pull the limit under the other's launch.
constexpr size_t kLdsPerCu = 160 * 1024;
inline hipError_t allow_dynamic_lds(const void *kern, size_t bytes)
{
	if (bytes <= 48 * 1024) return hipSuccess;
	return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
}

inline int device_cus()
{
	static const int cus = []() {
		int dev = 0, n = 256;
		if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
		return n > 0 ? n : 256;
	}();
	return cus;
}
inline int waves_per_workgroup(int64_t waves)
{
	static const int forced = []() {
		const char *e = getenv("ASIF_HIP_WG_WAVES"); // developer switch: 1, 2 or 4
		const int v = e ? atoi(e) : 0;
		return (v == 1 || v == 2 || v == 4) ? v : 0;
	}();
	if (forced) return forced;
	return waves > (int64_t)2 * device_cus() ? 4 : 1;
}

inline int grid_for(int64_t B, int G, int block)
{
	const int64_t threads = B * G;
	return (int)((threads + block - 1) / block);
}

} // namespace asif
