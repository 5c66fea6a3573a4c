// qp_kernel.hpp -- one batched-QP kernel body, parameterised by where the problem comes from and
// where the answer goes.  The generic C-ABI entry (k_qp.hip) and the two-stage filters (implicit, TB:
// rows staged in HBM by the trajectory kernel) instantiate it with different policies.
#pragma once
#include <type_traits>
#include <utility>
#include "admm_small.hpp"
#include "launchers.hpp"

namespace asif {

// Rows staged in HBM: A[(nc*nv)][ld] column-major inside an instance, b[nc][ld].
template <int NV, int NC, int G>
__device__ __forceinline__ void load_rows(const double *A, const double *b, int64_t ld, int64_t i, int g,
                                          uint64_t be_mask, QpLaneData<NV, (NC + G - 1) / G> &qp)
{
	constexpr int RPL = (NC + G - 1) / G;
#pragma unroll
	for (int k = 0; k < RPL; k++) {
		const int r = g + k * G;
		const bool valid = r < NC;
		const int rr = valid ? r : 0;
#pragma unroll
		for (int j = 0; j < NV; j++) {
			const double v = A[(int64_t)(rr + j * NC) * ld + i];
			qp.A[k][j] = valid ? v : 0.0;
		}
		const double bv = b[(int64_t)rr * ld + i];
		qp.b[k] = valid ? bv : -1e20; // out-of-range rows are inert: 0.x >= -big
		qp.eq[k] = valid && ((be_mask >> rr) & 1ull);
	}
}

// policies whose load() reads rows staged in HBM declare kStagedRows: their neighbouring workgroups share cache lines
template <class P, class = void>
struct reads_staged_rows : std::false_type {};
template <class P>
struct reads_staged_rows<P, std::enable_if_t<P::kStagedRows>> : std::true_type {};

// policies that declare pending(i) have stage 2 solve only the instances for which it is true (the rows kernel has
// answered the others): a wave none of whose instances is pending exits at once
template <class P, class = void>
struct has_pending : std::false_type {};
template <class P>
struct has_pending<P, std::void_t<decltype(std::declval<const P &>().pending((int64_t)0))>> : std::true_type {};

template <int NV, int NC, int G, class Policy>
__global__ __launch_bounds__(256) void qp_policy_kernel(asif_hip_solver S, Policy pol)
{
	constexpr int RPL = (NC + G - 1) / G;
	// G > 1: a wave covers 64 / G consecutive instances, i.e. 8 * 64 / G bytes of every SoA row -- less than the 128-byte
	// line for G >= 8, so neighbouring workgroups share lines; with the dispatcher's round-robin over the XCDs they sit
	// on different L2s and each fetches the line for itself.  The XCD-contiguous numbering puts them on the same XCD
	// (C3's QP kernel: 44 -> 23 MB, 26.1 -> 25.4 us; C4's 14.2 -> 12.6 us).  Only for policies that read staged rows: a
	// policy that reads 24 bytes per instance has nothing to share, and C5's 7.5 us kernel lost 0.75 us to it.
	constexpr bool kRemap = G > 1 && reads_staged_rows<Policy>::value;
	// (blockDim.x is 64 or 256: a shift -- the 64-bit division by a run-time value is a hundred scalar instructions)
	const int64_t nblk = (pol.B * G + blockDim.x - 1) >> (31 - __clz((int)blockDim.x));
	const int64_t blk = kRemap ? xcd_contiguous_index(blockIdx.x, nblk) : (int64_t)blockIdx.x;
	if (blk >= nblk) return; // wave-uniform: padding block of the XCD-rounded grid
	const int64_t tid = blk * blockDim.x + threadIdx.x;
	const int g = (int)(tid % G);
	int64_t i = tid / G;
	bool live = i < pol.B;
	if (!live) i = pol.B - 1;
	if constexpr (has_pending<Policy>::value) {
		live = live && pol.pending(i);
		if (!__any(live)) return; // wave-uniform
	}
	QpLaneData<NV, RPL> qp;
	pol.template load<NV, NC, G>(i, g, qp);
	AdmmSmall<NV, RPL, G> admm;
	double sol[NV];
	int status, iters;
	admm.solve(qp, S, sol, status, iters, false, S.polish != 1);
	if (live && g == 0) pol.template store<NV>(i, sol, status, iters);
}

template <int NV, int NC, int G, class Policy>
static int launch_policy(const asif_hip_solver &S0, const Policy &pol, hipStream_t stream, int default_scaling = 2)
{
	// independent waves: four per workgroup when the launch is large (launchers.hpp: waves_per_workgroup)
	const int block = 64 * waves_per_workgroup(grid_for(pol.B, G, 64));
	const asif_hip_solver S = resolve_scaling(S0, default_scaling);
	const unsigned nblk = grid_for(pol.B, G, block);
	hipLaunchKernelGGL((qp_policy_kernel<NV, NC, G, Policy>), dim3((G > 1 && reads_staged_rows<Policy>::value) ? xcd_grid(nblk) : nblk), dim3(block), 0,
	                   stream, S, pol);
	return (int)hipGetLastError();
}

} // namespace asif
