// k_qp.hip -- batched solve of pre-assembled QPs: the QPWrapperAbstract path
// initialize -> solve -> getSolution (include/qpwrapper_abstract.h:30-43; src/qpwrapper_osqp.cpp:55-261),
// cold start, one QP per lane group, SoA in HBM (component-major, instance-minor) so that the G=1
// mapping reads whole 512-byte lines per wave instruction.
#include "qp_kernel.hpp"
#include "admm_wave.hpp"
#include "qp_lds.hpp"
#include "qp_inv.hpp"
#include <cstdlib>

namespace asif {

int launch_qp_wave(const asif_hip_solver &S, const QpArgs &a, hipStream_t stream);
int launch_qp_lds(const asif_hip_solver &S, const QpArgs &a, hipStream_t stream);

struct GenericQpPolicy {
	static constexpr bool kStagedRows = true; // load() reads rows from HBM (qp_kernel.hpp: XCD-contiguous blocks)
	int64_t B, ld;
	const double *Hd, *c, *A, *b, *lb, *ub;
	uint64_t be_mask;
	double *sol;
	int32_t *status, *iters;

	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
#pragma unroll
		for (int j = 0; j < NV; j++) {
			qp.Hd[j] = Hd[j * ld + i];
			qp.c[j] = c[j * ld + i];
			qp.lb[j] = lb[j * ld + i];
			qp.ub[j] = ub[j * ld + i];
		}
		load_rows<NV, NC, G>(A, b, ld, i, g, be_mask, qp);
	}
	template <int NV>
	__device__ __forceinline__ void store(int64_t i, const double (&x)[NV], int st, int it) const
	{
#pragma unroll
		for (int j = 0; j < NV; j++) sol[j * ld + i] = x[j];
		status[i] = st;
		if (iters) iters[i] = it;
	}
};

// Same, for a shape padded up to a compiled one: rows nc..NC-1 are inert (0.x >= -big).
struct PaddedQpPolicy : GenericQpPolicy {
	int nc;

	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
		constexpr int RPL = (NC + G - 1) / G;
#pragma unroll
		for (int j = 0; j < NV; j++) {
			qp.Hd[j] = Hd[j * ld + i];
			qp.c[j] = c[j * ld + i];
			qp.lb[j] = lb[j * ld + i];
			qp.ub[j] = ub[j * ld + i];
		}
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			const int r = g + k * G;
			const bool valid = r < nc;
			const int rr = valid ? r : 0;
#pragma unroll
			for (int j = 0; j < NV; j++) {
				const double v = A[(int64_t)(rr + j * nc) * ld + i];
				qp.A[k][j] = valid ? v : 0.0;
			}
			const double bv = b[(int64_t)rr * ld + i];
			qp.b[k] = valid ? bv : -1e20;
			qp.eq[k] = valid && ((be_mask >> rr) & 1ull);
		}
	}
};

int launch_qp_small(const asif_hip_solver &S, const QpArgs &a, hipStream_t stream)
{
	if (a.B <= 0) return 0;
	const GenericQpPolicy p = {a.B, a.ld, a.Hd, a.c, a.A, a.b, a.lb, a.ub, a.be_mask, a.sol, a.status, a.iters};
	const int G = S.lanes_per_qp;
	if (a.H || a.nc > 64) return launch_qp_lds(S, a, stream);   // full cost matrix / more rows than the masks of the
	                                                             // in-register kernels hold: the LDS kernel
	if (G == 64) return launch_qp_wave(S, a, stream); // explicitly ask for a wave-per-QP kernel
	// shapes of the filter classes at the configs' sizes (C2 2x4, C3 3x41, C4 2x18)
	if (a.nv == 2 && a.nc == 4) {
		if (G == 0 || G == 1) return launch_policy<2, 4, 1>(S, p, stream);
		if (G == 2) return launch_policy<2, 4, 2>(S, p, stream);
		if (G == 4) return launch_policy<2, 4, 4>(S, p, stream);
		return ASIF_HIP_EINVAL;
	}
	if (a.nv == 2 && a.nc == 18) {
		if (G == 0 || G == 2) return launch_policy<2, 18, 2>(S, p, stream);
		if (G == 1) return launch_policy<2, 18, 1>(S, p, stream);
		if (G == 4) return launch_policy<2, 18, 4>(S, p, stream);
		return ASIF_HIP_EINVAL;
	}
	if (a.nv == 3 && a.nc == 41) {
		if (G == 0 || G == 4) return launch_policy<3, 41, 4>(S, p, stream);
		if (G == 8) return launch_policy<3, 41, 8>(S, p, stream);
		if (G == 16) return launch_policy<3, 41, 16>(S, p, stream);
		return ASIF_HIP_EINVAL;
	}
	// other two-variable shapes (the realizable filter's facet problem 2x5 and its eliminated QP, 2x20..2x44)
	// ride on a padded in-register kernel, so that they get the active-set finish as well
	if (a.nv == 2 && a.nc >= 1 && a.nc <= 48 && G == 0) {
		PaddedQpPolicy pp;
		static_cast<GenericQpPolicy &>(pp) = p;
		pp.nc = a.nc;
		if (a.nc <= 8) return launch_policy<2, 8, 1>(S, pp, stream);
		return launch_policy<2, 48, 4>(S, pp, stream);
	}
	return launch_qp_wave(S, a, stream);
}

template <int NVMAX>
static int launch_wave(const asif_hip_solver &S0, const QpArgs &a, hipStream_t stream)
{
	// plain ADMM (no active-set finish) leans on the equilibration for its convergence rate and for what
	// the unscaled tolerances mean: never fewer than four Ruiz passes here
	asif_hip_solver S = S0;
	if (S.scaling_iters < 4) S.scaling_iters = 4;
	hipLaunchKernelGGL((qp_wave_kernel<NVMAX>), dim3(xcd_grid(a.B)), dim3(64), 0, stream, S, a);
	return (int)hipGetLastError();
}

// One wavefront per QP.  polish == 0 asks for the plain OSQP-style ADMM of admm_wave.hpp (nv <= 32, nc + nv <= 64,
// diagonal cost); everything else goes to the LDS kernel of qp_lds.hpp.
int launch_qp_wave(const asif_hip_solver &S, const QpArgs &a, hipStream_t stream)
{
	if (a.B <= 0) return 0;
	if (a.B > 0x7fffffffLL) return ASIF_HIP_EUNSUPPORTED;
	if (S.polish == 0 && !a.H && a.nv >= 1 && a.nv <= 32 && a.nc + a.nv <= 64) {
		int r;
		if (a.nv <= 8) r = launch_wave<8>(S, a, stream);
		else if (a.nv <= 16) r = launch_wave<16>(S, a, stream);
		else r = launch_wave<32>(S, a, stream);
		if (r) return r;
		// What the iterations leave at max_iter is not a verdict.  The problems the robust / realizable classes lift
		// (multipliers without cost: LP-like, degenerate) keep ~0.5 % of their instances there at any practical budget,
		// with iterates 1e-2 away from the optimum, where OSQP at its 1e-3 tolerances would have reported "solved".
		// Second launch, same stream: exactly those instances go through the exact method of qp_lds.hpp; every other
		// workgroup leaves at once.  Status and solution then equal the oracle's (tests/test_gpu_qp_generic.py).
		QpArgs b = a;
		b.only_status = kStatusMaxIter;
		b.iters = nullptr; // keep the iteration counts of the first pass
		return launch_qp_lds(S, b, stream);
	}
	return launch_qp_lds(S, a, stream);
}

template <int VPT, int RPT, bool FULLH>
static int launch_lds(const asif_hip_solver &S0, const QpArgs &a_in, hipStream_t stream)
{
	asif_hip_solver S = S0;
	if (S.scaling_iters == 0) S.scaling_iters = 4;
	else if (S.scaling_iters < 0) S.scaling_iters = 0;
	QpArgs a = a_in;
	size_t bytes = lds_doubles(a.nv, a.nc, VPT, RPT, FULLH) * sizeof(double);
	if (bytes > 160 * 1024) return ASIF_HIP_EUNSUPPORTED;
	{
		// K_J kept between Newton steps where the shape leaves room for it (qp_lds.hpp, build<true>); ASIF_HIP_QP_KEEP_KJ=0:
		// developer switch, the full sum at every change of the active set
		static const bool off = []() {
			const char *v = getenv("ASIF_HIP_QP_KEEP_KJ");
			return v && v[0] == '0';
		}();
		const size_t with = lds_doubles(a.nv, a.nc, VPT, RPT, FULLH, true) * sizeof(double);
		if (!off && with <= 160 * 1024) {
			a.keep_kj = 1;
			bytes = with;
		}
	}
	auto kern = qp_lds_kernel<VPT, RPT, FULLH>;
	if (a.warm_x) kern = qp_lds_kernel<VPT, RPT, FULLH, true>;
	if (bytes > 48 * 1024) {
		hipError_t e = allow_dynamic_lds((const void *)kern, bytes);
		if (e != hipSuccess) return (int)e;
	}
	hipLaunchKernelGGL(kern, dim3(xcd_grid(a.B)), dim3(64), bytes, stream, S, a);
	return (int)hipGetLastError();
}

// nv <= 32, nc <= 32, diagonal cost: two QPs per wave, the inverse of K_J kept by rank-one steps (qp_inv.hpp).
// Shapes are padded to the next compiled size <NVMAX, NCMAX>.
#ifndef ASIF_INV_TWO_WAVES_MIN
#define ASIF_INV_TWO_WAVES_MIN 16384
#endif
constexpr int64_t kInvTwoWavesMin = ASIF_INV_TWO_WAVES_MIN;
template <int NVMAX, int NCMAX, int HW = 32>
static int launch_inv(const asif_hip_solver &S0, const QpArgs &a, hipStream_t stream)
{
	asif_hip_solver S = S0;
	if (S.scaling_iters == 0) S.scaling_iters = 4;
	else if (S.scaling_iters < 0) S.scaling_iters = 0;
	constexpr int QPW = 64 / HW;
	const size_t bytes = QPW * inv_half_doubles(NVMAX, NCMAX, HW) * sizeof(double);
	auto kern = qp_inv_kernel<NVMAX, NCMAX, HW>;
	// <18, 12>, <14, 10>, <10, 6> and <8, 16> need 253 / 229 / 187 / 251 registers as they are: two waves fit a SIMD
	// without the tighter allocation, whose code is slower at every batch size (18 x 12: 250 against 295 us per 16 384,
	// 753 against 897 per 65 536)
	constexpr bool fits_two = (NVMAX == 18 && NCMAX == 12) || (NVMAX == 8 && NCMAX == 16) || (NVMAX == 14 && NCMAX == 10) ||
	                          (NVMAX == 10 && NCMAX == 6) || (NVMAX == 6 && NCMAX == 4);
	if constexpr (HW == 32 && !fits_two) {
		if (a.B >= kInvTwoWavesMin) kern = qp_inv_kernel<NVMAX, NCMAX, HW, 2>; // eight waves' worth of problems per SIMD: qp_inv.hpp, MINW
	}
	if (a.warm_x) kern = qp_inv_kernel<NVMAX, NCMAX, HW, 1, true>; // closed loops: batches of one to a few thousand
	if (bytes > 48 * 1024) {
		hipError_t e = allow_dynamic_lds((const void *)kern, bytes);
		if (e != hipSuccess) return (int)e;
	}
	hipLaunchKernelGGL(kern, dim3(xcd_grid((a.B + QPW - 1) / QPW)), dim3(64), bytes, stream, S, a);
	return (int)hipGetLastError();
}
template <int NVMAX>
static int launch_inv_nc(const asif_hip_solver &S, const QpArgs &a, hipStream_t stream)
{
	return a.nc <= 16 ? launch_inv<NVMAX, 16>(S, a, stream) : launch_inv<NVMAX, 32>(S, a, stream);
}

// any shape with nv <= 128 and nc <= 128 whose LDS footprint fits 160 KB (86 x 65 with a full cost matrix: 148 KB)
int launch_qp_lds(const asif_hip_solver &S, const QpArgs &a, hipStream_t stream)
{
	if (a.B <= 0) return 0;
	if (a.nv < 1 || a.nv > 128 || a.nc < 0 || a.nc > 128 || a.B > 0x7fffffffLL) return ASIF_HIP_EUNSUPPORTED;
	static const bool inv_off = []() { // developer switch: ASIF_HIP_QP_INV=0 sends the small shapes to the wave-per-QP kernel
		const char *v = getenv("ASIF_HIP_QP_INV");
		return v && v[0] == '0';
	}();
	if (!a.H && a.nv <= 32 && a.nc <= 32 && !inv_off) {
		// The lifted problem of ASIFrobust is (2 + 4 N) x 3 N for N safety functions (src/asif_robust.cpp:21-22): each N the
		// library carries (up to ASIF_HIP_MAX_HALFPLANES = 8; N = 8 on the whole-wave kernel below)
		// has the kernel at its own size, rounded to even.  The padded grid below made 18 x 12 (BASELINE's C5 lifted) walk
		// two zero columns and four zero rows in every product -- a tenth / a quarter of each loop, and the loops are the
		// kernel -- and sent 26 x 18 to <32, 32>.  Zeros added to the same two chains: the same bits
		// (tests/test_gpu_qp_lds.py).  ASIF_HIP_QP_INV_EXACT=0: developer switch, the padded sizes.
		static const bool exact_off = []() {
			const char *v = getenv("ASIF_HIP_QP_INV_EXACT");
			return v && v[0] == '0';
		}();
		if (!exact_off) {
			if (a.nv == 6 && a.nc <= 4) return launch_inv<6, 4>(S, a, stream);     // N = 1: 6 x 3
			if (a.nv == 10 && a.nc <= 6) return launch_inv<10, 6>(S, a, stream);   // N = 2
			if (a.nv == 14 && a.nc <= 10) return launch_inv<14, 10>(S, a, stream); // N = 3: 14 x 9
			if (a.nv == 18 && a.nc == 12) return launch_inv<18, 12>(S, a, stream); // N = 4
			if (a.nv == 22 && a.nc <= 16) return launch_inv<22, 16>(S, a, stream); // N = 5: 22 x 15 (DoubleIntegrator_Robust)
			if (a.nv == 26 && a.nc <= 18) return launch_inv<26, 18>(S, a, stream); // N = 6
			if (a.nv == 30 && a.nc <= 22) return launch_inv<30, 22>(S, a, stream); // N = 7: 30 x 21
		}
		if (a.nv <= 8) return launch_inv_nc<8>(S, a, stream);
		if (a.nv <= 20) return launch_inv_nc<20>(S, a, stream); // ASIFrobust with four safety functions: 18 x 12
		if (a.nv <= 24) return launch_inv_nc<24>(S, a, stream); // five: 22 x 15
		return launch_inv_nc<32>(S, a, stream);
	}
	// 32 < nv or nc <= 64, diagonal cost: the same kernel with the whole wave on one problem (ASIFrealizable's lifted
	// problems: 38 x 29 on the 100 Hz kernel, 62 x 47 on the 50-point one)
	if (!a.H && a.nv <= 64 && a.nc <= 64 && !inv_off) {
		static const bool exact_off = []() { // (ASIF_HIP_QP_INV_EXACT=0: the padded sizes, as above)
			const char *v = getenv("ASIF_HIP_QP_INV_EXACT");
			return v && v[0] == '0';
		}();
		if (a.nv == 34 && a.nc <= 24 && !exact_off) return launch_inv<34, 24, 64>(S, a, stream); // ASIFrobust, N = 8
		if (a.nv == 38 && a.nc <= 30 && a.nc > 16 && !exact_off) return launch_inv<38, 30, 64>(S, a, stream); // ASIFrealizable, 100 Hz kernels: 38 x 29
		if (a.nv <= 40) return a.nc <= 32 ? launch_inv<40, 32, 64>(S, a, stream) : launch_inv<40, 64, 64>(S, a, stream);
		return a.nc <= 48 ? launch_inv<64, 48, 64>(S, a, stream) : launch_inv<64, 64, 64>(S, a, stream);
	}
	const bool small = a.nv <= 64 && a.nc <= 64;
	if (a.H) return small ? launch_lds<1, 1, true>(S, a, stream) : launch_lds<2, 2, true>(S, a, stream);
	return small ? launch_lds<1, 1, false>(S, a, stream) : launch_lds<2, 2, false>(S, a, stream);
}

} // namespace asif
