// k_explicit.hip -- fused explicit CBF filter: ASIF::filter (src/asif.cpp:176-210) for a batch.
//
// One launch does, per instance: safetySet + dynamics (model functors), Lfh = Dh f, Lgh = Dh g
// (src/asif.cpp:279-284), rows A = [Lgh | h], b = -Lfh (:295-303), the QP
//   min (u-uDes)'(u-uDes) + relaxCost (delta-relaxLb)^2   s.t. rows, lb<=u<=ub, delta pinned (:84-98)
// solved in registers (dual active-set stage of gi_small.hpp, which decides this class's one-variable problem on its
// own; admm_small.hpp's iterations for the other solver modes), then inputSaturate (:343-352) and the return code
// (:199-209).
// HBM traffic is the algorithmic minimum: 8(nx+nu) bytes in, 8(nu+1)+4 bytes out per instance, SoA,
// consecutive lanes -> consecutive instances (G = 1) so every load/store is a fully coalesced line.
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include "admm_small.hpp"
#include "launchers.hpp"

namespace asif {

// PRE: the instantiation for problems the dual active-set stage is known to decide on its own (one input, default
// solver mode or asif_hip_solver::presolve): no ADMM / finish code, hence none of their register footprint.
// SEL: the optional paths of src/asif.cpp (npSSmax < npSS row selection, caller-supplied Lie derivatives) are
// compiled in; the default instantiation (every row, the model's own Lie derivatives) does not carry them.
template <class M, int G, bool PRE, bool SEL = false, bool WHOLE = false>
__device__ __forceinline__ void explicit_filter_body(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a,
                                                     bool assemble_only)
{
	constexpr int NX = M::NX, NU = M::NU, NP = M::NPSS, NV = NU + 1, NC = NP;
	constexpr int RPL = (NC + G - 1) / G;
	const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int g = (int)(tid % G);
	int64_t i = tid / G;
	const bool live = i < a.B;
	if (!live) i = a.B - 1; // keep the lane in the group reductions; it never stores

	double x[NX], uDes[NU];
#pragma unroll
	for (int k = 0; k < NX; k++) x[k] = a.x[k * a.ld + i];
#pragma unroll
	for (int k = 0; k < NU; k++) uDes[k] = a.udes[k * a.ld + i];

	double h[NP], Dh[NP * NX], f[NX], gm[NX * NU];
	M::safetySet(o, x, h, Dh);
	M::dynamics(o, x, f, gm);
	double Lfh[NP], Lgh[NP * NU];
#pragma unroll
	for (int r = 0; r < NP; r++) {
		double s = 0.0;
#pragma unroll
		for (int k = 0; k < NX; k++) s += Dh[r + k * NP] * f[k];
		Lfh[r] = s;
#pragma unroll
		for (int j = 0; j < NU; j++) {
			double t = 0.0;
#pragma unroll
			for (int k = 0; k < NX; k++) t += Dh[r + k * NP] * gm[k + j * NX];
			Lgh[r + j * NP] = t;
		}
	}
	// npSSmax < npSS: only the rows of the smallest h are kept, in ascending order of h (src/asif.cpp:250-268);
	// pos[r] = position of safety function r in that order (ties: lower index first), kept iff pos[r] < nkeep
	const int nkeep = SEL ? o.npKeep : NP;
	int pos[NP];
#pragma unroll
	for (int r = 0; r < NP; r++) pos[r] = r;
	if (SEL && nkeep < NP) { // wave-uniform
#pragma unroll
		for (int r = 0; r < NP; r++) {
			int p = 0;
#pragma unroll
			for (int q = 0; q < NP; q++) p += (h[q] < h[r] || (h[q] == h[r] && q < r)) ? 1 : 0;
			pos[r] = p;
		}
	}
	if (SEL && a.lfh) { // caller-supplied Lie derivatives, indexed by row position (src/asif.cpp:287-292)
#pragma unroll
		for (int r = 0; r < NP; r++) {
			const int p = pos[r] < nkeep ? pos[r] : 0;
			Lfh[r] = a.lfh[(int64_t)p * a.ld + i];
#pragma unroll
			for (int j = 0; j < NU; j++) Lgh[r + j * NP] = a.lgh[(int64_t)(p + j * nkeep) * a.ld + i];
		}
	}
	if (assemble_only) {
		if (live && g == 0) {
#pragma unroll
			for (int r = 0; r < NP; r++) {
				if (pos[r] < nkeep) {
					const int p = pos[r];
#pragma unroll
					for (int j = 0; j < NU; j++) a.A[(int64_t)(p + j * nkeep) * a.ld + i] = Lgh[r + j * NP];
					a.A[(int64_t)(p + NU * nkeep) * a.ld + i] = h[r];
					a.b[(int64_t)p * a.ld + i] = -Lfh[r];
				}
			}
			a.code[i] = 1;
			if (a.diag) // which safety function sits in each row
				for (int r = 0; r < NP; r++)
					if (pos[r] < nkeep) a.diag[(int64_t)pos[r] * a.ld + i] = (double)r;
		}
		return;
	}

	QpLaneData<NV, RPL> qp;
#pragma unroll
	for (int j = 0; j < NU; j++) {
		qp.Hd[j] = 1.0;
		qp.c[j] = -2.0 * uDes[j];
		qp.lb[j] = o.lb[j];
		qp.ub[j] = o.ub[j];
	}
	qp.Hd[NU] = o.relaxCost;
	qp.c[NU] = -2.0 * o.relaxCost * o.relaxLb;
	qp.lb[NU] = o.relaxLb;
	qp.ub[NU] = o.relaxLb; // src/asif.cpp:91: the explicit class pins the relaxation variable
#pragma unroll
	for (int k = 0; k < RPL; k++) {
		// row r = g + k*G of the NC rows; out-of-range rows are inert (0.x >= -big)
#pragma unroll
		for (int j = 0; j < NV; j++) qp.A[k][j] = 0.0;
		qp.b[k] = -1e20;
		qp.eq[k] = false;
#pragma unroll
		for (int r = 0; r < NC; r++)
			if (r == g + k * G && pos[r] < nkeep) {
#pragma unroll
				for (int j = 0; j < NU; j++) qp.A[k][j] = Lgh[r + j * NP];
				qp.A[k][NU] = h[r];
				qp.b[k] = -Lfh[r];
			}
	}
	if constexpr (PRE) {
		static_assert(NU == 1 && G == 1, "one input: the pinned relaxation variable leaves a one-variable problem");
		// The explicit class pins its relaxation variable (lb = ub = relaxLb, src/asif.cpp:88-91): the dual active-set
		// stage eliminates it and what is left has one variable, which that stage always decides (gi_small.hpp:
		// solve_1d -- optimal or infeasible, never "undecided").  This instantiation therefore carries neither the ADMM
		// iterations nor the finish: 40 VGPRs instead of 256 + AGPRs, eight waves per SIMD instead of one.
		double sol[NV];
		int steps;
		const int verdict = GiSmall<NV, RPL, G>::template solve_with_pinned<NU>(qp, g, 8 * NV + 4, sol, steps);
		if constexpr (WHOLE) {
			// whole-line stores (see launch_explicit_di): a failed lane stores back what it finds in its slot
			const bool ok = verdict == kGiOptimal;
			double u = fmin(fmax(sol[0], o.lb[0]), o.ub[0]), rl = sol[NU];
			if (__any(live & !ok)) {
				if (live & !ok) {
					u = a.uact[i];
					rl = a.relax[i];
				}
			}
			if (live) {
				a.uact[i] = u;
				a.relax[i] = rl;
				a.rc[i] = ok ? ASIF_HIP_RC_OK : ASIF_HIP_RC_QP_FAILED;
			}
		} else if (live) {
			if (verdict == kGiOptimal) {
				a.uact[i] = fmin(fmax(sol[0], o.lb[0]), o.ub[0]);
				a.relax[i] = sol[NU];
				a.rc[i] = ASIF_HIP_RC_OK;
			} else {
				a.rc[i] = ASIF_HIP_RC_QP_FAILED; // uAct and relax stay untouched, src/asif.cpp:208-209
			}
		}
		if (live) {
			if (a.diag) {
				a.diag[0 * a.ld + i] = 0.0;
				a.diag[1 * a.ld + i] = 0.0;
				a.diag[(a.ndiag - 1) * a.ld + i] = 0.0;
			}
		}
		return;
	}

	AdmmSmall<NV, RPL, G> admm;
	double sol[NV];
	int status, iters;
	admm.solve(qp, S, sol, status, iters, false, S.polish != 1);

	if (live && g == 0) {
		if (status == kStatusSolved) {
#pragma unroll
			for (int j = 0; j < NU; j++) a.uact[j * a.ld + i] = fmin(fmax(sol[j], o.lb[j]), o.ub[j]);
			a.relax[i] = sol[NU];
			a.rc[i] = ASIF_HIP_RC_OK;
		} else {
			a.rc[i] = ASIF_HIP_RC_QP_FAILED; // uAct and relax stay untouched, src/asif.cpp:208-209
		}
		if (a.diag) {
			a.diag[0 * a.ld + i] = (double)admm.stat_rounds;
			a.diag[1 * a.ld + i] = (double)admm.stat_farkas;
			a.diag[(a.ndiag - 1) * a.ld + i] = (double)iters;
		}
	}
}

template <class M, int G, bool PRE, bool SEL = false>
__global__ __launch_bounds__((PRE ? 256 : 64), (G >= 2 ? 2 : 1)) void explicit_filter_kernel(DevOptions o, asif_hip_solver S, FilterArgs a,
                                                             bool assemble_only)
{
	explicit_filter_body<M, G, PRE, SEL>(o, S, a, assemble_only);
}

// The light instantiation takes the handful of options the explicit class reads in a 56-byte block of its own: a
// launch that carries the 2 KB DevOptions structure in its argument block costs the host 3.9 us against 2.8 us for a
// small one (tools/scratch/launch_cost.hip) -- more than this kernel runs -- and costs a replayed graph 0.25 us per
// node; a pointer to a device copy would cost the kernel a dependent load before its first instruction.
// batch size from which the whole-line stores take over: where the 44 B per instance of one launch no longer fit the
// 256 MiB Infinity Cache (below it the partially written lines are merged in cache; measured)
constexpr long long kWholeLinesFrom = 6291456;
struct ExplicitOpts {
	double lb[ASIF_HIP_MAX_NU], ub[ASIF_HIP_MAX_NU], relaxCost, relaxLb;
	int npKeep;
};
template <class M, bool SEL, bool WHOLE = false>
__global__ __launch_bounds__(256) void explicit_light_kernel(ExplicitOpts e, FilterArgs a)
{
	static_assert(M::kIgnoresOptions, "the model's functors must not read DevOptions: only the class's fields are passed");
	DevOptions o = {};
#pragma unroll
	for (int j = 0; j < ASIF_HIP_MAX_NU; j++) {
		o.lb[j] = e.lb[j];
		o.ub[j] = e.ub[j];
	}
	o.relaxCost = e.relaxCost;
	o.relaxLb = e.relaxLb;
	o.npKeep = e.npKeep;
	const asif_hip_solver none = {};
	explicit_filter_body<M, 1, true, SEL, WHOLE>(o, none, a, false);
}

// Closed loop, T control steps per launch (the caller's side of filter(): examples/DoubleIntegrator.cpp:81-116).
// Per step:  rc = filter(x, uDes, uAct, relax)  exactly as above (cold start, like every batched call);  then the
// plant's forward-Euler step  fCl = 0 + f + uAct*g,  x += dt*fCl  (:96-110) with the dynamics at the state the
// filter saw.  When the filter fails uAct keeps its previous value -- the example never resets it (:89-94).
// The state stays in registers for the whole rollout; HBM traffic is 8(nx+2nu+1)+4 bytes in and out per
// instance per LAUNCH (plus the optional logs), not per step.
// LIGHT: as explicit_filter_kernel's PRE -- one input, the dual active-set stage decides every step's QP on its own
// (relaxation variable eliminated, one variable left); no ADMM / finish code in the instantiation.
template <class M, bool LIGHT = false>
__global__ __launch_bounds__(LIGHT ? 256 : 64) void explicit_rollout_kernel(DevOptions o, asif_hip_solver S, RolloutArgs a)
{
	constexpr int NX = M::NX, NU = M::NU, NP = M::NPSS, NV = NU + 1, NC = NP;
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = i < a.B;
	if (!live) i = a.B - 1;
	const int64_t ld = a.ld;
	double x[NX], uDes[NU], uAct[NU], relax;
#pragma unroll
	for (int k = 0; k < NX; k++) x[k] = a.x[k * ld + i];
#pragma unroll
	for (int k = 0; k < NU; k++) {
		uDes[k] = a.udes[k * ld + i];
		uAct[k] = a.uact[k * ld + i];
	}
	relax = a.relax[i];
	int nfail = 0;
	// one object for the rollout: it carries the working set from step to step (the light instantiation has none)
	std::conditional_t<LIGHT, int, AdmmSmall<NV, NC, 1>> admm{};
#pragma unroll 1
	for (int t = 0; t < a.T; t++) { // wave-uniform trip count
		double h[NP], Dh[NP * NX], f[NX], gm[NX * NU];
		M::safetySet(o, x, h, Dh);
		M::dynamics(o, x, f, gm);
		QpLaneData<NV, NC> qp;
#pragma unroll
		for (int r = 0; r < NP; r++) {
			double s = 0.0;
#pragma unroll
			for (int k = 0; k < NX; k++) s += Dh[r + k * NP] * f[k];
#pragma unroll
			for (int j = 0; j < NU; j++) {
				double tt = 0.0;
#pragma unroll
				for (int k = 0; k < NX; k++) tt += Dh[r + k * NP] * gm[k + j * NX];
				qp.A[r][j] = tt;
			}
			qp.A[r][NU] = h[r];
			qp.b[r] = -s;
			qp.eq[r] = false;
		}
		if (o.npKeep < NP) { // wave-uniform; npSSmax < npSS (src/asif.cpp:250-268): rows beyond the npKeep smallest h are inert
#pragma unroll
			for (int r = 0; r < NP; r++) {
				int p = 0;
#pragma unroll
				for (int q = 0; q < NP; q++) p += (h[q] < h[r] || (h[q] == h[r] && q < r)) ? 1 : 0;
				const bool kept = p < o.npKeep;
#pragma unroll
				for (int j = 0; j < NV; j++) qp.A[r][j] = kept ? qp.A[r][j] : 0.0;
				qp.b[r] = kept ? qp.b[r] : -1e20;
			}
		}
#pragma unroll
		for (int j = 0; j < NU; j++) {
			qp.Hd[j] = 1.0;
			qp.c[j] = -2.0 * uDes[j];
			qp.lb[j] = o.lb[j];
			qp.ub[j] = o.ub[j];
		}
		qp.Hd[NU] = o.relaxCost;
		qp.c[NU] = -2.0 * o.relaxCost * o.relaxLb;
		qp.lb[NU] = o.relaxLb;
		qp.ub[NU] = o.relaxLb;
		double sol[NV];
		int status, iters;
		if constexpr (LIGHT) {
			static_assert(NU == 1, "one input: the pinned relaxation variable leaves a one-variable problem");
			const int verdict = GiSmall<NV, NC, 1>::template solve_with_pinned<NU>(qp, 0, 8 * NV + 4, sol, iters);
			status = verdict == kGiOptimal ? kStatusSolved : kStatusPrimalInf;
		} else {
			admm.solve(qp, S, sol, status, iters, S.warm_start != 0 && t > 0, S.polish != 1);
		}
		if (live && a.xlog) {
#pragma unroll
			for (int k = 0; k < NX; k++) a.xlog[((int64_t)t * NX + k) * ld + i] = x[k];
		}
		const bool ok = status == kStatusSolved;
		if (ok) {
#pragma unroll
			for (int j = 0; j < NU; j++) uAct[j] = fmin(fmax(sol[j], o.lb[j]), o.ub[j]);
			relax = sol[NU];
		} else nfail++;
		if (live && a.ulog) {
#pragma unroll
			for (int j = 0; j < NU; j++) a.ulog[((int64_t)t * NU + j) * ld + i] = uAct[j];
		}
		if (live && a.rclog) a.rclog[(int64_t)t * ld + i] = ok ? ASIF_HIP_RC_OK : ASIF_HIP_RC_QP_FAILED;
		{
#pragma clang fp contract(off)
#pragma unroll
			for (int k = 0; k < NX; k++) {
				double fcl = 0.0;
				fcl += f[k];
#pragma unroll
				for (int j = 0; j < NU; j++) fcl += uAct[j] * gm[k + j * NX];
				x[k] += a.dt * fcl;
			}
		}
	}
	if (!live) return;
#pragma unroll
	for (int k = 0; k < NX; k++) a.x[k * ld + i] = x[k];
#pragma unroll
	for (int j = 0; j < NU; j++) a.uact[j * ld + i] = uAct[j];
	a.relax[i] = relax;
	a.nfail[i] = nfail;
}

int launch_rollout_explicit_di(const DevOptions &o, const asif_hip_solver &S0, const RolloutArgs &a, hipStream_t stream)
{
	if (a.B <= 0 || a.T <= 0) return 0;
	const asif_hip_solver S = resolve_scaling(S0, 1, 1);
	if (S.polish == 2 || S.presolve) // the dual active-set stage alone decides this class's one-input problems
		hipLaunchKernelGGL((explicit_rollout_kernel<DoubleIntegrator, true>), dim3(grid_for(a.B, 1, 256)), dim3(256), 0,
		                   stream, o, S, a);
	else
		hipLaunchKernelGGL((explicit_rollout_kernel<DoubleIntegrator, false>), dim3(grid_for(a.B, 1, 64)), dim3(64), 0,
		                   stream, o, S, a);
	return (int)hipGetLastError();
}

template <class M, int G>
static int launch_g(const DevOptions &o, const asif_hip_solver &S0, const FilterArgs &a, bool assemble_only,
                    hipStream_t stream)
{
	const int block = 64;
	// one Ruiz pass by default: the rows are well scaled and a second pass only costs finish rounds
	const asif_hip_solver S = resolve_scaling(S0, 1, 1);
	const bool sel = o.npKeep < M::NPSS || a.lfh != nullptr;
	if (sel)
		hipLaunchKernelGGL((explicit_filter_kernel<M, G, false, true>), dim3(grid_for(a.B, G, block)), dim3(block), 0,
		                   stream, o, S, a, assemble_only);
	else
		hipLaunchKernelGGL((explicit_filter_kernel<M, G, false, false>), dim3(grid_for(a.B, G, block)), dim3(block), 0,
		                   stream, o, S, a, assemble_only);
	return (int)hipGetLastError();
}

int launch_explicit_di(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream)
{
	if (a.B <= 0) return 0;
	// default solver mode (dual active-set stage first) or presolve: that stage alone decides every instance of this
	// class with one input -- the light instantiation, one instance per lane, larger blocks
	if ((S.presolve || (S.polish == 2 && (S.lanes_per_qp == 0 || S.lanes_per_qp == 1))) && !assemble_only) {
		const int block = 256;
		const bool sel = o.npKeep < DoubleIntegrator::NPSS || a.lfh != nullptr;
		ExplicitOpts e;
		for (int j = 0; j < ASIF_HIP_MAX_NU; j++) {
			e.lb[j] = o.lb[j];
			e.ub[j] = o.ub[j];
		}
		e.relaxCost = o.relaxCost;
		e.relaxLb = o.relaxLb;
		e.npKeep = o.npKeep;
		// From ~1 M instances the kernel is bound by memory, and what binds it is the STORES: uAct / relax are written
		// only where the solve succeeded (src/asif.cpp:208-209 leaves them untouched otherwise), and lines written under a
		// partial lane mask cost the memory system a read-modify-write -- 3.9 TB/s against 6.0 TB/s for the same columns
		// stored whole (tools/scratch/stream_pattern.hip, profiles/r03/stream_pattern_microbench.txt; 16-byte accesses
		// change neither figure).  The whole-line instantiation reads the old values of the failed lanes itself (only in
		// waves that hold one) and stores every lane: 16 M instances 195 -> 175 us.  Below the switch-over the batch lives
		// in the caches, the partial lines are merged there, and the extra dependent load only costs (1 M: 10.4 -> 12.0 us).
		static const int64_t whole_from = []() {
			const char *v = getenv("ASIF_HIP_WHOLE_LINES_FROM"); // developer override of the switch-over batch size
			return v ? (int64_t)atoll(v) : (int64_t)kWholeLinesFrom;
		}();
		if (!sel && a.B >= whole_from) {
			hipLaunchKernelGGL((explicit_light_kernel<DoubleIntegrator, false, true>), dim3(grid_for(a.B, 1, block)), dim3(block), 0,
			                   stream, e, a);
			return (int)hipGetLastError();
		}
		if (sel)
			hipLaunchKernelGGL((explicit_light_kernel<DoubleIntegrator, true>), dim3(grid_for(a.B, 1, block)), dim3(block), 0,
			                   stream, e, a);
		else
			hipLaunchKernelGGL((explicit_light_kernel<DoubleIntegrator, false>), dim3(grid_for(a.B, 1, block)), dim3(block), 0,
			                   stream, e, a);
		return (int)hipGetLastError();
	}
	switch (S.lanes_per_qp) {
	case 2: return launch_g<DoubleIntegrator, 2>(o, S, a, assemble_only, stream);
	case 4: return launch_g<DoubleIntegrator, 4>(o, S, a, assemble_only, stream);
	case 0:
	case 1: return launch_g<DoubleIntegrator, 1>(o, S, a, assemble_only, stream);
	default: return ASIF_HIP_EINVAL;
	}
}

// class ASIF on the synthetic two-input model: nv = 3, five rows
int launch_explicit_p2(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream)
{
	if (a.B <= 0) return 0;
	if (S.lanes_per_qp > 1) return ASIF_HIP_EINVAL;
	return launch_g<PlanarTwoInput, 1>(o, S, a, assemble_only, stream);
}

} // namespace asif
