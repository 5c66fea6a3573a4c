// qp_lds.hpp -- one wavefront per QP, the dense reduced KKT LDL' factor, the constraint block and the iterate
// vectors in LDS: the general solver behind asif_hip_qp_solve_batch / QPWrapperHip for every shape the
// in-register kernels do not cover -- up to 128 variables and 128 general rows (dynamic LDS, <= 160 KB), cost
// matrix diagonal or full.  Covers what the reference's own classes hand to QPWrapperOsqp
// (src/qpwrapper_osqp.cpp:55-261): ASIFrobust 18 x 12 / 22 x 15 (src/asif_robust.cpp:21-22), ASIFrealizable
// 38 x 29, 62 x 47, 86 x 65 (src/asif_realizable.cpp:19-22), custom non-diagonal H (src/asif.cpp:167-174 ->
// src/qpwrapper_osqp.cpp:276-309).
//
// Method.  OSQP's ADMM (kept in admm_wave.hpp, asif_hip_solver::polish == 0) is an augmented-Lagrangian method
// whose inner minimisation is replaced by one alternating step; on the multiplier-lifted problems above (zero
// cost on most variables: LP-like, degenerate) that step needs thousands of iterations and its iterates never
// pin an active set.  Here the inner minimisation is done EXACTLY instead:
//     min_x  1/2 x'Px + q'x + 1/(2 gamma) |x - xhat|^2 + sum_i mu_i/2 dist^2(a_i x + y_i/mu_i, [l_i,u_i])
// is piecewise quadratic, strictly convex and C^1; a semismooth Newton iteration with the generalised Hessian
//     K_J = P + I/gamma + sum_{i in J} mu_i a_i a_i'        J = rows currently outside their interval
// and an exact line search (the derivative along the step is piecewise linear) ends in finitely many steps.
// Outer loop: y_i <- mu_i (s_i - proj(s_i)), xhat <- x (proximal method of multipliers); duals of inactive rows
// are exactly zero.  Termination on the scaled residuals (relative 1e-10 by default); primal infeasibility by
// OSQP's certificate test on the dual increment.  K_J is the same nv x nv matrix OSQP-style ADMM factors
// (P + sigma I + A' diag(rho) A) with row weights that follow the active set -- same LDS factor, same kernel
// layout, refactored when J changes.
//
// Layout.  Lane t owns variables t, t+64 (VPT = 1 or 2) and general rows t, t+64 (RPT): their bounds, duals,
// penalties and scalings in VGPRs.  LDS: At[nv][RS] the scaled row block transposed (RS odd: the column walk
// of lane-owned variables and the row walk of lane-owned rows are both conflict-free), S[nv][NVP] the factor
// stored square and symmetric (row k holds column k of L as well, so the forward and the backward substitution
// both read contiguous lanes), P packed (full H only), broadcast vectors.  Substitutions keep the right-hand
// side in registers and broadcast pivots with v_readlane.  Bound rows (the identity block the wrapper appends,
// src/qpwrapper_osqp.cpp:319-343) are never stored: they only touch the diagonal.
#pragma once
#include "admm_small.hpp"
#include "launchers.hpp"

namespace asif {

constexpr double kLdsGamma = 1e7;     // proximal weight 1/gamma on |x - xhat|^2: weak -- the multiplier iteration, not the
                                      // proximal term, is what should pace the outer loop (1e4 cost 3.5x the Newton steps;
                                      // 1e8 leaves stragglers of 80 steps on near-singular K_J)
constexpr double kLdsMu0 = 10.0;      // initial penalty; equalities 100x
constexpr double kLdsMuMax = 1e4;     // cap: beyond it the rounding of mu (s - proj s) sets the residual floor
// The multiplier iteration contracts the primal residual by rho = 1 / (1 + mu sigma) per outer step (sigma: the
// problem's own constant).  When a step contracts by less than 10x the penalties are raised -- not by a fixed factor
// of ten per outer step (round 3: the hard 5 % of the lifted robust problems climbed 10 -> 1e4 over three outer
// steps, each with its Newton steps and a rebuild of the inverse) but by the factor that the OBSERVED contraction says
// is needed for rho = kLdsRhoTarget, at least ten, the cap as before: sigma = (1 / rho - 1) / mu from the step just
// made.  numpy prototype on 1 200 seeded 18 x 12 problems against the exact oracle: Newton steps of the hardest
// instance 21 -> 15, of the 99th percentile 18 -> 13, mean 4.49 -> 4.34, same verdicts, same 3.9e-9; jumping after
// the FIRST outer step (no contraction observed yet), a higher cap (1e5: stragglers at the rounding floor) and an
// active-set finish from the multipliers' support (degenerate: 17 near-dependent rows at the boundary) were tried
// there and are not taken.
constexpr double kLdsRhoTarget = 1e-3;
__host__ __device__ inline double penalty_jump(double pri, double pri_prev)
{
	const double rho = pri < 0.999 * pri_prev ? pri / pri_prev : 0.999;
	const double f = (1.0 / kLdsRhoTarget - 1.0) / (1.0 / rho - 1.0);
	return f < 10.0 ? 10.0 : (f > 1e6 ? 1e6 : f);
}
constexpr int kLdsMaxDeltas = 16; // rank-one updates of the kept K_J between two full sums (LdsQp::build)
constexpr int kLdsMaxOuter = 80;
constexpr int kLdsMaxInner = 60;

__host__ __device__ inline int lds_rs(int nc) { return (nc < 1 ? 1 : nc) | 1; }
__host__ __device__ inline int lds_nvp(int nv) { return nv | 1; }
// doubles of dynamic LDS for a shape
__host__ __device__ inline size_t lds_doubles(int nv, int nc, int vpt, int rpt, bool fullh, bool kp = false)
{
	size_t n = (size_t)nv * lds_rs(nc) + (size_t)nv * lds_nvp(nv);
	if (fullh) n += (size_t)nv * (nv + 1) / 2;
	if (kp) n += (size_t)nv * (nv + 1) / 2; // the unfactored K_J kept between Newton steps (LdsQp::build<true>)
	n += 64 * vpt * 2;                 // va, dinv
	n += 64 * rpt;                     // vr
	n += 5 * (64 * (size_t)(rpt + vpt)); // line-search view of all rows: s, dl, l, u, mu
	n += 2 + 2 * 64 * (size_t)(rpt + vpt); // line-search evaluation points: 0, 1 and the breakpoints
	n += 32 * rpt;                     // active-row list (int32)
	return n;
}

// The two rows of 16 of every group of 32, side by side: a = the even row's value in both rows, b = the odd row's
// (v_permlane16_swap_b32, new on gfx950: a vector instruction where ds_swizzle went through the LDS path and a
// wait for it, in every one of the dozen reductions of a Newton step).  rows32 likewise for the two halves of the wave.
#ifndef ASIF_INV_SWIZZLE
__device__ __forceinline__ void rows16(double v, double &a, double &b)
{
	const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
	const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
	const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
	a = __hiloint2double((int)rh[0], (int)rl[0]);
	b = __hiloint2double((int)rh[1], (int)rl[1]);
}
#else
__device__ __forceinline__ void rows16(double v, double &a, double &b) // (the ds_swizzle form, for comparison builds)
{
	const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
	const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
	const double o = __hiloint2double(hi, lo);
	const bool odd = (threadIdx.x & 16) != 0;
	a = odd ? o : v;
	b = odd ? v : o;
}
#endif
__device__ __forceinline__ void rows32(double v, double &a, double &b)
{
	const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
	const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
	const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
	a = __hiloint2double((int)rh[0], (int)rl[0]);
	b = __hiloint2double((int)rh[1], (int)rl[1]);
}
// whole-wave reductions, the xor butterfly: four DPP steps inside the rows of 16, then the rows of a group of 32, then
// the two groups (the same pairs in the same order as six __shfl_xor stages, whose two widest went through ds_bpermute)
#ifdef ASIF_LDS_SHFL // comparison builds: the __shfl_xor form
__device__ __forceinline__ double wmax(double v)
{
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
	return v;
}
__device__ __forceinline__ double wmin(double v)
{
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) v = fmin(v, __shfl_xor(v, m, 64));
	return v;
}
__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
	return v;
}
#else
__device__ __forceinline__ double wmax(double v)
{
	v = fmax(v, dpp_xchg<1>(v));
	v = fmax(v, dpp_xchg<2>(v));
	v = fmax(v, dpp_xchg<4>(v));
	v = fmax(v, dpp_xchg<8>(v));
	double a, b;
	rows16(v, a, b);
	v = fmax(a, b);
	rows32(v, a, b);
	return fmax(a, b);
}
__device__ __forceinline__ double wmin(double v)
{
	v = fmin(v, dpp_xchg<1>(v));
	v = fmin(v, dpp_xchg<2>(v));
	v = fmin(v, dpp_xchg<4>(v));
	v = fmin(v, dpp_xchg<8>(v));
	double a, b;
	rows16(v, a, b);
	v = fmin(a, b);
	rows32(v, a, b);
	return fmin(a, b);
}
__device__ __forceinline__ double wsum(double v)
{
	v += dpp_xchg<1>(v);
	v += dpp_xchg<2>(v);
	v += dpp_xchg<4>(v);
	v += dpp_xchg<8>(v);
	double a, b;
	rows16(v, a, b);
	v = a + b;
	rows32(v, a, b);
	return a + b;
}
#endif
__device__ __forceinline__ double lane_get(double v, int src) // src wave-uniform
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
}

template <int VPT, int RPT, bool FULLH>
struct LdsQp {
	// LDS
	double *At, *S, *Pp, *Kp, *va, *dinv, *vr, *ls_s, *ls_d, *ls_l, *ls_u, *ls_m, *ls_t;
	int *alist;
	int lane, nv, nc, RS, NVP, nact;
	// variables owned by this lane
	bool isv[VPT];
	double x[VPT], xh[VPT], q[VPT], Pd[VPT], D[VPT], ab[VPT], Eb[VPT], lbs[VPT], ubs[VPT], yb[VPT], mub[VPT];
	// general rows owned by this lane
	bool isr[RPT];
	double l[RPT], u[RPT], E[RPT], y[RPT], mu[RPT];
	double cs;

	// The workgroup IS one wavefront, and a wavefront's LDS instructions execute in issue order: a write by one lane
	// is visible to a later read by another without a hardware barrier.  What is needed is that the compiler keeps
	// that order -- a wavefront-scope fence -- not s_barrier with its full drain of the LDS queue at every phase.
	__device__ __forceinline__ void sync()
	{
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	__device__ __forceinline__ int vj(int v) const { return lane + 64 * v; }
	__device__ __forceinline__ static int pidx(int a, int b) { return a <= b ? b * (b + 1) / 2 + a : a * (a + 1) / 2 + b; }

	// rows: out[r] = a_i . v   (v: per-variable registers);  ABS: |a_i| . |v|
	template <bool ABS = false>
	__device__ __forceinline__ void row_dot(const double (&v)[VPT], double (&out)[RPT])
	{
#pragma unroll
		for (int k = 0; k < VPT; k++) va[vj(k)] = isv[k] ? (ABS ? fabs(v[k]) : v[k]) : 0.0;
		sync();
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			double s = 0.0;
			const int i = vj(r);
			if (isr[r]) {
#pragma unroll 8
				for (int j = 0; j < nv; j++) s += (ABS ? fabs(At[j * RS + i]) : At[j * RS + i]) * va[j];
			}
			out[r] = s;
		}
		sync();
	}
	// variables: out[v] = sum_i a_ij w_i   (w: per-row registers);  ABS: sum_i |a_ij| |w_i|
	template <bool ABS = false>
	__device__ __forceinline__ void col_dot(const double (&w)[RPT], double (&out)[VPT])
	{
#pragma unroll
		for (int r = 0; r < RPT; r++) vr[vj(r)] = isr[r] ? (ABS ? fabs(w[r]) : w[r]) : 0.0;
		sync();
#pragma unroll
		for (int k = 0; k < VPT; k++) {
			double s = 0.0;
			const int j = vj(k);
			if (isv[k]) {
#pragma unroll 8
				for (int i = 0; i < nc; i++) s += (ABS ? fabs(At[j * RS + i]) : At[j * RS + i]) * vr[i];
			}
			out[k] = s;
		}
		sync();
	}
	// out = P v;  ABS: |P| |v|
	template <bool ABS = false>
	__device__ __forceinline__ void p_mul(const double (&v)[VPT], double (&out)[VPT])
	{
		if constexpr (FULLH) {
#pragma unroll
			for (int k = 0; k < VPT; k++) va[vj(k)] = isv[k] ? (ABS ? fabs(v[k]) : v[k]) : 0.0;
			sync();
#pragma unroll
			for (int k = 0; k < VPT; k++) {
				double s = 0.0;
				const int j = vj(k);
				if (isv[k]) {
#pragma unroll 8
					for (int c = 0; c < nv; c++) s += (ABS ? fabs(Pp[pidx(c, j)]) : Pp[pidx(c, j)]) * va[c];
				}
				out[k] = s;
			}
			sync();
		} else {
#pragma unroll
			for (int k = 0; k < VPT; k++) out[k] = ABS ? fabs(Pd[k] * v[k]) : Pd[k] * v[k];
		}
	}

	// power-of-two Ruiz equilibration of [P A'; A 0] with the identity block of the bounds, as admm_small.hpp
	__device__ __forceinline__ void scale(int iters)
	{
		cs = 1.0;
#pragma unroll
		for (int k = 0; k < VPT; k++) { D[k] = 1.0; ab[k] = 1.0; Eb[k] = 1.0; }
#pragma unroll
		for (int r = 0; r < RPT; r++) E[r] = 1.0;
		for (int it = 0; it < iters; it++) {
			double Dt[VPT], Et[RPT], pcol[VPT];
#pragma unroll
			for (int k = 0; k < VPT; k++) {
				const int j = vj(k);
				double v = 0.0, pc = fabs(Pd[k]);
				if (isv[k]) {
					for (int i = 0; i < nc; i++) v = fmax(v, fabs(At[j * RS + i]));
					if constexpr (FULLH)
						for (int c = 0; c < nv; c++) pc = fmax(pc, fabs(Pp[pidx(c, j)]));
				}
				pcol[k] = pc;
				v = fmax(v, fmax(pc, fabs(ab[k])));
				Dt[k] = isv[k] ? pow2_rsqrt(limit_scaling(v)) : 1.0;
			}
#pragma unroll
			for (int r = 0; r < RPT; r++) {
				double v = 0.0;
				const int i = vj(r);
				if (isr[r])
					for (int j = 0; j < nv; j++) v = fmax(v, fabs(At[j * RS + i]));
				Et[r] = isr[r] ? pow2_rsqrt(limit_scaling(v)) : 1.0;
				E[r] *= Et[r];
			}
#pragma unroll
			for (int k = 0; k < VPT; k++) va[vj(k)] = Dt[k];
			sync();
#pragma unroll
			for (int r = 0; r < RPT; r++) {
				const int i = vj(r);
				if (isr[r])
					for (int j = 0; j < nv; j++) At[j * RS + i] *= Et[r] * va[j];
			}
			if constexpr (FULLH) {
#pragma unroll
				for (int k = 0; k < VPT; k++) {
					const int j = vj(k);
					if (isv[k])
						for (int c = 0; c <= j; c++) Pp[pidx(c, j)] *= Dt[k] * va[c];
				}
			}
			sync();
			double cm = 0.0, qn = 0.0;
#pragma unroll
			for (int k = 0; k < VPT; k++) {
				const double Etb = pow2_rsqrt(limit_scaling(fabs(ab[k])));
				Eb[k] *= Etb;
				ab[k] *= Etb * Dt[k];
				Pd[k] *= Dt[k] * Dt[k];
				q[k] *= Dt[k];
				D[k] *= Dt[k];
				if (isv[k]) {
					double pc = fabs(Pd[k]);
					if constexpr (FULLH) {
						const int j = vj(k);
						for (int c = 0; c < nv; c++) pc = fmax(pc, fabs(Pp[pidx(c, j)]));
					}
					cm += pc;
					qn = fmax(qn, fabs(q[k]));
				}
			}
			(void)pcol;
			cm = wsum(cm) / (double)nv;
			qn = limit_scaling(wmax(qn));
			const double ct = pow2_floor_inv(limit_scaling(fmax(cm, qn)));
#pragma unroll
			for (int k = 0; k < VPT; k++) {
				Pd[k] *= ct;
				q[k] *= ct;
			}
			if constexpr (FULLH) {
#pragma unroll
				for (int k = 0; k < VPT; k++) {
					const int j = vj(k);
					if (isv[k])
						for (int c = 0; c <= j; c++) Pp[pidx(c, j)] *= ct;
				}
				sync();
			}
			cs *= ct;
		}
	}

	// S <- P + I/gamma + sum_{i active} mu_i a_i a_i' + diag(active bounds), then LDL' in place.
	// actr / actb: this lane's rows / bounds currently outside their interval.
	//
	// DELTA (shapes whose LDS has room for it: Kp, the unfactored K_J, upper triangle packed by columns -- lane j's column
	// is contiguous): K_J is kept from one Newton step to the next and only the rows and bounds that ENTERED or LEFT the
	// active set since (inr / inb: what Kp holds) are added or taken out, each a rank-one term; S is Kp's copy, factored
	// in place as before.  The full sum is O(|J| nv^2) per Newton step -- half the kernel's time at 86 x 65, where most
	// of the 65 rows are equalities that never leave -- the update O(changes nv^2).  A term taken out leaves the rounding
	// of its two additions behind (eps mu |a|^2, against a diagonal that may be 1/gamma): the caller rebuilds in full
	// whenever the penalties change and after kLdsMaxDeltas updates in a row; the gradient and the line search never
	// see K_J, so its accuracy can cost Newton steps, not the answer.
	template <bool DELTA>
	__device__ __forceinline__ void build(const bool (&actr)[RPT], const bool (&actb)[VPT], bool (&inr)[RPT], bool (&inb)[VPT])
	{
		// compact list of the general rows to add (DELTA: or to take out) + their weights
		int base = 0;
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			const bool act = isr[r] && actr[r];
			const bool a = DELTA ? (act != inr[r]) : act;
			const unsigned long long m = __ballot(a);
			const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
			if (a) alist[pos] = vj(r);
			base += __popcll(m);
			vr[vj(r)] = (DELTA && !act) ? -mu[r] : mu[r];
			inr[r] = act;
		}
		nact = base;
		sync();
		// upper triangle, row-major: S[c][j], j >= c  (lane owns column index j), eight entries of the column at a
		// time in registers: per active row one dependent read pair (its index, this lane's coefficient) feeds eight
		// independent broadcast reads, instead of an exposed LDS latency per term
#pragma unroll
		for (int k = 0; k < VPT; k++) {
			const int j = vj(k);
			if (isv[k]) {
				const double bnew = actb[k] ? mub[k] * ab[k] * ab[k] : 0.0, bold = inb[k] ? mub[k] * ab[k] * ab[k] : 0.0;
				auto finish = [&](int c, double sv) { // entry (c, j) of K_J, c <= j
					if constexpr (DELTA) {
						double *kp = Kp + pidx(c, j);
						if (c == j) sv += bnew - bold;
						sv += *kp;
						*kp = sv;
					} else {
						if constexpr (FULLH) sv += Pp[pidx(c, j)];
						if (c == j) {
							if constexpr (!FULLH) sv += Pd[k];
							sv += 1.0 / kLdsGamma + bnew;
						}
						if (Kp) Kp[pidx(c, j)] = sv;
					}
					S[c * NVP + j] = sv;
				};
				// uniform control flow (every lane walks all rows c < nv; the store is what c <= j masks): full blocks of
				// eight rows without index clamps, then the last nv mod 8 rows one at a time
				int c0 = 0;
				for (; c0 + 8 <= nv; c0 += 8) {
					double acc[8];
#pragma unroll
					for (int u = 0; u < 8; u++) acc[u] = 0.0;
					for (int t = 0; t < nact; t++) {
						const int i = alist[t];
						const double w = vr[i] * At[j * RS + i];
#pragma unroll
						for (int u = 0; u < 8; u++) acc[u] += w * At[(c0 + u) * RS + i];
					}
#pragma unroll
					for (int u = 0; u < 8; u++)
						if (c0 + u <= j) finish(c0 + u, acc[u]);
				}
				for (; c0 < nv; c0++) {
					double acc = 0.0;
#pragma unroll 4
					for (int t = 0; t < nact; t++) {
						const int i = alist[t];
						acc += vr[i] * At[j * RS + i] * At[c0 * RS + i];
					}
					if (c0 <= j) finish(c0, acc);
				}
			}
			inb[k] = actb[k];
		}
		sync();
	}
	// LDL' of S in place (right-looking); false if a pivot is not positive
	__device__ __forceinline__ bool factor()
	{
		bool ok = true;
		for (int k = 0; k < nv; k++) {
			const double dk = S[k * NVP + k];
			ok = ok && (dk > 0.0);
			// reciprocal of the pivot by the hardware seed and two Newton steps (full precision for a positive, normal
			// pivot; the IEEE division's rescaling and fix-up steps are for operands a factorisation does not survive)
			double inv = __builtin_amdgcn_rcp(dk);
			inv = fma(fma(-dk, inv, 1.0), inv, inv);
			inv = fma(fma(-dk, inv, 1.0), inv, inv);
			double ljk[VPT];
#pragma unroll
			for (int v = 0; v < VPT; v++) {
				const int j = vj(v);
				ljk[v] = (isv[v] && j > k) ? S[k * NVP + j] * inv : 0.0;
			}
			// rows c > k of the trailing block, both triangles (the lower one is scratch until its column is final):
			// one exec mask for the whole loop instead of a compare per entry, reads eight rows deep
#pragma unroll
			for (int v = 0; v < VPT; v++) {
				const int j = vj(v);
				if (isv[v] && j > k) {
					int c0 = k + 1;
					for (; c0 + 8 <= nv; c0 += 8) { // full blocks: all sixteen reads before the first write
						double sc[8], sv[8];
#pragma unroll
						for (int u = 0; u < 8; u++) {
							sc[u] = S[k * NVP + c0 + u];
							sv[u] = S[(c0 + u) * NVP + j];
						}
#pragma unroll
						for (int u = 0; u < 8; u++) S[(c0 + u) * NVP + j] = sv[u] - ljk[v] * sc[u];
					}
					for (; c0 < nv; c0++) S[c0 * NVP + j] -= ljk[v] * S[k * NVP + c0];
				}
			}
			sync();
#pragma unroll
			for (int v = 0; v < VPT; v++) {
				const int j = vj(v);
				if (isv[v] && j > k) {
					S[k * NVP + j] = ljk[v];
					S[j * NVP + k] = ljk[v];
				}
				if (j == k) dinv[j] = inv;
			}
			sync();
		}
		return ok;
	}

	// r <- K^-1 r  (registers; pivots broadcast by v_readlane, factor rows from LDS)
	__device__ __forceinline__ void solve(double (&r)[VPT])
	{
		// four pivots at a time: their factor rows are read before the first dependent step (the rows do not depend on
		// the substitution, only the broadcast right-hand sides do)
		for (int k0 = 0; k0 < nv - 1; k0 += 4) {
			double f[4][VPT];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int k = k0 + u < nv - 1 ? k0 + u : nv - 2;
#pragma unroll
				for (int v = 0; v < VPT; v++) f[u][v] = (isv[v] && vj(v) > k) ? S[k * NVP + vj(v)] : 0.0;
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int k = k0 + u;
				if (k < nv - 1) {
					double rk = lane_get(r[0], k & 63);
					if (VPT > 1 && k >= 64) rk = lane_get(r[VPT - 1], k & 63);
#pragma unroll
					for (int v = 0; v < VPT; v++) r[v] -= f[u][v] * rk;
				}
			}
		}
#pragma unroll
		for (int v = 0; v < VPT; v++) r[v] = isv[v] ? r[v] * dinv[vj(v)] : 0.0;
		for (int k0 = nv - 1; k0 >= 1; k0 -= 4) {
			double f[4][VPT];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int k = k0 - u >= 1 ? k0 - u : 1;
#pragma unroll
				for (int v = 0; v < VPT; v++) f[u][v] = vj(v) < k ? S[k * NVP + vj(v)] : 0.0;
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int k = k0 - u;
				if (k >= 1) {
					double rk = lane_get(r[0], k & 63);
					if (VPT > 1 && k >= 64) rk = lane_get(r[VPT - 1], k & 63);
#pragma unroll
					for (int v = 0; v < VPT; v++) r[v] -= f[u][v] * rk;
				}
			}
		}
	}
};

// status / iters follow asif_hip_qp_solve_batch's contract (QPWrapperOsqp::solve, src/qpwrapper_osqp.cpp:225-238)
// WARM: the instantiation behind asif_hip_qp_solve_batch_warm (qp_inv.hpp: same contract, same units)
template <int VPT, int RPT, bool FULLH, bool WARM = false>
__global__ __launch_bounds__(64) void qp_lds_kernel(asif_hip_solver S_, QpArgs a)
{
	extern __shared__ double lds[];
	using W = LdsQp<VPT, RPT, FULLH>;
	W s;
	const int lane = threadIdx.x;
	const int64_t qi = xcd_contiguous_index(blockIdx.x, a.B);
	if (qi >= a.B) return; // wave-uniform (one wave per workgroup)
	if (a.only_status != 0 && a.status[qi] != a.only_status) return; // second pass: this instance is already decided
	const int nv = a.nv, nc = a.nc;
	const int64_t ld = a.ld;
	s.lane = lane;
	s.nv = nv;
	s.nc = nc;
	s.RS = lds_rs(nc);
	s.NVP = lds_nvp(nv);
	{
		double *p = lds;
		s.At = p; p += (size_t)nv * s.RS;
		s.S = p; p += (size_t)nv * s.NVP;
		s.Pp = p; if (FULLH) p += (size_t)nv * (nv + 1) / 2;
		s.Kp = a.keep_kj ? p : nullptr; if (a.keep_kj) p += (size_t)nv * (nv + 1) / 2;
		s.va = p; p += 64 * VPT;
		s.dinv = p; p += 64 * VPT;
		s.vr = p; p += 64 * RPT;
		const int m = 64 * (RPT + VPT);
		s.ls_s = p; p += m;
		s.ls_d = p; p += m;
		s.ls_l = p; p += m;
		s.ls_u = p; p += m;
		s.ls_m = p; p += m;
		s.ls_t = p; p += 2 + 2 * m;
		s.alist = (int *)p;
	}
	// ---- form translation (src/qpwrapper_osqp.cpp:263-376): P = 2H, q = c, rows [A; I], l = [b; lb], u = [inf | b; ub]
#pragma unroll
	for (int r = 0; r < RPT; r++) {
		const int i = s.vj(r);
		s.isr[r] = i < nc;
		double lo = -kInfty, hi = kInfty;
		if (s.isr[r]) {
			for (int j = 0; j < nv; j++) s.At[j * s.RS + i] = a.A[(int64_t)(i + j * nc) * ld + qi];
			lo = a.b[(int64_t)i * ld + qi];
			const bool eq = (i < 64 ? (a.be_mask >> i) : (a.be_mask2 >> (i - 64))) & 1ull;
			hi = eq ? lo : kInfty;
		}
		s.l[r] = lo;
		s.u[r] = hi;
	}
#pragma unroll
	for (int k = 0; k < VPT; k++) {
		const int j = s.vj(k);
		s.isv[k] = j < nv;
		s.Pd[k] = 0.0;
		s.q[k] = 0.0;
		s.lbs[k] = -kInfty;
		s.ubs[k] = kInfty;
		if (s.isv[k]) {
			s.q[k] = a.c[(int64_t)j * ld + qi];
			s.lbs[k] = a.lb[(int64_t)j * ld + qi];
			s.ubs[k] = a.ub[(int64_t)j * ld + qi];
			if constexpr (FULLH) {
				// the wrapper hands OSQP the upper triangle of 2H (src/qpwrapper_osqp.cpp:136-153)
				for (int c = 0; c <= j; c++) s.Pp[W::pidx(c, j)] = 2.0 * a.H[(int64_t)(c + j * nv) * ld + qi];
			} else {
				s.Pd[k] = 2.0 * a.Hd[(int64_t)j * ld + qi];
			}
		}
	}
	s.sync();
	s.scale(S_.scaling_iters);
#pragma unroll
	for (int r = 0; r < RPT; r++) {
		s.l[r] *= s.E[r];
		s.u[r] *= s.E[r];
		s.y[r] = 0.0;
		s.mu[r] = (s.u[r] - s.l[r] < kRhoTol) ? 100.0 * kLdsMu0 : kLdsMu0;
	}
#pragma unroll
	for (int k = 0; k < VPT; k++) {
		s.lbs[k] *= s.Eb[k];
		s.ubs[k] *= s.Eb[k];
		s.x[k] = 0.0;
		s.xh[k] = 0.0;
		s.yb[k] = 0.0;
		s.mub[k] = (s.ubs[k] - s.lbs[k] < kRhoTol) ? 100.0 * kLdsMu0 : kLdsMu0;
	}
	if constexpr (WARM) {
		if (a.warm_in != 0) {
			auto sane = [](double v) { return fabs(v) < 1e100 ? v : 0.0; }; // NaN, inf, nonsense: no start
#pragma unroll
			for (int k = 0; k < VPT; k++)
				if (s.isv[k]) {
					s.x[k] = sane(a.warm_x[(int64_t)s.vj(k) * ld + qi]) / s.D[k];
					s.xh[k] = s.x[k];
					s.yb[k] = sane(a.warm_y[(int64_t)(nc + s.vj(k)) * ld + qi]) * s.cs / s.Eb[k];
				}
#pragma unroll
			for (int r = 0; r < RPT; r++)
				if (s.isr[r]) s.y[r] = sane(a.warm_y[(int64_t)s.vj(r) * ld + qi]) * s.cs / s.E[r];
		}
	}
	const double tol = fmax(S_.eps_rel, 1e-10) * 1e-2; // default eps 1e-8 -> 1e-10 on the scaled residuals
	const double big = kInfty * kMinScaling;
	const int max_newton = S_.max_iter > 0 ? S_.max_iter : 4000;
	int status = 0, newton = 0;
	bool have_factor = false, fact_ok = true, kj_valid = false;
	int deltas = 0;
	bool pactr[RPT], pactb[VPT];
#pragma unroll
	for (int r = 0; r < RPT; r++) pactr[r] = false;
#pragma unroll
	for (int k = 0; k < VPT; k++) pactb[k] = false;
	double pri_prev = -1.0, best_res = 1e300;
	[[maybe_unused]] int met = 0; // WARM: consecutive multiplier updates that met the termination test
	// section timers of a scratch build (tools/dev_lds_sections.py); compiled out of the library
#ifdef ASIF_LDS_PROFILE
	long long tsec[7] = {0, 0, 0, 0, 0, 0, 0}, tmark = __builtin_readcyclecounter();
#define LDS_T(k) { const long long tn_ = __builtin_readcyclecounter(); tsec[k] += tn_ - tmark; tmark = tn_; }
#else
#define LDS_T(k)
#endif

	for (int outer = 0; outer < kLdsMaxOuter && status == 0; outer++) {
		double sr[RPT], sb[VPT]; // s = a.x + y/mu of the rows / bounds at the current x
		double gfloor = 0.0, gscale = 1.0;
		for (int inner = 0; inner < kLdsMaxInner; inner++) {
			LDS_T(5)
			// ---- gradient of the inner objective
			double ax[RPT], rr[RPT], px[VPT], atr[VPT], g[VPT];
			s.row_dot(s.x, ax);
			bool actr[RPT], actb[VPT];
#pragma unroll
			for (int r = 0; r < RPT; r++) {
				sr[r] = ax[r] + s.y[r] / s.mu[r];
				const double pj = fmin(fmax(sr[r], s.l[r]), s.u[r]);
				rr[r] = s.isr[r] ? s.mu[r] * (sr[r] - pj) : 0.0;
				actr[r] = s.isr[r] && (sr[r] < s.l[r] || sr[r] > s.u[r]);
			}
			s.col_dot(rr, atr);
			s.p_mul(s.x, px);
			// scale of the gradient's own terms (a sum of large cancelling terms must not look like a small gradient):
			// an order of magnitude, taken at the first step of each inner solve and kept for the rest of it
			double atra[VPT], pxa[VPT];
			if (inner == 0) {
				s.template col_dot<true>(rr, atra);
				s.template p_mul<true>(s.x, pxa);
			}
			double gn = 0.0, gs = 0.0;
			bool changed = !have_factor;
			double rbv[VPT];
#pragma unroll
			for (int k = 0; k < VPT; k++) {
				sb[k] = s.ab[k] * s.x[k] + s.yb[k] / s.mub[k];
				const double pj = fmin(fmax(sb[k], s.lbs[k]), s.ubs[k]);
				const double rb = s.isv[k] ? s.mub[k] * (sb[k] - pj) : 0.0;
				rbv[k] = rb;
				actb[k] = s.isv[k] && (sb[k] < s.lbs[k] || sb[k] > s.ubs[k]);
				const double ay = atr[k] + s.ab[k] * rb;
				g[k] = s.isv[k] ? px[k] + s.q[k] + (s.x[k] - s.xh[k]) * (1.0 / kLdsGamma) + ay : 0.0;
				gn = fmax(gn, fabs(g[k]));
				if (inner == 0) gs = fmax(gs, fmax(pxa[k], fmax(fabs(s.q[k]), atra[k] + fabs(s.ab[k] * rb))));
				changed = changed || (actb[k] != pactb[k]);
			}
#pragma unroll
			for (int r = 0; r < RPT; r++) changed = changed || (actr[r] != pactr[r]);
			gn = wmax(gn);
			if (inner == 0) gscale = 1.0 + wmax(gs);
			gs = gscale;
			if (inner == 0) {
				// rounding floor of the gradient: r_i = mu_i (s_i - proj s_i) carries mu_i eps (terms of s_i)
				double axa[RPT], er[RPT], fl[VPT];
				s.template row_dot<true>(s.x, axa);
#pragma unroll
				for (int r = 0; r < RPT; r++) {
					const double bd = sr[r] < s.l[r] ? fabs(s.l[r]) : (sr[r] > s.u[r] ? fabs(s.u[r]) : 0.0);
					er[r] = s.isr[r] ? 2.2e-16 * s.mu[r] * (axa[r] + fabs(s.y[r]) / s.mu[r] + bd) : 0.0;
				}
				s.template col_dot<true>(er, fl);
				double f = 0.0;
#pragma unroll
				for (int k = 0; k < VPT; k++) {
					const double bd = sb[k] < s.lbs[k] ? fabs(s.lbs[k]) : (sb[k] > s.ubs[k] ? fabs(s.ubs[k]) : 0.0);
					const double eb = 2.2e-16 * s.mub[k] * (fabs(s.ab[k] * s.x[k]) + fabs(s.yb[k]) / s.mub[k] + bd);
					if (s.isv[k]) f = fmax(f, fl[k] + fabs(s.ab[k]) * eb);
				}
				gfloor = wmax(f) + 2.2e-16 * gs;
			}
			(void)rbv;
			LDS_T(0)
			if (gn <= 0.1 * tol * gs || gn <= 8.0 * gfloor) break;
			if (newton >= max_newton) break;
			// ---- Newton direction on the current active set
			if (__any(changed)) {
				// (pactr / pactb double as "what Kp holds": build() brings them up to date)
				if (kj_valid && deltas < kLdsMaxDeltas) {
					s.template build<true>(actr, actb, pactr, pactb);
					deltas++;
				} else {
					s.template build<false>(actr, actb, pactr, pactb);
					kj_valid = s.Kp != nullptr;
					deltas = 0;
				}
				LDS_T(6)
				fact_ok = s.factor() && fact_ok;
				have_factor = true;
			}
			LDS_T(1)
			double d[VPT];
#pragma unroll
			for (int k = 0; k < VPT; k++) d[k] = -g[k];
			s.solve(d);
			LDS_T(2)
			newton++;
			// ---- exact line search: phi'(t) = qa + t a1 + sum_i mu_i dl_i (s_i + t dl_i - proj(s_i + t dl_i))
			double dl[RPT], pd[VPT];
			s.row_dot(d, dl);
			s.p_mul(d, pd);
			double qa = 0.0, a1 = 0.0;
#pragma unroll
			for (int k = 0; k < VPT; k++) {
				if (s.isv[k]) {
					qa += (px[k] + s.q[k] + (s.x[k] - s.xh[k]) * (1.0 / kLdsGamma)) * d[k];
					a1 += pd[k] * d[k] + d[k] * d[k] * (1.0 / kLdsGamma);
				}
			}
			qa = wsum(qa);
			a1 = wsum(a1);
			// all rows (general, then bounds) side by side in LDS; this lane's breakpoints in (0, 1]
			double bp[2 * (RPT + VPT)];
			int nb = 0;
#pragma unroll
			for (int r = 0; r < RPT; r++) {
				const int i = s.vj(r);
				s.ls_s[i] = sr[r];
				s.ls_d[i] = s.isr[r] ? dl[r] : 0.0;
				s.ls_l[i] = s.l[r];
				s.ls_u[i] = s.u[r];
				s.ls_m[i] = s.isr[r] ? s.mu[r] : 0.0;
				const double t1 = (s.l[r] - sr[r]) / dl[r], t2 = (s.u[r] - sr[r]) / dl[r];
				bp[nb++] = (s.isr[r] && t1 > 0.0 && t1 <= 1.0) ? t1 : 2.0;
				bp[nb++] = (s.isr[r] && t2 > 0.0 && t2 <= 1.0) ? t2 : 2.0;
			}
#pragma unroll
			for (int k = 0; k < VPT; k++) {
				const int i = 64 * RPT + s.vj(k);
				const double dlb = s.ab[k] * d[k];
				s.ls_s[i] = sb[k];
				s.ls_d[i] = s.isv[k] ? dlb : 0.0;
				s.ls_l[i] = s.lbs[k];
				s.ls_u[i] = s.ubs[k];
				s.ls_m[i] = s.isv[k] ? s.mub[k] : 0.0;
				const double t1 = (s.lbs[k] - sb[k]) / dlb, t2 = (s.ubs[k] - sb[k]) / dlb;
				bp[nb++] = (s.isv[k] && t1 > 0.0 && t1 <= 1.0) ? t1 : 2.0;
				bp[nb++] = (s.isv[k] && t2 > 0.0 && t2 <= 1.0) ? t2 : 2.0;
			}
			s.sync();
			auto dphi = [&](double t) {
				double f = qa + t * a1;
#pragma unroll 4
				for (int i = 0; i < nc; i++) { // general rows
					const double st = s.ls_s[i] + t * s.ls_d[i];
					f += s.ls_m[i] * s.ls_d[i] * (st - fmin(fmax(st, s.ls_l[i]), s.ls_u[i]));
				}
#pragma unroll 4
				for (int i = 64 * RPT; i < 64 * RPT + nv; i++) { // bounds
					const double st = s.ls_s[i] + t * s.ls_d[i];
					f += s.ls_m[i] * s.ls_d[i] * (st - fmin(fmax(st, s.ls_l[i]), s.ls_u[i]));
				}
				return f;
			};
			// bracket of the zero of phi' among {0} u breakpoints u {1}.  The evaluation points go into one list (0, 1,
			// then the breakpoints in (0, 1], compacted by ballots) and every lane evaluates phi' at its own point:
			// one pass over the rows for up to 64 points instead of one pass per point.
			double tlo = 0.0, flo = 0.0, thi = 2.0, fhi = 0.0;
			{
				int npt = 2;
				if (lane == 0) s.ls_t[0] = 0.0;
				if (lane == 1) s.ls_t[1] = 1.0;
#pragma unroll
				for (int e = 0; e < 2 * (RPT + VPT); e++) {
					const bool v = bp[e] <= 1.0;
					const unsigned long long m = __ballot(v);
					if (v) s.ls_t[npt + __popcll(m & ((1ull << lane) - 1ull))] = bp[e];
					npt += __popcll(m);
				}
				s.sync();
				for (int base = 0; base < npt; base += 64) {
					const bool mine = base + lane < npt;
					const double tb = mine ? s.ls_t[base + lane] : 1.0;
					const double fb = dphi(tb);
					if (base == 0) {
						const double f0 = lane_get(fb, 0), f1 = lane_get(fb, 1);
						flo = f0;
						if (f1 >= 0.0) { thi = 1.0; fhi = f1; }
					}
					const bool isbp = mine && (base + lane >= 2);
					// best lower bracket: largest t with f < 0; best upper: smallest t with f >= 0
					const double cl = (isbp && fb < 0.0) ? tb : -1.0;
					const double ch = (isbp && fb >= 0.0) ? tb : 3.0;
					const double gl = wmax(cl), gh = wmin(ch);
					if (gl > tlo) {
						tlo = gl;
						flo = wmax(cl == gl ? fb : -1e300); // f at that breakpoint (negative: max picks it among ties)
					}
					if (gh < thi) {
						thi = gh;
						fhi = wmin(ch == gh ? fb : 1e300);
					}
				}
				s.sync();
			}
			double t = 1.0;
			if (thi <= 1.0) t = (fhi > flo) ? tlo - flo * (thi - tlo) / (fhi - flo) : tlo;
			if (!(t > 0.0)) t = thi <= 1.0 ? thi : 1.0; // degenerate bracket: take the upper end
#pragma unroll
			for (int k = 0; k < VPT; k++) s.x[k] += t * d[k];
			LDS_T(3)
		}
		LDS_T(5)
		// ---- multiplier update, residuals, certificates
		double ax[RPT], ynew[RPT], aty[VPT], px[VPT];
		s.row_dot(s.x, ax);
		double pri = 0.0, nax = 0.0, ndy = 0.0, lhs = 0.0, vcert[RPT];
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			const double sv = ax[r] + s.y[r] / s.mu[r];
			ynew[r] = s.isr[r] ? s.mu[r] * (sv - fmin(fmax(sv, s.l[r]), s.u[r])) : 0.0;
			const double viol = ax[r] - fmin(fmax(ax[r], s.l[r]), s.u[r]);
			if (s.isr[r]) {
				pri = fmax(pri, fabs(viol));
				nax = fmax(nax, fabs(ax[r]));
			}
			double v = ynew[r] - s.y[r];
			if (s.u[r] > big) v = (s.l[r] < -big) ? 0.0 : fmin(v, 0.0);
			else if (s.l[r] < -big) v = fmax(v, 0.0);
			vcert[r] = s.isr[r] ? v : 0.0;
			ndy = fmax(ndy, fabs(vcert[r]));
			lhs += vcert[r] > 0.0 ? s.u[r] * vcert[r] : (vcert[r] < 0.0 ? s.l[r] * vcert[r] : 0.0);
			s.y[r] = ynew[r];
		}
		s.col_dot(ynew, aty);
		double atv[VPT], atya[VPT], pxa[VPT];
		s.col_dot(vcert, atv);
		s.p_mul(s.x, px);
		s.template col_dot<true>(ynew, atya);
		s.template p_mul<true>(s.x, pxa);
		double dua = 0.0, nd = 0.0, natv = 0.0;
#pragma unroll
		for (int k = 0; k < VPT; k++) {
			const double axb = s.ab[k] * s.x[k];
			const double sv = axb + s.yb[k] / s.mub[k];
			const double ybn = s.isv[k] ? s.mub[k] * (sv - fmin(fmax(sv, s.lbs[k]), s.ubs[k])) : 0.0;
			const double viol = axb - fmin(fmax(axb, s.lbs[k]), s.ubs[k]);
			double v = ybn - s.yb[k];
			if (s.ubs[k] > big) v = (s.lbs[k] < -big) ? 0.0 : fmin(v, 0.0);
			else if (s.lbs[k] < -big) v = fmax(v, 0.0);
			if (!s.isv[k]) v = 0.0;
			s.yb[k] = ybn;
			s.xh[k] = s.x[k];
			if (s.isv[k]) {
				pri = fmax(pri, fabs(viol));
				nax = fmax(nax, fabs(axb));
				const double ay = aty[k] + s.ab[k] * ybn;
				dua = fmax(dua, fabs(px[k] + s.q[k] + ay));
				nd = fmax(nd, fmax(pxa[k], fmax(fabs(s.q[k]), atya[k] + fabs(s.ab[k] * ybn))));
				ndy = fmax(ndy, fabs(v));
				lhs += v > 0.0 ? s.ubs[k] * v : (v < 0.0 ? s.lbs[k] * v : 0.0);
				natv = fmax(natv, fabs(atv[k] + s.ab[k] * v));
			}
		}
		pri = wmax(pri); nax = wmax(nax); dua = wmax(dua); nd = wmax(nd);
		ndy = wmax(ndy); lhs = wsum(lhs); natv = wmax(natv);
		const double rp = pri / (1.0 + nax), rd = dua / (1.0 + nd);
		best_res = fmin(best_res, fmax(rp, rd));
		bool done = rp <= tol && rd <= tol;
		if constexpr (WARM) {
			// a warm solve is done when three multiplier updates in a row meet the test (qp_inv.hpp: why)
			met = done ? met + 1 : 0;
			done = done && met >= (a.warm_in != 0 ? 3 : 1);
		}
		if (done) status = kStatusSolved;
		else if (ndy > 1e-4 && lhs < -1e-6 * ndy && natv < 1e-6 * ndy) status = kStatusPrimalInf;
		else if (newton >= max_newton || !fact_ok) status = kStatusMaxIter;
		else {
			double mumin = 1e300, mumax = 0.0;
#pragma unroll
			for (int r = 0; r < RPT; r++)
				if (s.isr[r]) { mumin = fmin(mumin, s.mu[r]); mumax = fmax(mumax, s.mu[r]); }
#pragma unroll
			for (int k = 0; k < VPT; k++)
				if (s.isv[k]) { mumin = fmin(mumin, s.mub[k]); mumax = fmax(mumax, s.mub[k]); }
			mumin = wmin(mumin);
			mumax = wmax(mumax);
			double f = 1.0, cap = kLdsMuMax;
			if (rp <= tol) {
				// rows are met, the dual residual is not: a softer penalty lowers the rounding floor of mu (s - proj s)
				if (mumax > 100.0 * kLdsMu0) f = 0.1;
			} else if (pri_prev >= 0.0 && pri > 0.5 * pri_prev && mumin >= kLdsMuMax) {
				f = 10.0; // stalled at the cap: the multipliers have far to go (rows with tiny coefficients)
				cap = 1e8;
			} else if (pri_prev >= 0.0 && pri > 0.1 * pri_prev) {
				f = penalty_jump(pri, pri_prev); // not fast enough -> stiffer penalties, by what the contraction seen asks for
			}
			if (f != 1.0) {
#pragma unroll
				for (int r = 0; r < RPT; r++) s.mu[r] = f > 1.0 ? fmin(s.mu[r] * f, fmax(s.mu[r], cap)) : fmax(s.mu[r] * f, kLdsMu0);
#pragma unroll
				for (int k = 0; k < VPT; k++) s.mub[k] = f > 1.0 ? fmin(s.mub[k] * f, fmax(s.mub[k], cap)) : fmax(s.mub[k] * f, kLdsMu0);
				have_factor = false;
				kj_valid = false; // the rows' weights in Kp are the old penalties
			}
		}
		pri_prev = pri;
		LDS_T(4)
	}
	if (status == 0 || status == kStatusMaxIter) {
		// budget spent: OSQP's "solved inaccurate" counts as solved for the wrapper (src/qpwrapper_osqp.cpp:225)
		status = best_res <= 1e3 * tol ? kStatusSolved : kStatusMaxIter;
	}
#pragma unroll
	for (int k = 0; k < VPT; k++)
		if (s.isv[k]) a.sol[(int64_t)s.vj(k) * ld + qi] = s.D[k] * s.x[k];
	if (lane == 0) {
		a.status[qi] = status;
		if (a.iters) a.iters[qi] = newton;
	}
	if constexpr (WARM) {
		const bool ok = status == kStatusSolved; // a problem without a solution leaves a cold start behind
#pragma unroll
		for (int k = 0; k < VPT; k++)
			if (s.isv[k]) {
				a.warm_x[(int64_t)s.vj(k) * ld + qi] = ok ? s.D[k] * s.x[k] : 0.0;
				a.warm_y[(int64_t)(nc + s.vj(k)) * ld + qi] = ok ? s.Eb[k] * s.yb[k] / s.cs : 0.0;
			}
#pragma unroll
		for (int r = 0; r < RPT; r++)
			if (s.isr[r]) a.warm_y[(int64_t)s.vj(r) * ld + qi] = ok ? s.E[r] * s.y[r] / s.cs : 0.0;
	}
#ifdef ASIF_LDS_PROFILE
	if (lane == 0)
		for (int k = 0; k < 7; k++) a.sol[(int64_t)k * ld + qi] = (double)tsec[k]; // scratch build: times instead of x
#endif
#undef LDS_T
}

} // namespace asif
