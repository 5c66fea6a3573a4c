// qp_inv.hpp -- TWO QPs per wavefront (one per half-wave of 32 lanes) for the small lifted problems the reference's
// classes hand to QPWrapperOsqp: ASIFrobust 18 x 12 (src/asif_robust.cpp:21-22, 89-148), 22 x 15 on the shipped
// half-planes, and every pre-assembled shape with nv <= 32, nc <= 32 and a diagonal cost.  Same method as qp_lds.hpp
// (proximal method of multipliers, inner problems solved exactly by a semismooth Newton iteration with an exact line
// search -- see that file for the derivation, the termination tests and the penalty schedule, which are kept to the
// letter); what changes is the layout and the linear algebra:
//
//  * qp_lds.hpp gives a QP the whole wave: with 18 variables and 12 rows, 18 of 64 lanes own anything, and every
//    change of the active set J rebuilds K_J = P + I/gamma + sum_{i in J} mu_i a_i a_i' and factors it again
//    (57 % of its time).  Here lane t of a half-wave owns variable t and row t of ITS QP; the wave's instruction
//    stream serves two problems.
//  * K_J is never factored.  Lane t keeps row t of K_J^-1 in registers.  From the diagonal matrix P + I/gamma (its
//    inverse is known) every row that enters J is one Sherman-Morrison step -- adding a positive semidefinite term,
//    the denominator 1 + mu a'K^-1 a is >= 1: no pivoting question, nv fused multiply-adds per lane for K^-1 a and nv
//    for the update.  A row LEAVING J would be a downdate (denominator 1 - ..., ill-conditioned where the cost has no
//    curvature): not done -- the inverse is rebuilt from the diagonal by |J| additions.  The Newton direction is
//    d = -K^-1 g (one product, no triangular solves with their nv dependent broadcasts), followed by ONE step of
//    iterative refinement through the exact operator, d += K^-1 (-g - K_J d): the running inverse may carry 1e-8 of
//    cancellation (1/gamma = 1e-7 against mu a a'), the refined direction does not; the line search is exact and the
//    gradient is computed from the data, so the inverse's accuracy sets the iteration count, never the answer.
//    (numpy prototype against the exact oracle, 600 instances each of 18 x 12 and 22 x 15: same verdicts and errors as
//    the factorisation, 5.1 instead of 4.5 Newton steps, 1.3 rebuilds + 20 rank-one steps per QP.)
//  * the line search first asks for phi'(1) alone (every lane its own row, one reduction): the derivative is
//    monotone, so phi'(1) < 0 means the full step, as the breakpoint search would have found; that search (every lane
//    a breakpoint, a pass over all rows each) runs only for a half whose step has to be cut.
//  * reductions run over 32 lanes: four DPP steps inside the rows of 16 and one ds_swizzle across them.
//
//  * every vector instruction costs its wave four cycles or more and the FP64 ones 5.5-6.25 (tools/scratch/
//    lanes_issue.hip), so the instruction count IS the cost: shapes are padded to compile-time sizes <NVMAX, NCMAX>
//    (zero rows / columns) so that every loop is unrolled and every LDS address is a base register plus an immediate;
//    1/mu is kept instead of dividing by mu; the other divisions whose operands are well scaled are a hardware
//    reciprocal seed and two Newton steps.
//
// LDS per half: the scaled row block transposed At[NVMAX][NCMAX + 1], three broadcast vectors, the line search's view
// of all rows.  <20, 16> (18 x 12): 7.3 KB per QP.
#pragma once
#include "qp_lds.hpp"

namespace asif {

// LDS doubles of one problem: hw = lanes that own it (32: two problems per wave; 64: one, for 32 < nv, nc <= 64)
__host__ __device__ inline size_t inv_half_doubles(int nvmax, int ncmax, int hw = 32)
{
	return (size_t)nvmax * (ncmax + 1) + 3 * hw + 5 * 2 * hw + (2 + 4 * hw);
}
// 1 / d by the hardware seed and two Newton steps: full precision for a normal operand away from the ends of the range
// (0, denormals and infinities come back as NaN / inf -- every caller's comparisons treat that as "no value")
__device__ __forceinline__ double fast_rcp(double d)
{
	double r = __builtin_amdgcn_rcp(d);
	r = fma(fma(-d, r, 1.0), r, r);
	r = fma(fma(-d, r, 1.0), r, r);
	return r;
}

// reductions over the HW lanes that own a problem: four DPP steps inside the rows of 16, one row swap across the two
// rows of a 32-lane group, and for HW = 64 one half swap across the two groups
template <int HW>
__device__ __forceinline__ double hsum(double v)
{
	v += dpp_xchg<1>(v);
	v += dpp_xchg<2>(v);
	v += dpp_xchg<4>(v);
	v += dpp_xchg<8>(v);
	double a, b;
	rows16(v, a, b);
	v = a + b;
	if constexpr (HW == 64) {
		rows32(v, a, b);
		v = a + b;
	}
	return v;
}
template <int HW>
__device__ __forceinline__ double hmax(double v)
{
	v = fmax(v, dpp_xchg<1>(v));
	v = fmax(v, dpp_xchg<2>(v));
	v = fmax(v, dpp_xchg<4>(v));
	v = fmax(v, dpp_xchg<8>(v));
	double a, b;
	rows16(v, a, b);
	v = fmax(a, b);
	if constexpr (HW == 64) {
		rows32(v, a, b);
		v = fmax(a, b);
	}
	return v;
}
template <int HW>
__device__ __forceinline__ double hmin(double v)
{
	v = fmin(v, dpp_xchg<1>(v));
	v = fmin(v, dpp_xchg<2>(v));
	v = fmin(v, dpp_xchg<4>(v));
	v = fmin(v, dpp_xchg<8>(v));
	double a, b;
	rows16(v, a, b);
	v = fmin(a, b);
	if constexpr (HW == 64) {
		rows32(v, a, b);
		v = fmin(a, b);
	}
	return v;
}
// the ballot bits of this lane's group (HW = 32: its half of the wave's 64)
template <int HW>
__device__ __forceinline__ unsigned long long hballot(bool p, int h)
{
	const unsigned long long b = __ballot(p);
	if constexpr (HW == 64) return b;
	return h ? (b >> 32) : (b & 0xffffffffull);
}

// One 8-byte LDS read that stays one: the backend pairs neighbouring reads off one base register into ds_read2_b64,
// which occupies the CU's LDS array for 8 cycles where two ds_read_b64 take 2 each (MI355X_MICROARCH.md, LDS table) --
// and the LDS array is what the four waves of a CU share: with paired reads a wave of two 18 x 12 problems runs 39.7 us
// alone and 47.3 us beside three others, a long one 127 alone and up to 190 in a full launch.  A volatile access is
// never paired (order is kept, which the LDS does anyway).  18 x 12, 8 192 problems: 305 -> 283 us, bits unchanged.
// The whole-wave variant (HW = 64: one problem per wave, 38 x 29) is bound by its own instruction count instead and
// keeps the paired reads (unpaired: 2.51 -> 2.75 ms).
// ... except at 64 padded variables (62 x 47): there the paired form needs far more than the 512 registers a wave has
// (298 spilled: part of the inverse's row lives in scratch and is reloaded term by term in every product; the single
// reads 116): 7.09 -> 5.49 ms per 6 456.
template <int HW, int NVMAX>
__device__ __forceinline__ double lds1(const double *p)
{
#ifndef ASIF_INV_PAIRED_READS
	if constexpr (HW == 32 || NVMAX > 40) return *(const volatile __attribute__((address_space(3))) double *)p; // (p points into LDS: every caller's arrays do)
#endif
	return *p;
}

template <int NVMAX, int NCMAX, int HW>
struct InvQp {
	static constexpr int RS = NCMAX + 1; // odd: the column walk of a variable lane and the row walk of a row lane are conflict-free
	// this half's LDS
	double *At, *va, *vr, *rb, *ls_s, *ls_d, *ls_l, *ls_u, *ls_m, *ls_t;
	int t, h, nv, nc;
	bool isv, isr;
	double x, xh, q, Pd, D, ab, Eb, lbs, ubs, yb, mub, imub; // variable t (imub = 1 / mub)
	double l, u, E, y, mu, imu;                              // row t (imu = 1 / mu)
	double cs;
	double Kr[NVMAX]; // row t of K_J^-1

	// the workgroup is one wavefront, whose LDS instructions execute in order: a compiler fence is all a phase needs
	__device__ __forceinline__ void sync()
	{
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	// rows: a_t . v  (v per variable lane);  ABS: |a_t| . |v|
	template <bool ABS = false>
	__device__ __forceinline__ double row_dot(double v)
	{
		va[t] = isv ? (ABS ? fabs(v) : v) : 0.0;
		sync();
		double s0 = 0.0, s1 = 0.0; // two chains: a dependent FP64 FMA issues every ~6 cycles, an independent one every ~5
		const double *col = At + (t < NCMAX ? t : 0); // lanes beyond the padded rows shadow row 0 and are masked
#pragma unroll
		for (int j = 0; j < NVMAX; j += 2) {
			s0 += (ABS ? fabs(lds1<HW, NVMAX>(col + j * RS)) : lds1<HW, NVMAX>(col + j * RS)) * lds1<HW, NVMAX>(va + j);
			s1 += (ABS ? fabs(lds1<HW, NVMAX>(col + (j + 1) * RS)) : lds1<HW, NVMAX>(col + (j + 1) * RS)) * lds1<HW, NVMAX>(va + j + 1);
			if (j % 8 == 6) __builtin_amdgcn_sched_barrier(0); // eight terms' loads in flight, not all of them: registers
		}
		sync();
		return isr ? s0 + s1 : 0.0;
	}
	// variables: sum_i a_it w_i  (w per row lane)
	template <bool ABS = false>
	__device__ __forceinline__ double col_dot(double w)
	{
		vr[t] = isr ? (ABS ? fabs(w) : w) : 0.0;
		sync();
		double s0 = 0.0, s1 = 0.0;
		const double *row = At + (t < NVMAX ? t : 0) * RS; // lanes beyond the padded variables read row 0 and are masked by the caller
#pragma unroll
		for (int i = 0; i < NCMAX; i += 2) {
			s0 += (ABS ? fabs(lds1<HW, NVMAX>(row + i)) : lds1<HW, NVMAX>(row + i)) * lds1<HW, NVMAX>(vr + i);
			s1 += (ABS ? fabs(lds1<HW, NVMAX>(row + i + 1)) : lds1<HW, NVMAX>(row + i + 1)) * lds1<HW, NVMAX>(vr + i + 1);
			if (i % 8 == 6) __builtin_amdgcn_sched_barrier(0);
		}
		sync();
		return isv ? s0 + s1 : 0.0;
	}
	// one pass over this row's coefficients for a_t . v and |a_t| . |v|
	__device__ __forceinline__ void row_dot_pair(double v, double &plain, double &absd)
	{
		va[t] = isv ? v : 0.0;
		sync();
		double s0 = 0.0, s1 = 0.0;
		const double *col = At + (t < NCMAX ? t : 0);
#pragma unroll
		for (int j = 0; j < NVMAX; j++) {
			const double a = lds1<HW, NVMAX>(col + j * RS), b = lds1<HW, NVMAX>(va + j);
			s0 += a * b;
			s1 += fabs(a) * fabs(b);
			if (j % 8 == 7) __builtin_amdgcn_sched_barrier(0);
		}
		sync();
		plain = isr ? s0 : 0.0;
		absd = isr ? s1 : 0.0;
	}
	// one pass over this variable's column for sum_i a_it w_i and sum_i |a_it| |w_i|
	__device__ __forceinline__ void col_dot_pair(double w, double &plain, double &absd)
	{
		vr[t] = isr ? w : 0.0;
		sync();
		double s0 = 0.0, s1 = 0.0;
		const double *row = At + (t < NVMAX ? t : 0) * RS;
#pragma unroll
		for (int i = 0; i < NCMAX; i++) {
			const double a = lds1<HW, NVMAX>(row + i), b = lds1<HW, NVMAX>(vr + i);
			s0 += a * b;
			s1 += fabs(a) * fabs(b);
			if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
		}
		sync();
		plain = isv ? s0 : 0.0;
		absd = isv ? s1 : 0.0;
	}
	// the outer update's three products in one pass: A'y, |A|'|y| and A'v
	__device__ __forceinline__ void col_dot_triple(double y_, double v_, double &aty, double &atya, double &atv)
	{
		vr[t] = isr ? y_ : 0.0;
		rb[t] = isr ? v_ : 0.0;
		sync();
		double s0 = 0.0, s1 = 0.0, s2 = 0.0;
		const double *row = At + (t < NVMAX ? t : 0) * RS;
#pragma unroll
		for (int i = 0; i < NCMAX; i++) {
			const double a = lds1<HW, NVMAX>(row + i), b = lds1<HW, NVMAX>(vr + i);
			s0 += a * b;
			s1 += fabs(a) * fabs(b);
			s2 += a * lds1<HW, NVMAX>(rb + i);
			if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
		}
		sync();
		aty = isv ? s0 : 0.0;
		atya = isv ? s1 : 0.0;
		atv = isv ? s2 : 0.0;
	}
	// K^-1 v for a per-variable vector (entries beyond nv are zero on both sides)
	__device__ __forceinline__ double kinv_mul(double v)
	{
		va[t] = isv ? v : 0.0;
		sync();
		double s0 = 0.0, s1 = 0.0;
#pragma unroll
		for (int j = 0; j < NVMAX; j += 2) {
			s0 += Kr[j] * lds1<HW, NVMAX>(va + j);
			s1 += Kr[j + 1] * lds1<HW, NVMAX>(va + j + 1);
			if (j % 8 == 6) __builtin_amdgcn_sched_barrier(0);
		}
		sync();
		return s0 + s1;
	}
	// K^-1 a_i for a row of the block: its coefficients are read where they lie (column i of At; the padding is zero),
	// no broadcast vector to write and wait for
	__device__ __forceinline__ double kinv_row(int i)
	{
		double s0 = 0.0, s1 = 0.0;
		const double *col = At + i;
#pragma unroll
		for (int j = 0; j < NVMAX; j += 2) {
			s0 += Kr[j] * lds1<HW, NVMAX>(col + j * RS);
			s1 += Kr[j + 1] * lds1<HW, NVMAX>(col + (j + 1) * RS);
			if (j % 8 == 6) __builtin_amdgcn_sched_barrier(0);
		}
		return s0 + s1;
	}
	// K^-1 <- (K + c v v')^-1 given u = K^-1 v in every lane's ut (its own component) and vu = v'u:
	// K^-1 -= u u' c / (1 + c vu)
	__device__ __forceinline__ void rank_one(double ut, double vu, double c, bool on)
	{
		rb[t] = ut;
		sync();
		const double coef = on ? c * fast_rcp(1.0 + c * vu) * ut : 0.0; // denominator >= 1
#pragma unroll
		for (int j = 0; j < NVMAX; j++) {
			Kr[j] -= coef * lds1<HW, NVMAX>(rb + j);
			if (j % 8 == 7) __builtin_amdgcn_sched_barrier(0);
		}
		sync();
	}

	// power-of-two Ruiz equilibration of [P A'; A 0] with the identity block of the bounds (qp_lds.hpp: scale)
	__device__ __forceinline__ void scale(int iters)
	{
		cs = 1.0;
		D = 1.0;
		ab = 1.0;
		Eb = 1.0;
		E = 1.0;
		for (int it = 0; it < iters; it++) {
			double v = 0.0;
			{
				const double *row = At + (t < NVMAX ? t : 0) * RS;
#pragma unroll
				for (int i = 0; i < NCMAX; i++) v = fmax(v, fabs(row[i]));
				v = isv ? v : 0.0;
			}
			v = fmax(v, fmax(fabs(Pd), fabs(ab)));
			const double Dt = isv ? pow2_rsqrt(limit_scaling(v)) : 1.0;
			double rn = 0.0;
			const int tr = t < NCMAX ? t : 0; // lanes beyond the padded rows shadow row 0 (their results are masked)
#pragma unroll
			for (int j = 0; j < NVMAX; j++) rn = fmax(rn, fabs(At[j * RS + tr]));
			const double Et = isr ? pow2_rsqrt(limit_scaling(rn)) : 1.0;
			E *= Et;
			va[t] = Dt;
			sync();
			if (isr) {
#pragma unroll
				for (int j = 0; j < NVMAX; j++) At[j * RS + t] *= Et * va[j];
			}
			sync();
			const double Etb = pow2_rsqrt(limit_scaling(fabs(ab)));
			Eb *= Etb;
			ab *= Etb * Dt;
			Pd *= Dt * Dt;
			q *= Dt;
			D *= Dt;
			const double cm = hsum<HW>(isv ? fabs(Pd) : 0.0) / (double)nv;
			const double qn = limit_scaling(hmax<HW>(isv ? fabs(q) : 0.0));
			const double ct = pow2_floor_inv(limit_scaling(fmax(cm, qn)));
			Pd *= ct;
			q *= ct;
			cs *= ct;
		}
	}
};

constexpr int kInvFewBp = 8; // breakpoints of a line search up to which phi' is evaluated at each by a reduction (qp_inv_kernel)
// status / iters follow asif_hip_qp_solve_batch's contract (QPWrapperOsqp::solve, src/qpwrapper_osqp.cpp:225-238)
#ifndef ASIF_INV_MIN_WAVES
#define ASIF_INV_MIN_WAVES 1 // waves per SIMD the register allocation is held to (scratch builds try 2)
#endif
// MINW: waves per SIMD the register allocation is held to.  Two cost a wave 15 % (spills to scratch; a lone 19-step
// problem 118 -> 138 us) and buy a second wave to run beside it: nothing at 8 192 problems of 18 x 12 (281 / 278 us: the
// launch ends with its last long wave either way), 816 -> 668 us at 32 768 -- the launcher asks for two from 16 384 on.
// WARM: the instantiation behind asif_hip_qp_solve_batch_warm -- the start is read from, and the final iterate and
// multipliers are written to, QpArgs::warm_x / warm_y in the caller's units (x = D xs, y = E ys / cs: the scalings are
// powers of two, the round trip is exact).  The cold instantiations carry none of it.
template <int NVMAX, int NCMAX, int HW = 32, int MINW = ASIF_INV_MIN_WAVES, bool WARM = false>
__global__ __launch_bounds__(64, MINW) void qp_inv_kernel(asif_hip_solver S_, QpArgs a)
{
	static_assert((HW == 32 || HW == 64) && NVMAX % 2 == 0 && NCMAX % 2 == 0 && NVMAX <= HW && NCMAX <= HW, "padded sizes");
#ifdef ASIF_INV_WAVETIME
	const long long wt0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
	extern __shared__ double lds[];
	InvQp<NVMAX, NCMAX, HW> s;
	constexpr int RS = NCMAX + 1, QPW = 64 / HW; // problems per wave
	const int lane = threadIdx.x, h = lane / HW, t = lane % HW;
	const int nv = a.nv, nc = a.nc;
	const int64_t npair = (a.B + QPW - 1) / QPW;
	const int64_t pi = xcd_contiguous_index(blockIdx.x, npair);
	if (pi >= npair) return; // wave-uniform (one wave per workgroup)
	int64_t qi = QPW * pi + h;
	bool keep = qi < a.B; // an odd batch leaves the second half of its last wave without a problem
	if (!keep) qi = a.B - 1;
	if (a.only_status != 0 && a.status[qi] != a.only_status) keep = false; // second pass: already decided
	if (!__any(keep)) return;
	const int64_t ld = a.ld;
	s.t = t;
	s.h = h;
	s.nv = nv;
	s.nc = nc;
	{
		double *p = lds + (size_t)h * inv_half_doubles(NVMAX, NCMAX, HW);
		s.At = p; p += (size_t)NVMAX * RS;
		s.va = p; p += HW;
		s.vr = p; p += HW;
		s.rb = p; p += HW;
		s.ls_s = p; p += 2 * HW;
		s.ls_d = p; p += 2 * HW;
		s.ls_l = p; p += 2 * HW;
		s.ls_u = p; p += 2 * HW;
		s.ls_m = p; p += 2 * HW;
		s.ls_t = p;
	}
	// ---- form translation (src/qpwrapper_osqp.cpp:263-376): P = 2H, q = c, rows [A; I], l = [b; lb], u = [inf | b; ub]
	s.isr = t < nc;
	s.isv = t < nv;
	double dom = 0.0; // data outside the solvers' domain (NaN, inf, beyond 1e148): qp_lane.hpp, qp_data_nonfinite
	bool nanb = false;
	{
		double lo = -kInfty, hi = kInfty;
		// padding rows and columns are zero
		for (int e = t; e < NVMAX * RS; e += HW) s.At[e] = 0.0;
		s.sync();
		if (s.isr) {
			for (int j = 0; j < nv; j++) {
				const double v = a.A[(int64_t)(t + j * nc) * ld + qi];
				s.At[j * RS + t] = v;
				dom = fma(v, 1e160, dom);
			}
			lo = a.b[(int64_t)t * ld + qi];
			dom = fma(lo, 1e160, dom);
			hi = ((a.be_mask >> t) & 1ull) ? lo : kInfty; // t < nc <= 64: the first mask word
		}
		s.l = lo;
		s.u = hi;
	}
	s.Pd = 0.0;
	s.q = 0.0;
	s.lbs = -kInfty;
	s.ubs = kInfty;
	if (s.isv) {
		s.q = a.c[(int64_t)t * ld + qi];
		s.lbs = a.lb[(int64_t)t * ld + qi];
		s.ubs = a.ub[(int64_t)t * ld + qi];
		s.Pd = 2.0 * a.Hd[(int64_t)t * ld + qi];
		dom = fma(s.q, 1e160, fma(s.Pd, 1e160, dom));
		nanb = (s.lbs != s.lbs) | (s.ubs != s.ubs);
	}
	const bool outside = hballot<HW>(nanb | !(fabs(dom) < __builtin_huge_val()), h) != 0;
	s.va[t] = 0.0;
	s.vr[t] = 0.0;
	s.rb[t] = 0.0;
	s.sync();
	s.scale(S_.scaling_iters);
	s.l *= s.E;
	s.u *= s.E;
	s.y = 0.0;
	s.mu = (s.u - s.l < kRhoTol) ? 100.0 * kLdsMu0 : kLdsMu0;
	s.imu = 1.0 / s.mu;
	s.lbs *= s.Eb;
	s.ubs *= s.Eb;
	s.x = 0.0;
	s.xh = 0.0;
	s.yb = 0.0;
	s.mub = (s.ubs - s.lbs < kRhoTol) ? 100.0 * kLdsMu0 : kLdsMu0;
	s.imub = 1.0 / s.mub;
	if constexpr (WARM) {
		if (a.warm_in != 0 && !outside) {
			// a start that is not a finite number of sane size is no start (the buffer of a problem that was never solved)
			auto sane = [](double v) { return fabs(v) < 1e100 ? v : 0.0; };
			if (s.isv) {
				s.x = sane(a.warm_x[(int64_t)t * ld + qi]) / s.D;
				s.xh = s.x;
				s.yb = sane(a.warm_y[(int64_t)(nc + t) * ld + qi]) * s.cs / s.Eb;
			}
			if (s.isr) s.y = sane(a.warm_y[(int64_t)t * ld + qi]) * s.cs / s.E;
		}
	}
#pragma unroll
	for (int j = 0; j < NVMAX; j++) s.Kr[j] = 0.0;
	const double tol = fmax(S_.eps_rel, 1e-10) * 1e-2; // default eps 1e-8 -> 1e-10 on the scaled residuals
	const double big = kInfty * kMinScaling;
	const double ig = 1.0 / kLdsGamma;
	const int max_newton = S_.max_iter > 0 ? S_.max_iter : 4000;
	// per half: the problem's verdict (0 = still running), its Newton count, the state of its inverse
	int status = outside ? kStatusMaxIter : 0, newton = 0;
	bool kvalid = false, pactr = false, pactb = false, stiff = false;
	double pri_prev = -1.0, best_res = 1e300;
	[[maybe_unused]] int met = 0; // WARM: consecutive multiplier updates that met the termination test

	// section timers of a scratch build (tools/dev_inv_sections.py); compiled out of the library
#ifdef ASIF_INV_PROFILE
	long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmark = __builtin_readcyclecounter();
	int cnt_rebuild = 0, cnt_rows = 0, cnt_bounds = 0;
#define INV_T(k) { const long long tn_ = __builtin_readcyclecounter(); tsec[k] += tn_ - tmark; tmark = tn_; }
#else
#define INV_T(k)
#endif
	INV_T(6)
	for (int outer = 0; outer < kLdsMaxOuter; outer++) {
		if (!__any(status == 0)) break;
		const bool run = status == 0;
		double gfloor = 0.0, gscale = 1.0;
		bool inn = run; // this half is still inside its inner solve
		for (int inner = 0; inner < kLdsMaxInner; inner++) {
			if (!__any(inn)) break;
			// ---- gradient of the inner objective
			double ax, axa = 0.0, atr, atra = 0.0; // the |.| products feed the scale and floor of the first step only
			if (inner == 0) s.row_dot_pair(s.x, ax, axa);
			else ax = s.row_dot(s.x);
			const double sr = ax + s.y * s.imu;
			const double rr = s.isr ? s.mu * (sr - fmin(fmax(sr, s.l), s.u)) : 0.0;
			const bool actr = s.isr && (sr < s.l || sr > s.u);
			if (inner == 0) s.col_dot_pair(rr, atr, atra);
			else atr = s.col_dot(rr);
			const double px = s.Pd * s.x;
			const double sb = s.ab * s.x + s.yb * s.imub;
			const double rbv = s.isv ? s.mub * (sb - fmin(fmax(sb, s.lbs), s.ubs)) : 0.0;
			const bool actb = s.isv && (sb < s.lbs || sb > s.ubs);
			const double g = s.isv ? px + s.q + (s.x - s.xh) * ig + atr + s.ab * rbv : 0.0;
			const double gn = hmax<HW>(fabs(g));
			if (inner == 0) {
				// scale of the gradient's own terms and its rounding floor, once per inner solve (qp_lds.hpp)
				const double gs = s.isv ? fmax(fabs(px), fmax(fabs(s.q), atra + fabs(s.ab * rbv))) : 0.0;
				const double gsc = 1.0 + hmax<HW>(gs);
				const double bd = sr < s.l ? fabs(s.l) : (sr > s.u ? fabs(s.u) : 0.0);
				const double er = s.isr ? 2.2e-16 * s.mu * (axa + fabs(s.y) * s.imu + bd) : 0.0;
				const double fl = s.template col_dot<true>(er);
				const double bdb = sb < s.lbs ? fabs(s.lbs) : (sb > s.ubs ? fabs(s.ubs) : 0.0);
				const double eb = 2.2e-16 * s.mub * (fabs(s.ab * s.x) + fabs(s.yb) * s.imub + bdb);
				const double f = s.isv ? fl + fabs(s.ab) * eb : 0.0;
				gscale = gsc;
				gfloor = hmax<HW>(f) + 2.2e-16 * gsc;
			}
			inn = inn && !(gn <= 0.1 * tol * gscale || gn <= 8.0 * gfloor) && newton < max_newton;
			INV_T(0)
			if (!__any(inn)) break;
			// ---- K_J^-1 for the current active set
			{
				const bool gone = (pactr && !actr) || (pactb && !actb);
				const bool rebuild = inn && (!kvalid || hballot<HW>(gone, h) != 0);
#ifdef ASIF_INV_PROFILE
				cnt_rebuild += rebuild ? 1 : 0;
#endif
				if (__any(rebuild)) {
					// the bounds that are active now are diagonal terms: they start in the diagonal matrix the rebuild starts
					// from instead of entering one rank-one step each afterwards (a third of all steps on the lifted problems)
					const double dg = s.isv ? fast_rcp(s.Pd + ig + (actb ? s.mub * s.ab * s.ab : 0.0)) : 0.0;
#pragma unroll
					for (int j = 0; j < NVMAX; j++) s.Kr[j] = rebuild ? (j == t ? dg : 0.0) : s.Kr[j];
				}
				bool pendr = inn && actr && (rebuild || !pactr), pendb = inn && actb && !rebuild && !pactb;
				kvalid = kvalid || rebuild;
				pactr = inn ? actr : pactr;
				pactb = inn ? actb : pactb;
				// bounds entering: K += c e_j e_j', c = mu_b ab^2; K^-1 e_j is column j = (symmetry) row j, held by lane j
				s.vr[t] = s.mub * s.ab * s.ab;
				s.sync();
				while (__any(pendb)) {
					const unsigned long long m = hballot<HW>(pendb, h);
					const bool on = m != 0;
					const int j = on ? __ffsll((long long)m) - 1 : 0;
					if (on && t == j) {
#pragma unroll
						for (int k = 0; k < NVMAX; k++) s.va[k] = s.Kr[k];
					}
					s.sync();
					const double ut = s.va[t], ujj = s.va[j], c = s.vr[j];
					s.sync();
					s.rank_one(ut, ujj, c, on);
#ifdef ASIF_INV_PROFILE
					cnt_bounds += on ? 1 : 0;
#endif
					pendb = pendb && t != j;
				}
				// rows entering: K += mu_i a_i a_i'
				s.vr[t] = s.mu;
				s.sync();
				while (__any(pendr)) {
					const unsigned long long m = hballot<HW>(pendr, h);
					const bool on = m != 0;
					const int i = on ? __ffsll((long long)m) - 1 : 0;
					const double at = s.isv ? s.At[t * RS + i] : 0.0;
					const double c = s.vr[i];
					const double ut = s.kinv_row(i);
					const double vu = hsum<HW>(at * ut);
					s.rank_one(ut, vu, c, on);
#ifdef ASIF_INV_PROFILE
					cnt_rows += on ? 1 : 0;
#endif
					pendr = pendr && t != i;
				}
			}
			INV_T(1)
			// ---- Newton direction: d = -K^-1 g, one step of refinement through the exact operator
			double d = -s.kinv_mul(g);
			{
				const double ad = s.row_dot(d);
				const double kd = s.col_dot(actr ? s.mu * ad : 0.0) + (s.Pd + ig + (actb ? s.mub * s.ab * s.ab : 0.0)) * d;
				const double res = s.isv ? -g - kd : 0.0;
				d += s.kinv_mul(res);
				// A second step once this problem's penalties have been raised (the hard 5 %: K_J's condition is then
				// mu |a|^2 gamma ~ 1e11 and the running inverse carries its 1e-8 into a direction that one step does not
				// clean: with the penalties raised in one jump -- qp_lds.hpp, penalty_jump -- single instances zigzagged
				// through 60-75 Newton steps; with it none exceeds 19).  Problems that never leave the first penalty,
				// 95 % of the seeded ones, do not pay for it.
				if (__any(stiff && inn)) {
					const double ad2 = s.row_dot(d);
					const double kd2 = s.col_dot(actr ? s.mu * ad2 : 0.0) + (s.Pd + ig + (actb ? s.mub * s.ab * s.ab : 0.0)) * d;
					const double res2 = s.isv ? -g - kd2 : 0.0;
					const double corr = s.kinv_mul(res2);
					d += stiff ? corr : 0.0;
				}
			}
			newton += inn ? 1 : 0;
			INV_T(2)
			// ---- exact line search: phi'(t) = qa + t a1 + sum_i mu_i dl_i (s_i + t dl_i - proj(s_i + t dl_i))
			const double dl = s.row_dot(d);
			const double dlb = s.ab * d;
			const double qat = s.isv ? (px + s.q + (s.x - s.xh) * ig) * d : 0.0;
			const double a1t = s.isv ? (s.Pd + ig) * d * d : 0.0;
			double tstep = 1.0;
			{
				// phi'(1) alone first, in ONE reduction: phi' is nondecreasing, so phi'(1) < 0 means that no breakpoint in
				// (0, 1] brackets a zero (and that phi'(0) = g.d < 0: a descent direction)
				const double s1 = sr + dl, s1b = sb + dlb;
				const double fr = s.isr ? s.mu * dl * (s1 - fmin(fmax(s1, s.l), s.u)) : 0.0;
				const double fb = s.isv ? s.mub * dlb * (s1b - fmin(fmax(s1b, s.lbs), s.ubs)) : 0.0;
				const double f1 = hsum<HW>(qat + a1t + fr + fb);
				const bool cut = inn && !(f1 < 0.0);
				INV_T(3)
				if (__any(cut)) {
					// this lane's breakpoints in (0, 1]: where its row and its bound cross an end of their interval
					double bp[4];
					{
						const double idl = fast_rcp(dl), idlb = fast_rcp(dlb); // a zero or denormal dl gives NaN: no breakpoint
						const double t1 = (s.l - sr) * idl, t2 = (s.u - sr) * idl;
						bp[0] = (s.isr && t1 > 0.0 && t1 <= 1.0) ? t1 : 2.0;
						bp[1] = (s.isr && t2 > 0.0 && t2 <= 1.0) ? t2 : 2.0;
						const double t3 = (s.lbs - sb) * idlb, t4 = (s.ubs - sb) * idlb;
						bp[2] = (s.isv && t3 > 0.0 && t3 <= 1.0) ? t3 : 2.0;
						bp[3] = (s.isv && t4 > 0.0 && t4 <= 1.0) ? t4 : 2.0;
					}
					// No breakpoint at all -- every Newton step of the 95 % of the seeded lifted problems that take four: the
					// step is cut at 0.993-0.998 by the refinement's last digits, nothing changes sides on the way -- and phi' is
					// LINEAR on [0, 1]: its zero follows from phi'(0) = g.d (one more reduction) and the phi'(1) above, and the
					// search below (ten LDS vectors, a pass over all rows per lane, four reductions: a quarter of a wave's time)
					// has nothing to find.
					const bool anybp = hballot<HW>((bp[0] <= 1.0) | (bp[1] <= 1.0) | (bp[2] <= 1.0) | (bp[3] <= 1.0), h) != 0;
					double tlo = 0.0, thi = 1.0, fhi = f1;
					double flo = hsum<HW>(qat + dl * rr + dlb * rbv); // (dl, dlb, rr, rbv are zero on the lanes that own nothing)
					if (__any(cut && anybp)) {
					// the breakpoints of this half, side by side in LDS behind the two end points
					int npt = 2;
					if (t == 0) s.ls_t[0] = 0.0;
					if (t == 1) s.ls_t[1] = 1.0;
#pragma unroll
					for (int e = 0; e < 4; e++) {
						const bool v = bp[e] <= 1.0;
						const unsigned long long m = hballot<HW>(v, h);
						if (v) s.ls_t[npt + __popcll(m & ((1ull << t) - 1ull))] = bp[e];
						npt += __popcll(m);
					}
					s.sync();
					double tlo_s = 0.0, flo_s = flo, thi_s = 1.0, fhi_s = f1;
					const bool few = npt <= 2 + kInvFewBp; // (per half: what a half gets must not depend on its neighbour's count)
					{
						// A handful of them (2.5 on average where there is any): phi' at each by ONE reduction, every lane its own
						// row and bound, instead of every lane a pass over all rows.  The two end values are the ones above.
						for (int k = 2; __any(cut && few && k < npt); k++) {
							const bool have = few && k < npt;
							const double tk = have ? s.ls_t[k] : 1.0;
							const double sk = sr + tk * dl, skb = sb + tk * dlb;
							const double fr_k = s.isr ? s.mu * dl * (sk - fmin(fmax(sk, s.l), s.u)) : 0.0;
							const double fb_k = s.isv ? s.mub * dlb * (skb - fmin(fmax(skb, s.lbs), s.ubs)) : 0.0;
							const double fk = hsum<HW>(qat + tk * a1t + fr_k + fb_k);
							// best lower bracket: largest t with f < 0; best upper: smallest t with f >= 0
							if (have && fk < 0.0 && tk > tlo_s) { tlo_s = tk; flo_s = fk; }
							if (have && !(fk < 0.0) && tk < thi_s) { thi_s = tk; fhi_s = fk; }
						}
					}
					if (__any(cut && !few)) {
					double tlo_f = tlo_s, flo_f = flo_s, thi_f = thi_s, fhi_f = fhi_s;
					const double qa = hsum<HW>(qat), a1 = hsum<HW>(a1t);
					// all rows (general, then bounds) side by side in LDS
					s.ls_s[t] = sr;
					s.ls_d[t] = s.isr ? dl : 0.0;
					s.ls_l[t] = s.l;
					s.ls_u[t] = s.u;
					s.ls_m[t] = s.isr ? s.mu * dl : 0.0; // mu_i dl_i
					s.ls_s[HW + t] = sb;
					s.ls_d[HW + t] = s.isv ? dlb : 0.0;
					s.ls_l[HW + t] = s.lbs;
					s.ls_u[HW + t] = s.ubs;
					s.ls_m[HW + t] = s.isv ? s.mub * dlb : 0.0;
					s.sync();
					auto dphi = [&](double tt) {
						double f0 = qa + tt * a1, f1c = 0.0;
#pragma unroll 2
						for (int i = 0; i < NCMAX; i += 2) { // general rows (padding rows carry mu dl = 0)
							const double st = lds1<HW, NVMAX>(s.ls_s + i) + tt * lds1<HW, NVMAX>(s.ls_d + i);
							f0 += lds1<HW, NVMAX>(s.ls_m + i) * (st - fmin(fmax(st, lds1<HW, NVMAX>(s.ls_l + i)), lds1<HW, NVMAX>(s.ls_u + i)));
							const double su = lds1<HW, NVMAX>(s.ls_s + i + 1) + tt * lds1<HW, NVMAX>(s.ls_d + i + 1);
							f1c += lds1<HW, NVMAX>(s.ls_m + i + 1) * (su - fmin(fmax(su, lds1<HW, NVMAX>(s.ls_l + i + 1)), lds1<HW, NVMAX>(s.ls_u + i + 1)));
						}
#pragma unroll 2
						for (int i = HW; i < HW + NVMAX; i += 2) { // bounds
							const double st = lds1<HW, NVMAX>(s.ls_s + i) + tt * lds1<HW, NVMAX>(s.ls_d + i);
							f0 += lds1<HW, NVMAX>(s.ls_m + i) * (st - fmin(fmax(st, lds1<HW, NVMAX>(s.ls_l + i)), lds1<HW, NVMAX>(s.ls_u + i)));
							const double su = lds1<HW, NVMAX>(s.ls_s + i + 1) + tt * lds1<HW, NVMAX>(s.ls_d + i + 1);
							f1c += lds1<HW, NVMAX>(s.ls_m + i + 1) * (su - fmin(fmax(su, lds1<HW, NVMAX>(s.ls_l + i + 1)), lds1<HW, NVMAX>(s.ls_u + i + 1)));
						}
						return f0 + f1c;
					};
					// bracket of the zero of phi' among {0} u breakpoints u {1}: every lane evaluates phi' at its own point
					tlo_s = 0.0, flo_s = 0.0, thi_s = 2.0, fhi_s = 0.0;
					for (int base = 0; __any(base < npt); base += HW) {
						const bool mine = base + t < npt;
						const double tb = mine ? s.ls_t[base + t] : 1.0;
						const double fbv = dphi(tb);
						if (base == 0) {
							// points 0 and 1 of this half: through LDS (the lanes that hold them differ between the halves)
							if (t < 2) s.rb[t] = fbv;
							s.sync();
							const double f0 = s.rb[0], f1b = s.rb[1];
							s.sync();
							flo_s = f0;
							if (f1b >= 0.0) { thi_s = 1.0; fhi_s = f1b; }
						}
						const bool isbp = mine && (base + t >= 2);
						// best lower bracket: largest t with f < 0; best upper: smallest t with f >= 0
						const double cl = (isbp && fbv < 0.0) ? tb : -1.0;
						const double ch = (isbp && fbv >= 0.0) ? tb : 3.0;
						const double gl = hmax<HW>(cl), gh = hmin<HW>(ch);
						const double flc = hmax<HW>(cl == gl ? fbv : -1e300); // f at that breakpoint (negative: max picks it among ties)
						const double fhc = hmin<HW>(ch == gh ? fbv : 1e300);
						if (gl > tlo_s) { tlo_s = gl; flo_s = flc; }
						if (gh < thi_s) { thi_s = gh; fhi_s = fhc; }
					}
					s.sync();
					if (few) { // (a half with few breakpoints keeps what the reductions found)
						tlo_s = tlo_f;
						flo_s = flo_f;
						thi_s = thi_f;
						fhi_s = fhi_f;
					}
					}
					if (anybp) { // (a half without breakpoints keeps its two end points, whatever its neighbour had to search)
						tlo = tlo_s;
						flo = flo_s;
						thi = thi_s;
						fhi = fhi_s;
					}
					}
					double tt = 1.0;
					if (thi <= 1.0) tt = (fhi > flo) ? tlo - flo * (thi - tlo) / (fhi - flo) : tlo;
					if (!(tt > 0.0)) tt = thi <= 1.0 ? thi : 1.0; // degenerate bracket: take the upper end
					// phi'(0) = g.d >= 0: not a descent direction (never seen; an inverse gone wrong) -- no step, and the
					// inverse is rebuilt from the diagonal at the next one
					const bool ascent = cut && !(flo < 0.0) && tlo == 0.0;
					kvalid = kvalid && !ascent;
					tstep = cut ? (ascent ? 0.0 : tt) : 1.0;
				}
			}
			s.x += inn ? tstep * d : 0.0;
			INV_T(4)
		}
		INV_T(0)
		// ---- multiplier update, residuals, certificates
		const double ax = s.row_dot(s.x);
		double pri = 0.0, nax = 0.0, ndy = 0.0, lhs = 0.0, vcert, ynew;
		{
			const double sv = ax + s.y * s.imu;
			ynew = s.isr ? s.mu * (sv - fmin(fmax(sv, s.l), s.u)) : 0.0;
			const double viol = ax - fmin(fmax(ax, s.l), s.u);
			if (s.isr) {
				pri = fabs(viol);
				nax = fabs(ax);
			}
			double v = ynew - s.y;
			if (s.u > big) v = (s.l < -big) ? 0.0 : fmin(v, 0.0);
			else if (s.l < -big) v = fmax(v, 0.0);
			vcert = s.isr ? v : 0.0;
			ndy = fabs(vcert);
			lhs = vcert > 0.0 ? s.u * vcert : (vcert < 0.0 ? s.l * vcert : 0.0);
		}
		double aty, atv, atya;
		s.col_dot_triple(ynew, vcert, aty, atya, atv);
		const double px = s.Pd * s.x, pxa = fabs(px);
		double dua = 0.0, nd = 0.0, natv = 0.0, ybn;
		{
			const double axb = s.ab * s.x;
			const double sv = axb + s.yb * s.imub;
			ybn = s.isv ? s.mub * (sv - fmin(fmax(sv, s.lbs), s.ubs)) : 0.0;
			const double viol = axb - fmin(fmax(axb, s.lbs), s.ubs);
			double v = ybn - s.yb;
			if (s.ubs > big) v = (s.lbs < -big) ? 0.0 : fmin(v, 0.0);
			else if (s.lbs < -big) v = fmax(v, 0.0);
			if (!s.isv) v = 0.0;
			if (s.isv) {
				pri = fmax(pri, fabs(viol));
				nax = fmax(nax, fabs(axb));
				const double ay = aty + s.ab * ybn;
				dua = fabs(px + s.q + ay);
				nd = fmax(pxa, fmax(fabs(s.q), atya + fabs(s.ab * ybn)));
				ndy = fmax(ndy, fabs(v));
				lhs += v > 0.0 ? s.ubs * v : (v < 0.0 ? s.lbs * v : 0.0);
				natv = fabs(atv + s.ab * v);
			}
		}
		if (run) { // a decided half keeps its iterate and multipliers
			s.y = ynew;
			s.yb = ybn;
			s.xh = s.x;
		}
		pri = hmax<HW>(pri); nax = hmax<HW>(nax); dua = hmax<HW>(dua); nd = hmax<HW>(nd);
		ndy = hmax<HW>(ndy); lhs = hsum<HW>(lhs); natv = hmax<HW>(natv);
		const double rp = pri / (1.0 + nax), rd = dua / (1.0 + nd);
		int st = 0;
		bool done = rp <= tol && rd <= tol;
		if constexpr (WARM) {
			// From a cold start the test is first met after the multipliers have gone through an update on a settled active
			// set, and the point it accepts is far better than the test asks (|u - u_ref| 4e-9 ... 1e-7).  A warm start can
			// meet it at once, at a point only as good as the relative test -- 1e-10 of multipliers of size 1e4-1e5 is 2e-5
			// in u on the realizable filter's problems (host build of the method, same split): a warm solve is done when
			// THREE updates in a row meet it (host build: |u - u_ref| <= 6e-8 there, a cold start's own 1e-7, for 3-4 Newton
			// steps against a cold start's 9; two in a row: 7e-7 on the host, 7e-6 on the factorising kernel; each further
			// update costs a gradient and, now and then, one Newton step).
			if (run) met = done ? met + 1 : 0;
			done = done && met >= (a.warm_in != 0 ? 3 : 1);
		}
		if (done) st = kStatusSolved;
		else if (ndy > 1e-4 && lhs < -1e-6 * ndy && natv < 1e-6 * ndy) st = kStatusPrimalInf;
		else if (newton >= max_newton) st = kStatusMaxIter;
		if (run) {
			best_res = fmin(best_res, fmax(rp, rd));
			status = st;
		}
		const double mumin = hmin<HW>(fmin(s.isr ? s.mu : 1e300, s.isv ? s.mub : 1e300));
		const double mumax = hmax<HW>(fmax(s.isr ? s.mu : 0.0, s.isv ? s.mub : 0.0));
		if (run && st == 0) {
			double f = 1.0, cap = kLdsMuMax;
			if (rp <= tol) {
				// rows are met, the dual residual is not: a softer penalty lowers the rounding floor of mu (s - proj s)
				if (mumax > 100.0 * kLdsMu0) f = 0.1;
			} else if (pri_prev >= 0.0 && pri > 0.5 * pri_prev && mumin >= kLdsMuMax) {
				f = 10.0; // stalled at the cap: the multipliers have far to go (rows with tiny coefficients)
				cap = 1e8;
			} else if (pri_prev >= 0.0 && pri > 0.1 * pri_prev) {
				f = penalty_jump(pri, pri_prev); // not fast enough -> stiffer penalties, by what the contraction seen asks for (qp_lds.hpp)
			}
			if (f != 1.0) {
				s.mu = f > 1.0 ? fmin(s.mu * f, fmax(s.mu, cap)) : fmax(s.mu * f, kLdsMu0);
				s.mub = f > 1.0 ? fmin(s.mub * f, fmax(s.mub, cap)) : fmax(s.mub * f, kLdsMu0);
				s.imu = 1.0 / s.mu;
				s.imub = 1.0 / s.mub;
				kvalid = false;
				stiff = stiff || f > 1.0;
			}
			pri_prev = pri;
		}
		INV_T(5)
	}
	if (status == 0 || (status == kStatusMaxIter && !outside)) {
		// budget spent: OSQP's "solved inaccurate" counts as solved for the wrapper (src/qpwrapper_osqp.cpp:225)
		status = best_res <= 1e3 * tol ? kStatusSolved : kStatusMaxIter;
	}
#ifdef ASIF_INV_WAVETIME
	// scratch build (tools/dev_inv_wavetime.py): when this wave started and how long it ran (100 MHz counter) instead of x
	if (keep && t == 0) {
		const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
		a.sol[qi] = (double)wt0;
		a.sol[ld + qi] = (double)(t1 - wt0);
		a.status[qi] = status;
		if (a.iters) a.iters[qi] = newton;
	}
	return;
#endif
	if (keep) {
		if (s.isv) a.sol[(int64_t)t * ld + qi] = s.D * s.x;
		if (t == 0) {
			a.status[qi] = status;
			if (a.iters) a.iters[qi] = newton;
		}
		if constexpr (WARM) {
			// what the next solve() of this workspace starts from; a problem without a solution leaves a cold start
			const bool ok = status == kStatusSolved;
			if (s.isv) {
				a.warm_x[(int64_t)t * ld + qi] = ok ? s.D * s.x : 0.0;
				a.warm_y[(int64_t)(nc + t) * ld + qi] = ok ? s.Eb * s.yb / s.cs : 0.0;
			}
			if (s.isr) a.warm_y[(int64_t)t * ld + qi] = ok ? s.E * s.y / s.cs : 0.0;
		}
#ifdef ASIF_INV_PROFILE
		if (t == 0)
			for (int k = 0; k < 7; k++) a.sol[(int64_t)k * ld + qi] = (double)tsec[k]; // scratch build: times instead of x
		if (t == 0) {
			a.sol[(int64_t)7 * ld + qi] = cnt_rebuild;
			a.sol[(int64_t)8 * ld + qi] = cnt_rows;
			a.sol[(int64_t)9 * ld + qi] = cnt_bounds;
		}
#endif
	}
#undef INV_T
}

} // namespace asif
